#!/bin/bash
# same-box A/B of variant libraries: the fused kernels alone (tools/time_fwd.py, 4096 rays and 512 rays) and the small-batch steps
for rep in 1 2 3; do
  for v in ${@:-nopace pace}; do
    export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so
    echo -n "variant=$v  "; python3 tools/time_fwd.py 2>&1 | tail -1
    echo -n "variant=$v  512 rays: "; ROWS=65536 python3 tools/time_fwd.py 2>&1 | tail -1
  done
done
for v in ${@:-nopace pace}; do
  export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so
  for args in "--rays 512" "--config cfg1" ""; do
    python3 bench.py $args --steps 100 --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['all']
print('variant=$v bench $args: %.1f k rays/s %.4f ms/step ' % (d['value']/1e3, d['ms_per_step']), {k: round(v['us'],1) for k,v in r.items()})"
  done
done
