#!/bin/bash
# Same-box, interleaved: round 4's final tree (commit bc08e9b, copied to tools/experiments/_r04_tree by hand) against this one,
# bench.py on every workload.   tools/experiments/ab_r04_r05.sh > gpurun_out/ab_r04_r05.txt
root=$(pwd)
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-5s %-22s %9.1f k rays/s  %.4f ms/step' % ('$1', '$2', d['value']/1e3, d['ms_per_step']))"; }
for rep in 1 2 3; do
  for args in "--config cfg3" "--config cfg3 --rays 512" "--config cfg1" "--config cfg4" "--config cfg5" "--config cfg2"; do
    (cd $root/tools/experiments/_r04_tree && python3 bench.py $args --steps 100 --no-cpu-baseline --no-calibration 2>/dev/null | line r04 "$args")
    (cd $root && python3 bench.py $args --steps 100 --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | line r05 "$args")
  done
done
