#!/bin/bash
# A/B of the DURF_OVERLAP_OBJECTS modes on one box, interleaved:  tools/ab_overlap.sh [config] [modes...]
cfg=${1:-cfg3}; shift
modes=${@:-0 dw dwf 2}
cd /root/repo
for rep in 1 2 3; do
  for m in $modes; do
    DURF_OVERLAP_OBJECTS=$m python3 bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-calibration 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('mode %-3s  %.1f k rays/s  %.3f ms/step  dW launch %.0f us  non-MLP %.3f ms' % ('$m', d['value']/1e3, d['ms_per_step'], r['launch_us'], r['non_mlp_ms_per_step']))"
  done
done
