#!/bin/bash
# DURF_OVERLAP_OBJECTS modes against the batch size (cfg3 shape), interleaved on one box:  tools/ab_rays.sh [modes...]
modes=${@:-0 2 auto}
cd /root/repo
for rays in 512 1024 2048; do
for rep in 1 2; do
  for m in $modes; do
    DURF_OVERLAP_OBJECTS=$m python3 bench.py --config cfg3 --rays $rays --steps 200 --warmup 20 --no-cpu-baseline --no-calibration 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('rays $rays mode %-4s (object_streams %s)  %.1f k rays/s  %.3f ms/step' % ('$m', d['config']['object_streams'], d['value']/1e3, d['ms_per_step']))"
  done
done
done
