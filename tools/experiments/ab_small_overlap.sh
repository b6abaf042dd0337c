#!/bin/bash
# small batches: object launches on the side stream BEHIND the background launch (objects-first only from 2048 rays) vs one stream
for rep in 1 2 3; do
  for args in "--config cfg3 --rays 512" "--config cfg5" "--config cfg3 --rays 1024" "--config cfg2 --rays 512"; do
    for m in 262144 0; do
      DURF_OVERLAP_MIN_ROWS=$m python3 bench.py $args --steps 100 --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_rows=%-7s %-26s streams=%s %9.1f k rays/s  %.4f ms/step' % ('$m', '$args', d['config']['object_streams'], d['value']/1e3, d['ms_per_step']))"
    done
  done
done
