#!/bin/bash
# interleaved A/B of environment switches on one box:  tools/ab_env.sh <config> "<VAR=a VAR2=b>" "<VAR=c>" ...
cfg=$1; shift
cd /root/repo
for rep in 1 2 3; do
  for e in "$@"; do
    env $e python3 bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-calibration 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-50s %.1f k rays/s  %.3f ms/step  dW launch %.0f us  non-MLP %.3f ms' % ('$e', d['value']/1e3, d['ms_per_step'], r['launch_us'], r['non_mlp_ms_per_step']))"
  done
done
