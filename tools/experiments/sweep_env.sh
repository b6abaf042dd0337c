#!/bin/bash
# bench.py under environment variants, interleaved:  VARS="A=1 A=2" CFGS="cfg3 cfg5" ROUNDS=2 tools/sweep_env.sh <log>
log=${1:-gpurun_out/sweep_env.log}
for rep in $(seq ${ROUNDS:-2}); do for c in ${CFGS:-cfg3}; do for v in ${VARS:-X=0}; do
  echo "cfg=$c $v" >> $log
  env $v python bench.py --config $c --no-cpu-baseline --no-calibration 2>&1 | tail -1 >> $log
done; done; done
