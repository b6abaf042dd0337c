#!/bin/bash
# total workgroups of the background weight-gradient launch (DURF_DW_WGS; unset = the library default), interleaved
#   WGS="0 1024" CFGS="cfg3 cfg2 cfg5" tools/sweep_dw_wgs.sh       (run on the GPU box; appends to gpurun_out/dw_wgs.log)
for rep in 1 2; do for c in ${CFGS:-cfg3}; do for w in ${WGS:-0 1024 768}; do
  echo "cfg=$c wgs=$w" >> gpurun_out/dw_wgs.log
  if [ "$w" = 0 ]; then unset DURF_DW_WGS; else export DURF_DW_WGS=$w; fi
  python bench.py --config $c --steps 100 --warmup 5 --no-cpu-baseline --no-calibration 2>&1 | tail -1 >> gpurun_out/dw_wgs.log
done; done; done
