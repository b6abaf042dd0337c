#!/bin/bash
# kernel durations (rocprofv3 --stats) of the weight-gradient / finalize launches under env variants
#   VARS="DURF_DW_WGS_OBJ=256 DURF_DW_WGS_OBJ=512 X=0" CFGS="cfg3 cfg5" tools/fin_check.sh
cd /tmp && export TMPDIR=/tmp
for c in ${CFGS:-cfg3}; do for v in ${VARS:-X=0}; do
  rm -rf /tmp/p_fin
  env $v true
  export $v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fin -- python3 /root/repo/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-calibration > /dev/null 2>&1
  unset ${v%%=*}
  echo "== $c $v" >> /root/repo/gpurun_out/fin_stats.txt
  python3 /root/repo/tools/summarize_rocprof.py /tmp/p_fin | grep -E "k_dw_all|k_dw_finalize|k_bottleneck" >> /root/repo/gpurun_out/fin_stats.txt
done; done
