#!/bin/bash
# k_dw_finalize / k_bottleneck_grads durations under the finalize and split-count variants (rocprofv3 --stats)
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-"1 0" "0 0"}; do
  set -- $v
  export DURF_MERGE_FINALIZE=$1
  if [ "$2" = 0 ]; then unset DURF_DW_WGS; else export DURF_DW_WGS=$2; fi
  rm -rf /tmp/p_fin
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fin -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-calibration > /dev/null 2>&1
  echo "== merge=$1 wgs=$2" >> /root/repo/gpurun_out/fin_stats.txt
  python3 /root/repo/tools/summarize_rocprof.py /tmp/p_fin | grep -E "k_dw_all|k_dw_finalize|k_bottleneck" >> /root/repo/gpurun_out/fin_stats.txt
done
