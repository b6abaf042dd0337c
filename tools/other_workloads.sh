#!/bin/bash
# bench line + kernel table (rocprofv3 --stats) of the non-default workloads and modes -> gpurun_out/other_workloads.txt
#   cfg1 (BASELINE configs[0]: N=64, K=0, the gin-literal 512-ray batch, with its own CPU baseline), cfg3 at the gin-literal
#   512 rays, cfg2 / cfg4 / cfg5, the eval (render_image) mode, and the forced world-size-1 RCCL path
out=/root/repo/gpurun_out/other_workloads.txt
rm -f $out
cd /tmp && export TMPDIR=/tmp
run() {   # label, bench args...
  local label=$1; shift
  echo "=== $label: python3 bench.py $*" >> $out
  python3 /root/repo/bench.py "$@" 2>/dev/null | tail -1 >> $out
  rm -rf /tmp/p_ow
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ow -- python3 /root/repo/bench.py "$@" --steps 10 --warmup 12 --no-cpu-baseline --no-calibration --no-workloads > /dev/null 2>&1
  echo "--- rocprofv3 --kernel-trace --stats, 22 steps (eval: 12 images)" >> $out
  python3 /root/repo/tools/summarize_rocprof.py /tmp/p_ow | head -${ROWS:-18} >> $out
}
run cfg1 --config cfg1
run "cfg3 at the gin-literal batch (512 rays)" --config cfg3 --rays 512
run cfg2 --config cfg2 --no-cpu-baseline
ROWS=26 run cfg4 --config cfg4 --no-cpu-baseline
run cfg5 --config cfg5 --no-cpu-baseline
run eval --mode eval --no-cpu-baseline
ROWS=34 run "cfg3 through a world-size-1 RCCL group" --force-dist --no-cpu-baseline
