#!/bin/bash
# bench line + kernel table (rocprofv3 --stats) of the non-default workloads -> gpurun_out/other_workloads.txt
out=/root/repo/gpurun_out/other_workloads.txt
rm -f $out
cd /tmp && export TMPDIR=/tmp
for c in cfg2 cfg4 cfg5; do
  echo "=== $c: python3 bench.py --config $c --no-cpu-baseline" >> $out
  python3 /root/repo/bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 >> $out
  rm -rf /tmp/p_ow
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ow -- python3 /root/repo/bench.py --config $c --steps 10 --warmup 12 --no-cpu-baseline --no-calibration > /dev/null 2>&1
  echo "--- rocprofv3 --kernel-trace --stats, 22 steps" >> $out
  python3 /root/repo/tools/summarize_rocprof.py /tmp/p_ow | head -18 >> $out
done
