#!/bin/bash
# Per-kernel averages of a step through the Python-issued launches and through the one C call (same kernels, same order):
#   tools/experiments/onecall_vs_python_trace.sh "512 3 128"  ->  gpurun_out/onecall_trace.txt
cd /tmp && export TMPDIR=/tmp
args=${1:-"512 3 128"}
out=/root/repo/gpurun_out/onecall_trace.txt
: > $out
for m in python c; do
  rm -rf /tmp/oc_$m
  ONLY=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/oc_$m -- python3 /root/repo/tools/time_train_call.py $args >> $out 2>/dev/null
  echo "=== ONLY=$m" >> $out
  python3 /root/repo/tools/summarize_rocprof.py /tmp/oc_$m | head -32 >> $out
done
