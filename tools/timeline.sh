#!/bin/bash
# one step's kernel timeline + idle gaps of a bench workload:  tools/timeline.sh <tag> <bench args...>   -> gpurun_out/<tag>_*.txt
# (traced step 21 = timed step 9: one WITHOUT the roofline HIP events, which bench.py records on every 4th step of a 20-step run)
tag=$1; shift
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- python3 /root/repo/bench.py "$@" --steps 20 --warmup 12 --no-cpu-baseline --no-calibration --no-workloads > $out/${tag}_bench.json 2>/dev/null
python3 /root/repo/tools/gpu_idle_gaps.py /tmp/p_tl 20 12 > $out/${tag}_idle_gaps.txt 2>&1
python3 /root/repo/tools/step_timeline.py /tmp/p_tl 21 > $out/${tag}_step_timeline.txt 2>&1
