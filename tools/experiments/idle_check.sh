#!/bin/bash
# GPU idle time between the kernels of a step + one step's timeline (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_idle -- python3 /root/repo/bench.py --config ${1:-cfg3} --steps ${STEPS:-20} --warmup 12 --no-cpu-baseline --no-calibration > /root/repo/gpurun_out/idle_bench.json 2>/dev/null
python3 /root/repo/tools/gpu_idle_gaps.py /tmp/p_idle ${STEPS:-20} 12 > /root/repo/gpurun_out/idle_gaps.txt 2>&1
python3 /root/repo/tools/step_timeline.py /tmp/p_idle 20 > /root/repo/gpurun_out/step_timeline.txt 2>&1
