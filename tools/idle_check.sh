#!/bin/bash
# GPU idle time between the kernels of a step (rocprofv3 --kernel-trace + tools/gpu_idle_gaps.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p_idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_idle -- python3 /root/repo/bench.py --config ${1:-cfg3} --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-calibration > /root/repo/gpurun_out/idle_bench.json 2>/dev/null
python3 /root/repo/tools/gpu_idle_gaps.py /tmp/p_idle ${STEPS:-20} 3 > /root/repo/gpurun_out/idle_gaps.txt 2>&1
