#!/bin/bash
# Round-5 evidence set (one box, one gpurun call): the default bench line as the driver runs it (timed), the profile set
# tools/collect_profiles.sh collects, the other workloads with their kernel tables, step timelines, the GPU test suite.
cd /root/repo
mkdir -p gpurun_out/r05
( time python3 bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err ) 2> gpurun_out/r05/bench_default_wallclock.txt
bash tools/collect_profiles.sh r05 cfg3
bash tools/other_workloads.sh; cp gpurun_out/other_workloads.txt gpurun_out/r05/other_workloads.txt
bash tools/timeline.sh r05/cfg3 ; bash tools/timeline.sh r05/cfg3_512rays --rays 512; bash tools/timeline.sh r05/cfg1 --config cfg1
bash tools/timeline.sh r05/cfg4 --config cfg4; bash tools/timeline.sh r05/cfg5 --config cfg5
python3 bench.py --precision f32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05/bench_f32.json 2>/dev/null
python3 -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/r05/pytest_gpu.txt
