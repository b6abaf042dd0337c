#!/bin/bash
# round 6: the ONE evidence set of the round (library version 40): default bench line + per-op table + rocprofv3 stats + the three
# PMC passes (tools/collect_profiles.sh), step timelines of the small workloads, eval, the GPU suite.   -> gpurun_out/r06/
out=gpurun_out/r06; mkdir -p $out
timeout 1500 bash tools/collect_profiles.sh r06 cfg3
for w in "512rays --rays 512" "cfg5 --config cfg5" "cfg1 --config cfg1" "cfg4 --config cfg4" "cfg3"; do
  set -- $w; tag=$1; shift
  timeout 300 bash tools/timeline.sh r06/tl_$tag "$@"
done
timeout 200 python bench.py --mode eval --image-call > $out/eval_image_call.json 2> /dev/null
(timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -8) > $out/pytest_gpu.txt
