#!/bin/bash
# round 6: where the mixed forward's time goes -- the launch with its object items only / its background blocks only (timing
# probes, DURF_MIX_PROBE: results are wrong, durations are what is read):  gpurun_out/<tag>/
tag=${1:-r06_probe}; out=gpurun_out/$tag; mkdir -p $out
timeout 300 bash tools/timeline.sh $tag/tl512 --rays 512
DURF_MIX_PROBE=1 timeout 300 bash tools/timeline.sh $tag/tl512_noobj --rays 512
DURF_MIX_PROBE=2 timeout 300 bash tools/timeline.sh $tag/tl512_nobkgd --rays 512
DURF_OBJ_MIX=0 timeout 300 bash tools/timeline.sh $tag/tl512_nomix --rays 512
DURF_PREP_REDUCE_INKERNEL=1 timeout 300 bash tools/timeline.sh $tag/tlcfg1_inkernel --config cfg1
timeout 300 bash tools/timeline.sh $tag/tlcfg1 --config cfg1
