#!/bin/bash
# Round-4 kill experiment for "hand h / dz over mask-compacted" (VERDICT r3 item 1), producer side.
# Variants (tools/experiments/compact_probe.patch on csrc, built with tools/build_variant.sh):
#   cpv  -DPROBE_COMPACT=1                   per 1 KB chunk: 8 x (ballot, mbcnt lo/hi, popcount, running base, address)
#   cpvx -DPROBE_COMPACT=2                   ... + one s_mov_b64 per slot (the exec switch of a predicated store)
#   cps  -DPROBE_STORE8=8                    every 16-byte stash / dz store as 8 two-byte stores (same bytes, same results)
#   cpvs -DPROBE_COMPACT=2 -DPROBE_STORE8=8  both: the instruction stream of a slot-major compaction, dense bytes
#   cph  -DPROBE_COMPACT=2 -DPROBE_STORE8=4  both, HALF the bytes stored (what 50 % sparsity would leave; results wrong)
# Same-box, interleaved: fused forward (inference / training) and backward at 4096 rays x 128 samples.
out=gpurun_out/compact_probe.txt
rm -f $out
TAIL=1 tools/ab_variants.sh $out 3 python tools/time_fwd.py -- main cpv cpvx cps cpvs cph
cat $out
