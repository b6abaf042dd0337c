"""Run only the fused MLP forward (inference + training variants) at cfg2 size -- for PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops
dev = torch.device('cuda:0')
rows, N = 4096 * 128, 128
width, in_dim = 256, 60
flat = (torch.rand(ops.mlp_param_count(width, in_dim), device=dev) - 0.5) * 0.1
wf, wb = ops.pack_weights(width, in_dim, flat, want_bwd=True)
enc = (torch.randn(rows, 64, device=dev)).to(torch.bfloat16)
view = torch.randn(4096, 32, device=dev).to(torch.bfloat16)
raw = torch.empty(rows, 4, device=dev)
stash = torch.empty(ops.mlp_stash_bytes(width, rows), dtype=torch.uint8, device=dev)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev)
draw = torch.randn(rows, 4, device=dev) * 0.01
for _ in range(3):
    ops.mlp_fwd(width, rows, N, enc, view, wf, raw=raw)
    ops.mlp_fwd(width, rows, N, enc, view, wf, raw=raw, stash=stash, relu_mask=mask)
    ops.mlp_bwd(width, rows, N, draw, wb, mask)
torch.cuda.synchronize()
