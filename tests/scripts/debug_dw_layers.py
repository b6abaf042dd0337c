"""Debug aid: per-layer relative errors of the HIP weight gradients vs the bf16-emulating oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops
from oracle import durf_ref as R
from tests import helpers as H

cuda = torch.device('cuda:0')
width, in_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256, None
in_dim = 60 if width == 256 else 63
N, Bn = 32, int(sys.argv[2]) if len(sys.argv) > 2 else 48
rows = N * Bn
g = torch.Generator().manual_seed(3)
cfg = R.MLP_BKGD if width == 256 else R.MLP_BOX
shapes = R.mlp_layer_shapes(in_dim, 27, cfg)
params, flat = [], []
for fi, fo in shapes:
    lim = (6.0 / (fi + fo)) ** 0.5
    k = ((torch.rand(fi, fo, generator=g) * 2 - 1) * lim).requires_grad_(True)
    bb = ((torch.rand(fo, generator=g) - 0.5) * 0.2).requires_grad_(True)
    params.append([k, bb]); flat += [k.detach().reshape(-1), bb.detach()]
flat = torch.cat(flat).to(cuda)
x = torch.randn(Bn, N, in_dim, generator=g).to(torch.bfloat16).float()
cond = torch.randn(Bn, 27, generator=g).to(torch.bfloat16).float()
draw = torch.randn(rows, 4, generator=g) * 0.1
xp = torch.zeros(rows, 64); xp[:, :in_dim] = x.reshape(rows, in_dim)
enc_tile = H.tile(xp, 4).to(cuda)
view = torch.zeros(Bn, 32); view[:, :27] = cond
view = view.to(torch.bfloat16).to(cuda)
wf, wb = ops.pack_weights(width, in_dim, flat, want_bwd=True)
stash = torch.zeros(ops.mlp_stash_bytes(width, rows), dtype=torch.uint8, device=cuda)
mask = torch.zeros(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=cuda)
ops.mlp_fwd(width, rows, N, enc_tile, view, wf, stash=stash, relu_mask=mask)
dz, dz_out = ops.mlp_bwd(width, rows, N, draw.to(cuda), wb, mask)
part, bpart = ops.dw_buffers(width, cuda)
view_tile = ops.expand_view(rows, N, view)
ops.mlp_dw(width, rows, N, [enc_tile], [view_tile], [stash], [dz], [dz_out], part, bpart)
grad = torch.zeros_like(flat)
ops.mlp_dw_finalize(width, in_dim, rows, N, 1, part, bpart, grad, flat)
grad = grad.cpu()
rgb, dens = R.mlp_apply_bf16(params, x, cond, cfg)
out = torch.cat([rgb.reshape(rows, 3), dens.reshape(rows, 1)], -1)
(out * draw).sum().backward()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
off = 0
for li, (k, bb) in enumerate(params):
    gk = grad[off:off + k.numel()].reshape(k.shape); off += k.numel()
    gb = grad[off:off + bb.numel()]; off += bb.numel()
    print('Dense_%d  dW rel %.4f   db rel %.4f' % (li, rel(gk, k.grad), rel(gb, bb.grad)))
