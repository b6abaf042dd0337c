"""k_bkgd_hit_rays against a float64 torch evaluation of the same two layers, for hit counts around the block size"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from durf_amd import ops, obbpose_model, synthetic, utils
from tests import helpers as H
dev = torch.device('cuda:0')
utils.clear_gin()
b = synthetic.make_batch(64, 2, seed=3)
db = H.device_batch(b, dev)
model, variables = obbpose_model.construct_mipnerf(5, db, device=dev)
P = variables.mlp_flat('MLP_0')
g = torch.Generator().manual_seed(1)
P.copy_((torch.randn(P.shape, generator=g) * 0.05).to(dev))
L = ops._lib.lib()
off10, off11 = int(L.durf_mlp_layer_offset(256, 60, 10, 0)), int(L.durf_mlp_layer_offset(256, 60, 11, 0))
W10 = P[off10:off10 + 283 * 128].view(283, 128).double(); b10 = P[off10 + 283 * 128: off10 + 284 * 128].double()
W11 = P[off11:off11 + 128 * 3].view(128, 3).double(); b11 = P[off11 + 384: off11 + 387].double()
trunk = ops.bkgd_const_trunk_f32(P)
worst = 0.0
for B in (1, 3, 4, 5, 17, 100, 1023, 1024, 4096):
    for n in sorted(set([0, 1, B // 3, B - 1, B])):
        view27 = torch.randn(B, 27, device=dev)
        idx = torch.randperm(B, device=dev).int()
        count = torch.tensor([n], dtype=torch.int32, device=dev)
        out = ops.bkgd_hit_rays_f32(B, view27, P, idx, count, trunk=trunk)
        torch.cuda.synchronize()
        if n == 0:
            continue
        x = torch.cat([trunk[:256].double().expand(n, 256), view27[idx[:n].long()].double()], 1)
        h = torch.relu(x @ W10 + b10)
        want = torch.cat([h @ W11 + b11, trunk[256].double().expand(n, 1)], 1)
        err = float((out[:n].double() - want).abs().max() / want.abs().max())
        worst = max(worst, err)
        print('B %5d  n %5d  max rel err %.2e' % (B, n, err))
print('worst', worst)
assert worst < 1e-5
