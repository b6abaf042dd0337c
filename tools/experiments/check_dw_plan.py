import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops
from tests import helpers as H
cuda = torch.device('cuda:0')
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 128
rows, W, IN = B * N, 256, 60
KW = 16
torch.manual_seed(0)
flat = (torch.rand(ops.mlp_param_count(W, IN), device=cuda) - 0.5) * 0.2
wf, wb = ops.pack_weights(W, IN, flat, want_bwd=True)
enc = (torch.randn(rows * 64, device=cuda) * 0.5).to(torch.bfloat16)
view = (torch.randn(B * 32, device=cuda) * 0.5).to(torch.bfloat16)
stash = torch.empty(ops.mlp_stash_bytes(W, rows), dtype=torch.uint8, device=cuda)
mask = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=cuda)
ops.mlp_fwd(W, rows, N, enc, view, wf, stash=stash, relu_mask=mask)
draws = [torch.randn(rows, 4, device=cuda) * 1e-2 for _ in range(2)]
dzs = [ops.mlp_bwd(W, rows, N, d, wb, mask) for d in draws]
view_tile = ops.expand_view(rows, N, view)
part, bpart = ops.dw_buffers(W, cuda)
ops.dispatch_reset()
ops.mlp_dw(W, rows, N, [enc] * 2, [view_tile] * 2, [stash] * 2, [d[0] for d in dzs], [d[1] for d in dzs], part, bpart)
print(ops.dispatch_seen())
grad = torch.zeros_like(flat)
ops.mlp_dw_finalize(W, IN, rows, N, 2, part, bpart, grad, flat)

def untile(t, nks, off_ks, perm):
    nt = rows // 32
    v = t.view(torch.bfloat16).reshape(-1)[off_ks * nt * 512:(off_ks + nks) * nt * 512].reshape(nt, nks, 2, 32, 8)
    x = v.permute(0, 3, 1, 2, 4).reshape(rows, nks * 16).double()
    return x[:, H.cperm_cols(nks).to(x.device)] if perm else x
enc60 = untile(enc, 4, 0, False)[:, :60]
view27 = untile(view_tile, 2, 0, False)[:, :27]
h = [untile(stash, KW, j * KW, True) for j in range(8)]
hv = untile(stash, 8, 9 * KW, True)
def kern(l):
    fi, fo = [(60,256),(256,256),(256,256),(256,256),(256,256),(316,256),(256,256),(256,256),(256,1),(256,256),(283,128),(128,3)][l]
    o = ops.mlp_layer_offset(W, IN, l, False); ob = ops.mlp_layer_offset(W, IN, l, True)
    return (o, fi, fo, ob)
K9 = flat[kern(9)[0]:kern(9)[0] + 256 * 256].reshape(256, 256).double(); b9 = flat[kern(9)[3]:kern(9)[3] + 256].double()
K10 = flat[kern(10)[0]:kern(10)[0] + 283 * 128].reshape(283, 128).double()
bneck = h[7] @ K9 + b9
want = {l: [0, 0] for l in range(12)}
for dz, dzo in dzs:
    dZ = {j: untile(dz, KW, j * KW, True) for j in range(8)}
    dZ10 = untile(dz, 8, 9 * KW, True)
    dzo_ = untile(dzo, 1, 0, False)
    X = {0: enc60, 5: torch.cat([h[4], enc60], 1)}
    for l in (1, 2, 3, 4, 6, 7):
        X[l] = h[l - 1]
    for l in range(8):
        want[l][0] = want[l][0] + X[l].T @ dZ[l]; want[l][1] = want[l][1] + dZ[l].sum(0)
    want[8][0] = want[8][0] + h[7].T @ dzo_[:, 3:4]; want[8][1] = want[8][1] + dzo_[:, 3:4].sum(0)
    dB = dZ10 @ K10[:256].T
    want[9][0] = want[9][0] + h[7].T @ dB; want[9][1] = want[9][1] + dB.sum(0)
    want[10][0] = want[10][0] + torch.cat([bneck, view27], 1).T @ dZ10; want[10][1] = want[10][1] + dZ10.sum(0)
    want[11][0] = want[11][0] + hv.T @ dzo_[:, :3]; want[11][1] = want[11][1] + dzo_[:, :3].sum(0)
rel = lambda a, b: float((a - b).norm() / b.norm())
for l in range(12):
    o, fi, fo, ob = kern(l)
    gk = grad[o:o + fi * fo].reshape(fi, fo).double(); gb = grad[ob:ob + fo].double()
    print('Dense_%d  dK rel %.3e  db rel %.3e   |dK| %.3e' % (l, rel(gk, want[l][0]), rel(gb, want[l][1]), float(want[l][0].norm())))
    if l == 10:
        print('   bottleneck rows %.3e   view rows %.3e' % (rel(gk[:256], want[l][0][:256]), rel(gk[256:], want[l][0][256:])))
