"""Max / mean abs error of the production (bf16, hardware sin/exp) encode path against the oracle, for contracted and
uncontracted coordinates -- run by hand when the encode arithmetic changes:  python tests/encode_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import ops
from oracle import durf_ref as R
from tests import helpers as H
from tests.test_gpu_stages import _setup, _oracle_samples
cuda = torch.device('cuda:0')
for K, contraction in ((0, True), (3, True), (0, False)):
    N = 64
    b, ob, db = _setup(256, K, 11, cuda)
    o_s, d_s, inter, t_vals, samples = _oracle_samples(ob, K, N, b['ts'])
    masks = inter.float().sum(-1)
    bm = (1 - masks)[:, None, None]
    s2 = (bm * samples[0], bm[..., None] * samples[1])
    if contraction: s2 = R.new_space(s2)
    ref = R.integrated_pos_enc(s2, 0, 10).reshape(-1, 60)
    hit = inter.int().to(cuda).contiguous()
    ot, _ = ops.encode_bkgd(t_vals.to(cuda), o_s.to(cuda).contiguous(), d_s.to(cuda).contiguous(),
                            db['rays'].radii.reshape(-1), hit, contraction, tile=True, f32=False)
    ut = H.untile(ot.cpu(), 256 * N, 4)[:, :60]
    ok = (masks <= 1).repeat_interleave(N)
    err = (ut[ok] - ref[ok]).abs()
    print('K=%d contraction=%s: max abs %.2e  mean abs %.2e  (max |mean coord| %.1f)' % (K, contraction, float(err.max()), float(err.mean()), float(s2[0].abs().max())))
