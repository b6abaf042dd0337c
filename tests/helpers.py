"""Shared test plumbing: synthetic batches on either side (oracle on CPU, product on GPU)."""
import numpy as np
import torch

from durf_amd import synthetic, utils
from oracle import durf_ref as R


oracle_batch = R.batch_from_numpy


device_batch = synthetic.device_batch


def oracle_params_from_variables(variables, dt=torch.float32):
    """Variables (flat GPU buffer) -> the oracle's params dict (CPU copies)."""
    p = variables['params']
    out = {'box_centers': p['box_centers'].detach().cpu().to(dt).clone()}
    for name in variables.layout.mlp_names():
        out[name] = [[p[name]['Dense_%d' % i]['kernel'].detach().cpu().to(dt).clone(),
                      p[name]['Dense_%d' % i]['bias'].detach().cpu().to(dt).clone()] for i in range(len(p[name]))]
    return out


def untile(t, rows, nks):
    """bf16 tile layout (include/durf_hip.h) -> [rows, 16*nks] float32 (natural feature order)."""
    t = t.reshape(-1)[: (rows // 32) * nks * 512].reshape(rows // 32, nks, 2, 32, 8)
    return t.permute(0, 3, 1, 2, 4).reshape(rows, nks * 16).float()


def tile(x, nks):
    """[rows, 16*nks] -> bf16 tile layout (rows multiple of 32)."""
    rows = x.shape[0]
    t = x.reshape(rows // 32, 32, nks, 2, 8).permute(0, 2, 3, 1, 4).contiguous()
    return t.to(torch.bfloat16).reshape(rows, nks * 16)


def cperm_cols(nks):
    """column order that maps a C-perm stash tile (mlp_spec.h) back to natural features:
    natural[:, feat] = untiled[:, pos] with pos = 16ks + 8hi + e  <->  feat = 16ks + (e&3) + 8(e>>2) + 4hi."""
    pos_of_feat = np.zeros(nks * 16, dtype=np.int64)
    for ks in range(nks):
        for hi in range(2):
            for e in range(8):
                pos_of_feat[16 * ks + (e & 3) + 8 * (e >> 2) + 4 * hi] = 16 * ks + 8 * hi + e
    return torch.tensor(pos_of_feat)


def extra_fuzz_seeds(kind):
    """DURF_FUZZ_EXTRA=n widens the seeded random sweeps by n more seeds (one-off soak runs; default 0)"""
    import os
    n = int(os.environ.get('DURF_FUZZ_EXTRA', '0'))
    base = 1000 if kind == 'FWD' else 2000
    return list(range(base, base + n))


def slow_fuzz_seeds(seeds):
    """seeds that pass but are the slowest cases of their sweep (kept out of the default run for wall-clock only):
    included whenever DURF_FUZZ_EXTRA is set"""
    import os
    return list(seeds) if int(os.environ.get('DURF_FUZZ_EXTRA', '0')) > 0 else []
