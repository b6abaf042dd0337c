"""Evaluation metrics of the reference's test loop (train_boxpose.py:398,562-575):
PSNR (internal/math.py:49-51) and SSIM (internal/math.py:66-137) on device images."""
import numpy as np
import torch

from . import _lib
from .math import mse_to_psnr  # noqa: F401  (re-exported: psnr = mse_to_psnr(((pred - gt) ** 2).mean()))
from .ops import _p, _stream


def compute_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """math.compute_ssim for [H,W,C] fp32 device images -> 0-d tensor (or the [H-fs+1, W-fs+1, C] map)."""
    assert img0.shape == img1.shape and img0.dim() == 3
    H, W, Cc = img0.shape
    dev = img0.device
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    f_i = ((np.arange(filter_size) - hw + shift) / filter_sigma) ** 2          # math.py:96-100
    filt = np.exp(-0.5 * f_i)
    filt = (filt / filt.sum()).astype(np.float32)
    fd = torch.tensor(filt, device=dev)
    L = _lib.lib()
    scratch = torch.empty(int(L.durf_ssim_scratch_floats(H, W, Cc, filter_size)), device=dev)
    smap = torch.empty(H - filter_size + 1, W - filter_size + 1, Cc, device=dev) if return_map else None
    out = torch.empty((), device=dev)
    # bind the converted images to locals: a temporary made inside the call expression is freed as soon as its
    # address has been taken, and the caching allocator would hand the second conversion the same block
    a = img0.to(torch.float32).contiguous()
    b = img1.to(torch.float32).contiguous()
    _lib.check(L.durf_ssim(_stream(), H, W, Cc, _p(a), _p(b),
                           float(max_val), filter_size, _p(fd), float(k1), float(k2), _p(smap), _p(scratch), _p(out)),
               'durf_ssim')
    return smap if return_map else out
