"""Ray generation and 'timestep' batch assembly on the device.

Mirrors what `obbpose_dataset.Waymo._generate_rays_multi` (obbpose_dataset.py:1868-1916) precomputes on the
host for every pixel and `_next_train` (:1551-1587, batching == 'timestep') gathers per step: here the camera
table stays on the host (17 floats per camera), the images / LIDAR depth / sky masks of a timestep stay
resident in HBM, and each step's rays are generated for the sampled pixel indices only -- no per-step
host gather, no H2D copy of rays."""
import ctypes as C

import numpy as np
import torch

from . import _lib, utils
from .ops import _p, _stream


class TimestepData:
    """Device-resident data of one timestep: cameras (host table) + flattened images/depth/sky of its cameras."""

    def __init__(self, camtoworlds, focal, principal_point, h, w, images=None, depth=None, sky=None, device='cuda'):
        n = len(h)
        self.cams = np.zeros((n, 17), np.float32)
        for i in range(n):
            self.cams[i, :12] = np.asarray(camtoworlds[i], np.float32)[:3, :4].reshape(-1)
            self.cams[i, 12:] = (focal[i], principal_point[i][0], principal_point[i][1], h[i], w[i])
        self.n_rays = int(sum(int(a) * int(b) for a, b in zip(h, w)))
        flat = lambda xs, c: torch.cat([torch.as_tensor(np.asarray(x, np.float32)).reshape(-1, c) for x in xs]).to(device)
        self.images = flat(images, np.asarray(images[0]).shape[-1]) if images is not None else None
        self.depth = flat(depth, 1).reshape(-1) if depth is not None else None
        self.sky = flat(sky, 1).reshape(-1) if sky is not None else None
        self.device = torch.device(device)


def generate_batch(ts_data, ray_indices, near, far):
    """-> (BoxRays, pixels[B,C] | None, depth[B,1] | None, sky[B,1] | None) for device int32 `ray_indices`
    (None: the first B = n_rays rays in order, i.e. the full images -- the render_image path)."""
    dev = ts_data.device
    B = ts_data.n_rays if ray_indices is None else ray_indices.shape[0]
    f = lambda c: torch.empty(B, c, device=dev)
    o, d, v, r, lm, nr, fr = f(3), f(3), f(3), f(1), f(1), f(1), f(1)
    img = ts_data.images
    ch = 0 if img is None else img.shape[1]
    px = f(ch) if img is not None else None
    dp = f(1) if ts_data.depth is not None else None
    sk = f(1) if ts_data.sky is not None else None
    cams = ts_data.cams.reshape(-1)
    arr = (C.c_float * cams.size)(*cams.tolist())
    idx = None if ray_indices is None else ray_indices.to(torch.int32).contiguous()
    _lib.check(_lib.lib().durf_gen_batch(_stream(), B, ts_data.cams.shape[0], arr, _p(idx), float(near), float(far),
                                         _p(img), _p(ts_data.depth), _p(ts_data.sky), ch, _p(o), _p(d), _p(v), _p(r),
                                         _p(lm), _p(nr), _p(fr), _p(px), _p(dp), _p(sk)), 'durf_gen_batch')
    return utils.BoxRays(o, d, v, r, lm, nr, fr), px, dp, sk


# ---------------------------------------------------------------------------
# coarse-to-fine multi-resolution schedule (c2f_obb_dataset.py:306-313, 843-891)
# ---------------------------------------------------------------------------
C2F_FACTORS = (16, 12, 8, 4)


def c2f_factor(train_it, c2f_steps):
    """Downsampling factor of training iteration `train_it` (c2f_obb_dataset.py:306-313): 16 up to
    c2f_steps[0], 12 up to c2f_steps[1], 8 up to c2f_steps[2], then 4."""
    if train_it <= c2f_steps[0]:
        return 16
    if train_it <= c2f_steps[1]:
        return 12
    if train_it <= c2f_steps[2]:
        return 8
    return 4


class C2FTimestepData:
    """One timestep at the four C2F resolutions.  `levels`: {factor: dict(h, w, focal, images, depth, sky)};
    the C2F loader centres the principal point at (w/2, h/2) (c2f_obb_dataset.py:861-864)."""

    def __init__(self, camtoworlds, levels, device='cuda'):
        self.data = {}
        for f, lv in levels.items():
            pp = [(w * 0.5, h * 0.5) for h, w in zip(lv['h'], lv['w'])]
            self.data[int(f)] = TimestepData(camtoworlds, lv['focal'], pp, lv['h'], lv['w'], lv.get('images'),
                                             lv.get('depth'), lv.get('sky'), device=device)

    def at(self, train_it, c2f_steps):
        return self.data[c2f_factor(train_it, c2f_steps)]


class IndexUploader:
    """Per-step host -> device upload of the pixel indices a data loader draws with numpy (obbpose_dataset.py:1551-1587 draws
    them on the host) WITHOUT stopping the host: `torch.as_tensor(ndarray, device=...)` is a pageable copy, which HIP runs
    synchronously -- the host then waits for the whole previous step and every launch up to the first long kernel is exposed
    (measured in train_loop at 4096 rays: ~0.2 ms of GPU idle per step, 884 k -> 9xx k rays/s).  Here the draw is staged in
    one of a few pinned buffers and copied with non_blocking=True; a slot is reused only after its copy has run."""

    def __init__(self, device, slots=4):
        self.device = torch.device(device)
        self.slots = [None] * slots
        self.i = 0

    def __call__(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if self.device.type != 'cuda':
            return torch.as_tensor(idx, device=self.device)
        k = self.i % len(self.slots)
        self.i += 1
        slot = self.slots[k]
        if slot is None or slot[0].numel() < idx.size:
            slot = [torch.empty(max(idx.size, 1), dtype=torch.int32).pin_memory(), None]
            self.slots[k] = slot
        if slot[1] is not None:
            slot[1].synchronize()                       # the copy that last read this buffer has run
        slot[0][:idx.size].copy_(torch.from_numpy(idx))
        out = torch.empty(idx.size, dtype=torch.int32, device=self.device)
        out.copy_(slot[0][:idx.size], non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return out
