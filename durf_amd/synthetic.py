"""Synthetic random-pose ray batches with the reference's batch schema.

No Waymo/CARLA data exists in the build or GPU environments, so benchmarks and
parity tests use seeded synthetic batches whose per-field distributions follow the
reference's Waymo loader (internal/obbpose_dataset.py:1551-1587 batch assembly,
:1868-1916 ray generation, :1694-1750 box preprocessing, :1836-1853 depth / sky).
Schema: rays.{origins,directions,viewdirs}[B,3],
rays.{radii,lossmult,near,far}[B,1], pixels[B,3], depth[B,1], sky[B,1],
init[T,K,6], target[K,6], ext[K,3], ts (int).  Everything is numpy float32.
"""
import numpy as np

SEED = 20200823  # train_boxpose.py:325


def _rodrigues(r):
    th = np.sqrt(max(float(np.dot(r, r)), 1e-12)) + 1e-12
    Kx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]], dtype=np.float64)
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * (Kx @ Kx)


def _hits(o, d, centers, rots, ext):
    """float64 slab test of rays against K oriented boxes -> [B,K] bool."""
    out = np.zeros((o.shape[0], centers.shape[0]), dtype=bool)
    for k in range(centers.shape[0]):
        R = _rodrigues(rots[k])
        oo = (o - centers[k]) @ R.T
        dd = d @ R.T
        dd = dd / np.linalg.norm(dd, axis=-1, keepdims=True)
        with np.errstate(divide='ignore', invalid='ignore'):
            tmin = (-ext[k] - oo) / dd
            tmax = (ext[k] - oo) / dd
        tn = np.minimum(tmin, tmax).max(-1)
        tf = np.maximum(tmin, tmax).min(-1)
        out[:, k] = (tf > tn) & (tf > 0)
    return out


def make_batch(B, K, T=5, far=40.0, seed=SEED, hit_range=(0.05, 0.15), noise_boxes=0.0,
               img_hw=(320, 480), focal=515.0, allow_multi_hit=False, redraw_noisy_multi_hit=False):
    """One training batch.  `noise_boxes` > 0 emulates Config.random_box (init = target
    + U(-noise, noise) on the centres, configs/waymo.gin:6,8).  Rays that would hit two boxes
    at once are re-drawn unless allow_multi_hit: the reference sums their object-frame origins
    (obbpose_model.py:120-122, "assumes that objects do not occlude each other") and returns
    NaN for them, which turns the whole step into a no-op."""
    rng = np.random.default_rng(seed)
    H, W = img_hw
    cx, cy = W / 2.0, H / 2.0
    n_cam = 5
    cam_yaw = rng.uniform(-0.3, 0.3, n_cam)
    cam = rng.integers(0, n_cam, B)
    u = rng.uniform(0, W - 1, B)
    v = rng.uniform(0, H - 1, B)

    def cam_dirs(uu):
        dc = np.stack([(uu - cx) / focal, -(v - cy) / focal, -np.ones(B)], -1)
        c, s = np.cos(cam_yaw[cam]), np.sin(cam_yaw[cam])
        return np.stack([c * dc[:, 0] + s * dc[:, 2], dc[:, 1], -s * dc[:, 0] + c * dc[:, 2]], -1)

    d = cam_dirs(u)                                   # un-normalised (obbpose_dataset.py:1882-1888)
    dx = np.linalg.norm(cam_dirs(u + 1.0) - d, axis=-1)
    radii = (dx * 2 / np.sqrt(12))[:, None]           # :1895-1901
    o = rng.uniform(-0.5, 0.5, (B, 3))                # poses recentred and /5 (:1697-1700)
    viewdirs = d / np.linalg.norm(d, axis=-1, keepdims=True)

    ext = np.tile(np.array([0.2, 0.17, 0.45]), (K, 1)) * rng.uniform(0.85, 1.15, (K, 3))
    rots = np.zeros((K, 3))
    rots[:, 1] = rng.uniform(-np.pi, np.pi, K)        # yaw about y as a rotvec
    # place box k along an in-FOV direction of camera k % n_cam, then scale the
    # distances until the measured hit fraction lands in hit_range.
    bu = rng.uniform(0.2 * W, 0.8 * W, K)
    bv = rng.uniform(0.35 * H, 0.65 * H, K)
    bd = np.stack([(bu - cx) / focal, -(bv - cy) / focal, -np.ones(K)], -1)
    by = cam_yaw[np.arange(K) % n_cam]
    bdir = np.stack([np.cos(by) * bd[:, 0] + np.sin(by) * bd[:, 2], bd[:, 1],
                     -np.sin(by) * bd[:, 0] + np.cos(by) * bd[:, 2]], -1)
    bdir /= np.linalg.norm(bdir, axis=-1, keepdims=True)
    r0 = rng.uniform(2.0, 6.0, K)
    scale, frac = 1.0, 0.0
    centers = bdir * r0[:, None]
    if K > 0:
        for _ in range(40):
            centers = bdir * (r0 * scale)[:, None]
            frac = float(_hits(o, d, centers, rots, ext).any(-1).mean())
            if frac < hit_range[0]:
                scale *= 0.85
            elif frac > hit_range[1]:
                scale *= 1.2
            else:
                break
    if K > 1 and not allow_multi_hit:
        for _ in range(100):
            bad = np.nonzero(_hits(o, d, centers, rots, ext).sum(-1) > 1)[0]
            if bad.size == 0:
                break
            u[bad] = rng.uniform(0, W - 1, bad.size)
            v[bad] = rng.uniform(0, H - 1, bad.size)
            o[bad] = rng.uniform(-0.5, 0.5, (bad.size, 3))
            d = cam_dirs(u)
            dx = np.linalg.norm(cam_dirs(u + 1.0) - d, axis=-1)
            radii = (dx * 2 / np.sqrt(12))[:, None]
            viewdirs = d / np.linalg.norm(d, axis=-1, keepdims=True)
        frac = float(_hits(o, d, centers, rots, ext).any(-1).mean())
    target_ts = np.concatenate([centers, rots], -1)   # [K,6]
    # T timesteps: the boxes drift a little between timesteps
    drift = rng.normal(0, 0.05, (T, K, 3))
    ts = int(rng.integers(0, T))
    drift[ts] = 0.0
    init = np.tile(target_ts[None], (T, 1, 1))
    init[:, :, :3] += drift
    target = init[ts].copy()
    if noise_boxes > 0:
        init = init.copy()
        init[:, :, :3] += rng.uniform(-noise_boxes, noise_boxes, (T, K, 3))
        if redraw_noisy_multi_hit and K > 1:
            # the model intersects the rays with the NOISY boxes init[ts]: re-draw rays that hit two of those
            # (a separate generator, so batches made without this flag keep their values)
            rng2 = np.random.default_rng(seed + 7919)
            for _ in range(100):
                bad = np.nonzero(_hits(o, d, init[ts, :, :3], init[ts, :, 3:], ext).sum(-1) > 1)[0]
                if bad.size == 0:
                    break
                u[bad] = rng2.uniform(0, W - 1, bad.size)
                v[bad] = rng2.uniform(0, H - 1, bad.size)
                o[bad] = rng2.uniform(-0.5, 0.5, (bad.size, 3))
                d = cam_dirs(u)
                dx = np.linalg.norm(cam_dirs(u + 1.0) - d, axis=-1)
                radii = (dx * 2 / np.sqrt(12))[:, None]
                viewdirs = d / np.linalg.norm(d, axis=-1, keepdims=True)
            frac = float(_hits(o, d, init[ts, :, :3], init[ts, :, 3:], ext).any(-1).mean())

    depth = np.where(rng.uniform(0, 1, B) < 0.3, rng.uniform(0.5, 30.0, B), 0.0)[:, None]
    sky = np.where(rng.uniform(0, 1, B) < 0.1, 0.975, 0.0)[:, None]
    pixels = rng.uniform(0, 1, (B, 3))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    batch = dict(
        rays=dict(origins=f32(o), directions=f32(d), viewdirs=f32(viewdirs), radii=f32(radii),
                  lossmult=np.ones((B, 1), np.float32), near=np.zeros((B, 1), np.float32),
                  far=np.full((B, 1), far, np.float32)),
        pixels=f32(pixels), depth=f32(depth), sky=f32(sky), init=f32(init), target=f32(target),
        ext=f32(ext), ts=ts, hit_fraction=frac)
    return batch


def make_image_rays(H, W, focal=515.0, far=40.0, yaw=0.1, origin=(0.0, 0.0, 0.0)):
    """Full-image test rays [H,W,.] as render_image consumes them (obbpose_model.py:435)."""
    cx, cy = W / 2.0, H / 2.0
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))

    def dirs(u):
        dc = np.stack([(u - cx) / focal, -(vv - cy) / focal, -np.ones_like(u)], -1)
        c, s = np.cos(yaw), np.sin(yaw)
        return np.stack([c * dc[..., 0] + s * dc[..., 2], dc[..., 1],
                         -s * dc[..., 0] + c * dc[..., 2]], -1)
    d = dirs(uu)
    dx = np.linalg.norm(dirs(uu + 1.0) - d, axis=-1)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    one = np.ones((H, W, 1))
    return dict(origins=f32(np.broadcast_to(np.array(origin), (H, W, 3))), directions=f32(d),
                viewdirs=f32(d / np.linalg.norm(d, axis=-1, keepdims=True)),
                radii=f32((dx * 2 / np.sqrt(12))[..., None]), lossmult=f32(one),
                near=f32(0 * one), far=f32(far * one))


def device_batch(b, dev):
    """numpy batch of make_batch -> the product's batch of device tensors (rays as utils.BoxRays)."""
    import torch
    from . import utils
    rays = utils.BoxRays(**{k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in b['rays'].items()})
    out = {k: (torch.tensor(v, dtype=torch.float32, device=dev) if isinstance(v, np.ndarray) else v)
           for k, v in b.items() if k != 'rays'}
    out['rays'] = rays
    return out
