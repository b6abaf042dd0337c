"""CPU restatement (numpy) of the reference's ray generation, 'timestep' batch gather and SSIM.

TEST INFRASTRUCTURE -- only tests/ import this module; the product path (durf_amd/raygen.py,
durf_amd/metrics.py -> libdurf_hip.so) never does.  PARITY UNPINNED: the reference has no golden
vectors for these functions and JAX is not installable here; every function follows the cited lines.
"""
import numpy as np


def generate_rays_multi(h, w, focal, principal_point, camtoworlds, near, far):
    """obbpose_dataset.py:1868-1916 (Waymo._generate_rays_multi): per-image pinhole rays with
    un-normalised directions, radii from the distance to the next ROW's direction (x 2/sqrt(12)).
    h, w, focal: [n]; principal_point [n,2]; camtoworlds [n,3,4].  Returns a dict of per-image lists."""
    out = dict(origins=[], directions=[], viewdirs=[], radii=[], lossmult=[], near=[], far=[])
    for i in range(len(h)):
        x, y = np.meshgrid(np.arange(int(w[i]), dtype=np.float32), np.arange(int(h[i]), dtype=np.float32),
                           indexing='xy')                                          # :1871-1875
        cam_dirs = np.stack([(x - principal_point[i, 0]) / focal[i],
                             -(y - principal_point[i, 1]) / focal[i], -np.ones_like(x)], axis=-1)     # :1882-1886
        d = np.squeeze((cam_dirs[..., None, :] * camtoworlds[i, :3, :3]).sum(axis=-1))     # :1888-1889
        o = np.broadcast_to(camtoworlds[i, :3, -1], d.shape)                          # :1891
        v = d / np.linalg.norm(d, axis=-1, keepdims=True)                             # :1893
        dx = np.sqrt(np.sum((d[:-1, :, :] - d[1:, :, :]) ** 2, -1))                   # :1896-1898
        dx = np.concatenate([dx, dx[-2:-1, :]], 0)                                    # :1899
        r = dx[..., None] * 2 / np.sqrt(12)                                           # :1902
        ones = np.ones_like(o[..., :1])
        for k, a in (('origins', o), ('directions', d), ('viewdirs', v), ('radii', r), ('lossmult', ones),
                     ('near', near * ones), ('far', far * ones)):
            out[k].append(np.asarray(a, dtype=np.float32))
    return out


def timestep_batch(rays, images, depth, sky, cam_ids, ray_indices):
    """obbpose_dataset.py:1551-1557,1582-1583: the rays/pixels of one timestep are the flattened
    concatenation of its cameras; a batch gathers `ray_indices` from them."""
    flat = {k: np.concatenate([rays[k][c].reshape(-1, rays[k][c].shape[-1]) for c in cam_ids], 0) for k in rays}
    img = np.concatenate([images[c].reshape(-1, images[c].shape[-1]) for c in cam_ids], 0)
    dep = np.concatenate([depth[c].reshape(-1, 1) for c in cam_ids], 0)
    sk = np.concatenate([sky[c].reshape(-1, 1) for c in cam_ids], 0)
    return ({k: v[ray_indices] for k, v in flat.items()}, img[ray_indices], dep[ray_indices], sk[ray_indices])


def compute_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    """internal/math.py:66-137 for [H,W,C] images (separable 'valid' Gaussian blur, x then y)."""
    from scipy.signal import convolve2d
    img0, img1 = np.asarray(img0, np.float64), np.asarray(img1, np.float64)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    f_i = ((np.arange(filter_size) - hw + shift) / filter_sigma) ** 2
    filt = np.exp(-0.5 * f_i)
    filt /= np.sum(filt)

    def filt_fn(z):                      # filt_fn1(filt_fn2(z)): blur along W, then along H (:112-113)
        return np.stack([convolve2d(convolve2d(z[..., c], filt[None, :], mode='valid'), filt[:, None], mode='valid')
                         for c in range(z.shape[-1])], -1)
    mu0, mu1 = filt_fn(img0), filt_fn(img1)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    sigma00 = np.maximum(0., filt_fn(img0 ** 2) - mu00)
    sigma11 = np.maximum(0., filt_fn(img1 ** 2) - mu11)
    sigma01 = filt_fn(img0 * img1) - mu01
    sigma01 = np.sign(sigma01) * np.minimum(np.sqrt(sigma00 * sigma11), np.abs(sigma01))
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    ssim_map = ((2 * mu01 + c1) * (2 * sigma01 + c2)) / ((mu00 + mu11 + c1) * (sigma00 + sigma11 + c2))
    return ssim_map if return_map else ssim_map.mean()
