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


# ---------------------------------------------------------------------------------------------------------------
# Waymo scene preprocessing (obbpose_dataset.py:1632-1866) restated in the reference's own order of operations
# ---------------------------------------------------------------------------------------------------------------
def waymo_preprocess(images, poses_arr, masks3d, depth_list, sky_mask, config, split, rng):
    """What Waymo._load_renderings leaves in `self` (centering=True path).  images [n,H,W,3]; poses_arr [n,19];
    masks3d: the dict of 3D_boxes.npy; depth_list / sky_mask: [n,H,W].  rng: object with .uniform like np.random."""
    from scipy.spatial.transform import Rotation as R
    factor = config['factor'] if config['factor'] > 0 else 1
    poses = poses_arr[:, :15].reshape([-1, 3, 5]).transpose([1, 2, 0])               # :1662
    bds = poses_arr[:, 15:17].transpose([1, 0])
    principal_point = poses_arr[:, 17:]
    masks3d = dict(masks3d)
    box_pose, box_ext = [], []                                                       # :1677-1684
    for key in masks3d:
        if 'center' in key:
            box_pose.append(masks3d[key])
        elif 'ext' in key:
            box_ext.append(masks3d[key])
    box_pose, box_ext = np.array(box_pose), np.array(box_ext)
    poses[:2, 4, :] = np.floor(poses[:2, 4, :] * 1. / factor)                        # :1687-1689
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    principal_point = principal_point * 1. / factor
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    # _recenter_poses (:709-729)
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    z = poses[:, :3, 2].sum(0)
    vec2 = z / np.linalg.norm(z)
    vec2 = vec2 / np.linalg.norm(vec2)
    up = poses[:, :3, 1].sum(0)
    vec0 = np.cross(up, vec2)
    vec0 = vec0 / np.linalg.norm(vec0)
    vec1 = np.cross(vec2, vec0)
    vec1 = vec1 / np.linalg.norm(vec1)
    c2w = np.concatenate([np.stack([vec0, vec1, vec2, center], 1), hwf], 1)
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = np.concatenate([c2w[:3, :4], bottom], -2)
    p4 = np.concatenate([poses[:, :3, :4], np.tile(bottom[None], [poses.shape[0], 1, 1])], -2)
    p4 = np.linalg.inv(c2w) @ p4
    poses_ = poses.copy()
    poses_[:, :3, :4] = p4[:, :3, :4]
    poses = poses_
    poses[:, :3, 3] /= 5.0                                                           # :1700
    if config['random_box']:                                                         # :1701-1712
        random_box = box_pose.copy()
        random_box[:, :3, 3] += rng.uniform(-config['box_noise'], config['box_noise'], size=[box_pose.shape[0], 3])
        random_box = np.linalg.inv(c2w) @ random_box
        random_box[:, :3, 3] /= 5.0
    box_pose = np.linalg.inv(c2w) @ box_pose                                         # :1715-1716
    box_pose[:, :3, 3] /= 5.0
    yaw = np.array(R.from_matrix(np.linalg.inv(box_pose[:, :3, :3])).as_rotvec())    # :1718-1719
    if config['random_yaw'] and config['random_box']:
        rand_yaw = yaw.copy()
        rand_yaw += (rng.uniform(-config['yaw_noise'], config['yaw_noise'], size=yaw.shape) * (np.pi / 180.0))
        rand_pose = np.concatenate([random_box[:, :3, 3], rand_yaw], axis=-1)
    elif config['random_box']:
        rand_pose = np.concatenate([random_box[:, :3, 3], yaw], axis=-1)
    else:
        rand_pose = np.concatenate([box_pose[:, :3, 3], yaw], axis=-1)
    obbpose = np.concatenate([box_pose[:, :3, 3], yaw], axis=-1)
    box_ext = box_ext / (5.0 * 2.0)                                                  # :1734
    bpose = [k for k in masks3d if 'center' in k]
    for i, key in enumerate(bpose):                                                  # :1739-1750
        ts, car, _ = key.split('_')
        masks3d[key] = obbpose[i]
        masks3d[ts + '_' + car + '_off'] = rand_pose[i]
        masks3d[ts + '_' + car + '_ext'] = box_ext[i]
    n = len(images)
    timesteps = []                                                                   # :1786-1790
    for i in range(1, int(n / 5) + 1):
        timesteps.append(np.array([i, i, i, i, i]))
    timesteps = np.asarray(timesteps).reshape(-1)
    i_test = np.array([10, 12])                                                      # :1804
    i_train = np.array([i for i in np.arange(n) if i not in i_test])
    indices = i_train if split == 'train' else i_test
    depth = [np.array(d, dtype=np.float32) for d in np.asarray(depth_list)[indices]]
    for elem in depth:                                                               # :1836-1837
        elem[elem > 0.0] = elem[elem > 0.0] / 5.0
    sky = [np.array(s, dtype=np.float32) for s in np.asarray(sky_mask)[indices]]
    for elem in sky:                                                                 # :1852-1853
        elem[elem > 0.0] = 0.975
    poses = poses[indices]
    return dict(box_pose=masks3d, poses=poses, camtoworlds=poses[:, :3, :4], focal=poses[:, -1, -1], h=poses[:, 0, -1],
                w=poses[:, 1, -1], principal_point=principal_point[indices], timesteps=timesteps[indices],
                total_timesteps=int(timesteps[-1]), images=np.asarray(images)[indices], depth=depth, sky=sky,
                indices=indices)


def waymo_train_tables(box_pose, time_index, n_obj, n_ts, random_box):
    """The box part of a 'timestep' training batch (obbpose_dataset.py:1558-1580); time_index 0-based."""
    cars = np.arange(1, n_obj + 1)
    suffix = '_off' if random_box else '_center'
    batch_init = []
    for i in range(n_ts):
        batch_init.append(np.array([np.concatenate(box_pose[str(i + 1) + '_' + str(c) + suffix][:, None], axis=0)
                                    for c in cars]).reshape(-1, 6))
    batch_init = np.array(batch_init).reshape(n_ts, -1, 6)
    g = lambda t, suf, d: np.array([np.concatenate(box_pose[str(t) + '_' + str(c) + suf][..., None], axis=0)
                                    for c in cars]).reshape(-1, d)
    return dict(init=batch_init, target=g(time_index + 1, '_center', 6), box=g(time_index + 1, '_off', 6),
                can=g(1, '_off', 6), ext=g(time_index + 1, '_ext', 3))
