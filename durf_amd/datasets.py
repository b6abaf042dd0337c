"""Loaders for the reference's on-disk scene formats and the box-pose preprocessing of its Waymo loader
(internal/obbpose_dataset.py:1632-1866), feeding the device-side batch assembly (raygen.py).

A scene directory holds (obbpose_dataset.py:1634-1793):
  images[_<factor>]/*.{jpg,JPG,png}   natural-sorted, 5 cameras per timestep (FRONT, FRONT_LEFT, SIDE_LEFT, FRONT_RIGHT,
                                      SIDE_RIGHT)
  poses_bounds.npy   [n, 19] = 15 (3x5 pose | h, w, focal column) + 2 bounds + 2 principal point       (:1662-1664)
  3D_boxes.npy       pickled dict '<ts>_<car>_center' -> 4x4 box-to-world, '<ts>_<car>_ext' -> full extents [3]
  depth_images.npz / sky_masks.npz / 2D_boxes.npz     'arr_0' = one [H, W] map per image

Host side (numpy / scipy, like the reference): parse, recentre + /5, box poses -> (position, rotation vector of the
world->object rotation), optional box / yaw noise, half extents, train / test split, the per-timestep init / target /
ext tables.  Device side: each timestep's images / depth / sky live in HBM as one `raygen.TimestepData`; a training
batch samples one timestep and `batch_size` pixel indices of it (batching == 'timestep', :1551-1587) and the rays are
generated on the device (`durf_gen_batch`) -- nothing per-ray crosses PCIe per step.  Other batching modes and the
CARLA loaders are not built."""
import os
import re

import numpy as np
import torch

from . import raygen, utils

TEST_IMAGES = (10, 12)              # i_test of the reference's Waymo loader (:1804)
CAMS_PER_TIMESTEP = 5               # :1786-1789


def natural_key(s):
    """natsorted(): digit runs compare as numbers"""
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', s)]


def _normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """obbpose_dataset.py:731-738"""
    vec2 = _normalize(z)
    vec0 = _normalize(np.cross(up, vec2))
    vec1 = _normalize(np.cross(vec2, vec0))
    return np.stack([vec0, vec1, vec2, pos], 1)


def poses_avg(poses):
    """:722-729"""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    vec2 = _normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(vec2, up, center), hwf], 1)


def recenter_poses(poses):
    """:709-720 -> (recentred poses [n,3,5], c2w 4x4 of the average pose)"""
    poses_ = poses.copy()
    bottom = np.reshape([0, 0, 0, 1.], [1, 4])
    c2w = poses_avg(poses)
    c2w = np.concatenate([c2w[:3, :4], bottom], -2)
    bottom = np.tile(np.reshape(bottom, [1, 1, 4]), [poses.shape[0], 1, 1])
    p = np.concatenate([poses[:, :3, :4], bottom], -2)
    p = np.linalg.inv(c2w) @ p
    poses_[:, :3, :4] = p[:, :3, :4]
    return poses_, c2w


def preprocess_boxes(masks3d, c2w, config, rng=np.random):
    """obbpose_dataset.py:1677-1750 (config.centering): box-to-world 4x4 -> recentred, /5, rotation vector of the
    world->object rotation; optional position noise (random_box) / yaw noise (random_yaw) on the '_off' poses; half
    extents.  Returns the reference's `self.box_pose` dict: '<ts>_<car>_center' [6] (clean), '_off' [6] (what `init`
    is made of when random_box), '_ext' [3]; and rel_pose '<ts>_<car>_rel' 4x4."""
    from scipy.spatial.transform import Rotation as R
    keys_c = [k for k in masks3d if 'center' in k]
    box_pose = np.array([masks3d[k] for k in keys_c], dtype=np.float64)
    box_ext = np.array([masks3d[k] for k in masks3d if 'ext' in k], dtype=np.float64)
    random_box = None
    if config.random_box:                                                           # :1704-1712
        random_box = box_pose.copy()
        random_box[:, :3, 3] += rng.uniform(-config.box_noise, config.box_noise, size=[box_pose.shape[0], 3])
        random_box = np.linalg.inv(c2w) @ random_box
        random_box[:, :3, 3] /= 5.0
    box_pose = np.linalg.inv(c2w) @ box_pose                                        # :1715-1716
    box_pose[:, :3, 3] /= 5.0
    yaw = np.array(R.from_matrix(np.linalg.inv(box_pose[:, :3, :3])).as_rotvec())   # :1718-1719
    if config.random_yaw and config.random_box:                                     # :1722-1731
        rand_yaw = yaw + rng.uniform(-config.yaw_noise, config.yaw_noise, size=yaw.shape) * (np.pi / 180.0)
        rand_pose = np.concatenate([random_box[:, :3, 3], rand_yaw], axis=-1)
    elif config.random_box:
        rand_pose = np.concatenate([random_box[:, :3, 3], yaw], axis=-1)
    else:
        rand_pose = np.concatenate([box_pose[:, :3, 3], yaw], axis=-1)
    obbpose = np.concatenate([box_pose[:, :3, 3], yaw], axis=-1)                    # :1733
    box_ext = box_ext / (5.0 * 2.0)                                                 # :1734: full extent -> half, /5
    out, rel = dict(masks3d), {}
    can_pose = None
    for i, key in enumerate(keys_c):                                                # :1737-1750
        ts, car, _ = key.split('_')
        if '1_' in key and 'center' in key:
            can_pose = box_pose[i]
            rel[ts + '_' + car + '_rel'] = np.eye(4)
        else:
            rel[ts + '_' + car + '_rel'] = np.matmul(can_pose, np.linalg.inv(box_pose[i]))
        out[key] = obbpose[i]
        out[ts + '_' + car + '_off'] = rand_pose[i]
        out[ts + '_' + car + '_ext'] = box_ext[i]
    return out, rel


def load_scene(data_dir, config):
    """Everything `_load_renderings` reads (:1632-1793), before the split: dict of numpy arrays."""
    from PIL import Image
    factor = config.factor if config.factor > 0 else 1
    imgdir = os.path.join(data_dir, 'images' + ('_{}'.format(config.factor) if config.factor > 0 else ''))
    if not os.path.isdir(imgdir):
        raise ValueError('Image folder {} does not exist.'.format(imgdir))           # :1643
    files = sorted((f for f in os.listdir(imgdir) if f.endswith(('JPG', 'jpg', 'png'))), key=natural_key)
    images = np.array([np.array(Image.open(os.path.join(imgdir, f)), dtype=np.float32)[:, :, :3] / 255. for f in files])
    poses_arr = np.load(os.path.join(data_dir, 'poses_bounds.npy'))
    poses = poses_arr[:, :15].reshape([-1, 3, 5]).transpose([1, 2, 0])               # :1662
    bds = poses_arr[:, 15:17].transpose([1, 0])
    principal_point = poses_arr[:, 17:]
    if poses.shape[-1] != len(images):
        raise RuntimeError('Mismatch between imgs {} and poses {}'.format(len(images), poses.shape[-1]))
    masks3d = np.load(os.path.join(data_dir, '3D_boxes.npy'), allow_pickle=True).item()
    poses[:2, 4, :] = np.floor(poses[:2, 4, :] * 1. / factor)                        # :1690-1692
    poses[2, 4, :] = poses[2, 4, :] * 1. / factor
    principal_point = principal_point * 1. / factor
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)

    def npz(name, what):
        a = np.load(os.path.join(data_dir, name), allow_pickle=True)['arr_0']
        if len(a) != len(images):
            raise RuntimeError('Mismatch between imgs {} and {} {}'.format(len(images), what, len(a)))
        return a
    return dict(images=images, poses=poses, bds=bds, principal_point=principal_point, masks3d=masks3d,
                depth=npz('depth_images.npz', 'depth'), sky=npz('sky_masks.npz', 'depth'),
                masks2d=npz('2D_boxes.npz', 'masks2d'))


class Waymo:
    """obbpose_dataset.Waymo with batching == 'timestep' as an iterator of device batches (train) or full-image
    test cases (test).  The per-step batch is assembled on the device; `peek()` / `next()` like the reference's
    Dataset thread (:80-107), without the thread: there is nothing left for it to prefetch."""

    def __init__(self, split, data_dir, config, device='cuda', rank=0, world=1, seed=20201473):
        if config.batching != 'timestep':
            raise NotImplementedError('%s batching strategy is not implemented.' % config.batching)     # :1584-1586
        self.split, self.config, self.device, self.rank, self.world = split, config, torch.device(device), rank, world
        self.rng = np.random.RandomState(seed)                                      # np.random.seed(20201473), train_boxpose.py:328
        sc = load_scene(data_dir, config)
        poses, masks3d = sc['poses'], sc['masks3d']
        if config.centering:                                                        # :1694-1700
            poses, c2w = recenter_poses(poses)
            poses[:, :3, 3] /= 5.0
            self.box_pose, self.rel_poses = preprocess_boxes(masks3d, c2w, config, self.rng)
            self.random_box = bool(config.random_box)
        else:
            raise NotImplementedError('Waymo scenes are loaded with Config.centering = True (box poses are only '
                                      'converted to (position, rotvec) on that path, obbpose_dataset.py:1694-1750)')
        n = len(sc['images'])
        timesteps = np.repeat(np.arange(1, n // CAMS_PER_TIMESTEP + 1), CAMS_PER_TIMESTEP)          # :1786-1790
        self.total_timesteps = int(timesteps[-1])
        i_test = np.array(TEST_IMAGES)
        i_train = np.array([i for i in np.arange(n) if i not in i_test])
        idx = i_train if split == 'train' else (np.sort(np.concatenate([i_train, i_test])) if split == 'render' else i_test)
        self.indices, self.timesteps = idx, timesteps[idx]
        last_ts = list(self.box_pose.keys())[-1].split('_')[0]                      # :1828-1830
        self.n_obj = int(len(self.box_pose) / 3 / int(last_ts))
        self.cars = np.arange(1, self.n_obj + 1)
        depth = [np.where(d > 0.0, d / 5.0, d).astype(np.float32) for d in sc['depth'][idx]]        # :1836-1837
        sky = [np.where(s > 0.0, 0.975, s).astype(np.float32) for s in sc['sky'][idx]]              # :1852-1853
        images, pp = sc['images'][idx], sc['principal_point'][idx]
        poses = poses[idx]
        self.camtoworlds, self.focal = poses[:, :3, :4], poses[:, -1, -1]           # :1857-1862
        self.h, self.w = poses[:, 0, -1], poses[:, 1, -1]
        self.principal_point = pp
        self.near, self.far = config.near, config.far
        # one TimestepData per distinct timestep of the split (flatten_time, :1489-1507)
        self.ts_values = np.unique(self.timesteps)
        self.ts_data = []
        for t in self.ts_values:
            m = np.nonzero(self.timesteps == t)[0]
            self.ts_data.append(raygen.TimestepData(self.camtoworlds[m], self.focal[m], pp[m], self.h[m].astype(int),
                                                    self.w[m].astype(int), [images[i] for i in m],
                                                    [depth[i] for i in m], [sky[i] for i in m], device=self.device))
        self.images, self.depth, self.sky = images, depth, sky
        self.n_examples = len(images)
        self.it = 0
        self._peek = None
        self._tables = None

    # -- box tables ------------------------------------------------------------------------------
    def _pose_rows(self, ts, suffix):
        return np.array([np.asarray(self.box_pose['%d_%d_%s' % (ts, c, suffix)]).reshape(-1) for c in self.cars])

    def tables(self):
        """init [T,K,6] ('_off' when random_box else '_center', :1561-1572), per-timestep target / box / ext, can."""
        if self._tables is None:
            T = self.total_timesteps
            suf = 'off' if (self.random_box and self.split == 'train') else 'center'
            if self.split != 'train':
                suf = 'center'                                                       # __getitem__ / _next_test use the clean poses (:1600-1606)
            init = np.stack([self._pose_rows(t + 1, suf) for t in range(T)]).reshape(T, -1, 6)
            self._tables = dict(
                init=init.astype(np.float32),
                target=[self._pose_rows(t + 1, 'center').astype(np.float32) for t in range(T)],
                box=[self._pose_rows(t + 1, 'off').astype(np.float32) for t in range(T)],
                ext=[self._pose_rows(t + 1, 'ext').reshape(-1, 3).astype(np.float32) for t in range(T)],
                can=self._pose_rows(1, 'off').astype(np.float32))
        return self._tables

    # -- iteration -------------------------------------------------------------------------------
    def _dev(self, a):
        """device copy of a (static) table, made once: a per-step pageable upload would stop the host (raygen.IndexUploader)"""
        cache = self.__dict__.setdefault('_dev_cache', {})
        if id(a) not in cache:
            cache[id(a)] = (a, torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.device))
        return cache[id(a)][1]

    def _next_train(self):
        """:1551-1587: one timestep, batch_size pixel indices of its concatenated cameras (the SAME draw on every
        rank; each rank keeps its contiguous shard, as utils.shard does)."""
        tb = self.tables()
        time_index = int(self.rng.randint(0, len(self.ts_values), ()))
        td = self.ts_data[time_index]
        ray_indices = self.rng.randint(0, td.n_rays, (self.config.batch_size,))
        per = self.config.batch_size // self.world
        if getattr(self, '_upload', None) is None:
            self._upload = raygen.IndexUploader(self.device)
        mine = self._upload(ray_indices[self.rank * per:(self.rank + 1) * per])
        rays, px, dp, sk = raygen.generate_batch(td, mine, self.near, self.far)
        ts = int(self.ts_values[time_index]) - 1
        return dict(rays=rays, pixels=px, depth=dp, sky=sk, init=self._dev(tb['init']), ext=self._dev(tb['ext'][ts]),
                    ts=time_index, target=self._dev(tb['target'][ts]), box=self._dev(tb['box'][ts]), can=self._dev(tb['can']))

    def _next_test(self):
        """__getitem__ of the test split (:1597-1630): one full image, clean box poses, ts = time_index - 1."""
        tb = self.tables()
        idx = self.it
        self.it = (self.it + 1) % self.n_examples
        t = int(self.timesteps[idx])
        m = np.nonzero(self.timesteps == t)[0]
        td = self.ts_data[int(np.nonzero(self.ts_values == t)[0][0])]
        rays, px, dp, sk = raygen.generate_batch(td, None, self.near, self.far)
        H, W = int(self.h[idx]), int(self.w[idx])
        off = int(sum(int(self.h[i]) * int(self.w[i]) for i in m if i < idx))
        img = lambda x: x[off:off + H * W].reshape(H, W, -1)
        return dict(rays=utils.namedtuple_map(img, rays), pixels=img(px), depth=img(dp), sky=img(sk),
                    init=self._dev(tb['init']), ext=self._dev(tb['ext'][t - 1]), ts=t - 1,
                    target=self._dev(tb['target'][t - 1]), box=self._dev(tb['box'][t - 1]), can=self._dev(tb['can']))

    def peek(self):
        if self._peek is None:
            self._peek = self._next_train() if self.split == 'train' else self._next_test()
        return self._peek

    def __iter__(self):
        return self

    def __next__(self):
        if self._peek is not None:
            b, self._peek = self._peek, None
            return b
        return self._next_train() if self.split == 'train' else self._next_test()


def get_dataset(split, data_dir, config, device='cuda', rank=0, world=1):
    """obbpose_dataset.get_dataset (dataset_dict[config.dataset_loader]); only the Waymo loader is built."""
    if config.dataset_loader != 'waymo':
        raise NotImplementedError('dataset_loader %r: only the Waymo scene format is built' % config.dataset_loader)
    return Waymo(split, data_dir, config, device=device, rank=rank, world=world)
