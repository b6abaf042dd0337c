"""durf_amd.datasets (the reference's Waymo scene format + box-pose preprocessing, obbpose_dataset.py:1632-1866) against
the numpy restatement in oracle/durf_data_ref.py, on a small synthetic scene written to disk in the reference's
formats.  CPU: parsing, recentring, box poses, split, tables.  GPU (test_gpu_data.py style): the device batch assembly."""
import os

import numpy as np
import pytest
import torch
from PIL import Image
from scipy.spatial.transform import Rotation as R

from durf_amd import datasets, utils
from oracle import durf_data_ref as D

N_TS, N_CAM, N_OBJ, H, W = 3, 5, 2, 12, 16


def _write_scene(root, seed=0):
    rs = np.random.RandomState(seed)
    n = N_TS * N_CAM
    os.makedirs(os.path.join(root, 'images_4'))
    images = rs.randint(0, 256, (n, H, W, 3)).astype(np.uint8)
    for i in range(n):                                       # names that only sort right naturally: 2.png before 10.png
        Image.fromarray(images[i]).save(os.path.join(root, 'images_4', 'frame_%d.png' % i))
    poses_arr = np.zeros((n, 19))
    for i in range(n):
        rot = R.from_euler('yxz', [0.4 * (i % N_CAM) - 0.8, 0.05 * rs.randn(), 0.02 * rs.randn()]).as_matrix()
        pose = np.concatenate([rot, (rs.randn(3, 1) * 2 + np.array([[3.0 * (i // N_CAM)], [0.0], [0.0]])),
                               np.array([[H * 4.0], [W * 4.0], [60.0]])], 1)          # h, w, focal at full resolution
        poses_arr[i, :15] = pose.reshape(-1)
        poses_arr[i, 15:17] = (0.5, 80.0)
        poses_arr[i, 17:] = (W * 2.0 + rs.randn(), H * 2.0 + rs.randn())
    np.save(os.path.join(root, 'poses_bounds.npy'), poses_arr)
    boxes = {}
    for t in range(1, N_TS + 1):
        for c in range(1, N_OBJ + 1):
            m = np.eye(4)
            m[:3, :3] = R.from_euler('y', 0.3 * t + c).as_matrix()
            m[:3, 3] = rs.randn(3) * 5 + np.array([2.0 * t, 0.0, -10.0 * c])
            boxes['%d_%d_center' % (t, c)] = m
            boxes['%d_%d_ext' % (t, c)] = np.array([4.5, 1.8, 2.0]) + 0.1 * rs.rand(3)
    np.save(os.path.join(root, '3D_boxes.npy'), boxes, allow_pickle=True)
    depth = np.where(rs.rand(n, H, W) < 0.3, rs.rand(n, H, W) * 60 + 1, 0.0).astype(np.float32)
    sky = (rs.rand(n, H, W) < 0.1).astype(np.float32)
    np.savez(os.path.join(root, 'depth_images.npz'), depth)
    np.savez(os.path.join(root, 'sky_masks.npz'), sky)
    np.savez(os.path.join(root, '2D_boxes.npz'), np.zeros((n, H, W), np.float32))
    return images, poses_arr, boxes, depth, sky


def _config(**kw):
    utils.clear_gin()
    return utils.Config(dataset_loader='waymo', batching='timestep', centering=True, factor=4, near=0.0, far=40.0,
                        batch_size=64, **kw)


@pytest.mark.parametrize('random_box,random_yaw', [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize('split', ['train', 'test'])
def test_waymo_preprocessing_matches_the_restatement(tmp_path, split, random_box, random_yaw):
    images, poses_arr, boxes, depth, sky = _write_scene(str(tmp_path))
    config = _config(random_box=random_box, random_yaw=random_yaw)
    ds = datasets.Waymo(split, str(tmp_path), config, device='cpu', seed=5)
    cfg = dict(factor=4, random_box=random_box, random_yaw=random_yaw, box_noise=config.box_noise, yaw_noise=config.yaw_noise)
    ref = D.waymo_preprocess(images.astype(np.float32) / 255., poses_arr.copy(), boxes, depth, sky, cfg, split,
                             np.random.RandomState(5))
    assert sorted(ds.box_pose) == sorted(ref['box_pose'])
    for k in ref['box_pose']:
        np.testing.assert_allclose(np.asarray(ds.box_pose[k]), np.asarray(ref['box_pose'][k]), rtol=1e-12, atol=1e-12, err_msg=k)
    np.testing.assert_array_equal(ds.indices, ref['indices'])
    np.testing.assert_array_equal(ds.timesteps, ref['timesteps'])
    assert ds.total_timesteps == ref['total_timesteps'] == N_TS and ds.n_obj == N_OBJ
    for a, b in ((ds.camtoworlds, ref['camtoworlds']), (ds.focal, ref['focal']), (ds.h, ref['h']), (ds.w, ref['w']),
                 (ds.principal_point, ref['principal_point']), (ds.images, ref['images'])):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
    assert (ds.h == H).all() and (ds.w == W).all() and np.allclose(ds.focal, 15.0)       # floor(h/4), floor(w/4), focal/4
    for a, b in zip(ds.depth, ref['depth']):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(ds.sky, ref['sky']):
        np.testing.assert_array_equal(a, b)
    assert set(np.unique(np.concatenate([s.reshape(-1) for s in ds.sky]))) <= {0.0, np.float32(0.975)}
    # box tables of a training batch / a test case
    tb = ds.tables()
    for t in range(N_TS):
        want = D.waymo_train_tables(ref['box_pose'], t, N_OBJ, N_TS, random_box and split == 'train')
        np.testing.assert_allclose(tb['init'], want['init'], rtol=1e-6, atol=1e-7)
        for k in ('target', 'box', 'ext'):
            np.testing.assert_allclose(tb[k][t], want[k], rtol=1e-6, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(tb['can'], want['can'], rtol=1e-6, atol=1e-7)
    # the rotation vector is the WORLD -> OBJECT rotation of the recentred box, the extents are halves / 5
    k0 = '1_1_center'
    rot_w2o = R.from_rotvec(np.asarray(ds.box_pose[k0])[3:]).as_matrix()
    assert np.allclose(np.linalg.det(rot_w2o), 1.0)
    np.testing.assert_allclose(np.asarray(ds.box_pose['1_1_ext']), np.asarray(boxes['1_1_ext']) / 10.0, rtol=1e-12)


def test_natural_sort_and_errors(tmp_path):
    assert sorted(['f_10.png', 'f_2.png', 'f_1.png'], key=datasets.natural_key) == ['f_1.png', 'f_2.png', 'f_10.png']
    _write_scene(str(tmp_path))
    with pytest.raises(NotImplementedError):
        datasets.Waymo('train', str(tmp_path), _config().__class__(batching='all_images', centering=True, factor=4), device='cpu')
    with pytest.raises(ValueError, match='does not exist'):
        datasets.Waymo('train', str(tmp_path), utils.Config(batching='timestep', centering=True, factor=8), device='cpu')
    with pytest.raises(NotImplementedError):
        datasets.get_dataset('train', str(tmp_path), utils.Config(dataset_loader='carla_dyn'), device='cpu')


@pytest.mark.gpu
def test_waymo_device_batches_match_the_host_gather(cuda, tmp_path):
    """a 'timestep' training batch assembled on the device == the reference's host gather (rays of the timestep's
    concatenated cameras at the sampled indices, obbpose_dataset.py:1551-1557,1582-1583)"""
    images, poses_arr, boxes, depth, sky = _write_scene(str(tmp_path))
    config = _config(random_box=True)
    ds = datasets.Waymo('train', str(tmp_path), config, device=cuda, seed=9)
    ref = D.waymo_preprocess(images.astype(np.float32) / 255., poses_arr.copy(), boxes, depth, sky,
                             dict(factor=4, random_box=True, random_yaw=False, box_noise=config.box_noise, yaw_noise=5.0),
                             'train', np.random.RandomState(9))
    rays = D.generate_rays_multi(ref['h'], ref['w'], ref['focal'], ref['principal_point'], ref['camtoworlds'], 0.0, 40.0)
    rs = np.random.RandomState(9)
    rs.uniform(-config.box_noise, config.box_noise, size=[N_TS * N_OBJ, 3])      # the draw preprocess_boxes consumed
    un = np.unique(ref['timesteps'])
    for _ in range(3):
        b = next(ds)
        time_index = int(rs.randint(0, len(un), ()))
        cams = np.nonzero(ref['timesteps'] == un[time_index])[0]
        n_rays = sum(int(ref['h'][c]) * int(ref['w'][c]) for c in cams)
        idx = rs.randint(0, n_rays, (config.batch_size,))
        assert b['ts'] == time_index
        want_rays, want_px, want_dp, want_sk = D.timestep_batch(rays, list(ref['images']), ref['depth'], ref['sky'], cams, idx)
        for name in ('origins', 'directions', 'viewdirs', 'radii'):
            torch.testing.assert_close(getattr(b['rays'], name).cpu(), torch.tensor(want_rays[name]),
                                       rtol=3e-5 if name == 'radii' else 2e-6, atol=1e-7, msg=lambda m: name + ': ' + m)
        assert torch.equal(b['pixels'].cpu(), torch.tensor(want_px))
        assert torch.equal(b['depth'].cpu(), torch.tensor(want_dp)) and torch.equal(b['sky'].cpu(), torch.tensor(want_sk))
        tb = D.waymo_train_tables(ref['box_pose'], int(un[time_index]) - 1, N_OBJ, N_TS, True)
        np.testing.assert_allclose(b['init'].cpu().numpy(), tb['init'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(b['target'].cpu().numpy(), tb['target'], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(b['ext'].cpu().numpy(), tb['ext'], rtol=1e-6, atol=1e-7)
    # a test case is one full image with the clean poses
    dt = datasets.Waymo('test', str(tmp_path), config, device=cuda, seed=9)
    tc = next(dt)
    assert tc['rays'].origins.shape == (H, W, 3) and tc['pixels'].shape == (H, W, 3)
    assert torch.equal(tc['pixels'].cpu(), torch.tensor(images[10].astype(np.float32) / 255.))
    assert tc['ts'] == 2              # image 10 belongs to timestep 3 -> ts = time_index - 1
