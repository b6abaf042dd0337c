"""Data-parallel training step with the real kernels: 2 ranks (gloo, both on cuda:0 -- the GPU
box has one device; RCCL needs one GPU per rank) must produce exactly the parameters a single
process gets from averaging the two shards' gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

GIN = ('MipNerfModel.num_samples = 32\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = True\n'
       'MipNerfModel.no_yaw_opt = True\nConfig.randomized = False\nConfig.rand_bkgd = False\n'
       'Config.grad_max_norm = 1.0\nConfig.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n')


def _setup(dev):
    sys.path.insert(0, ROOT)
    from durf_amd import obbpose_model, synthetic, utils
    from tests import helpers as H
    utils.clear_gin()
    utils.parse_gin(GIN)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(512, 1, seed=41)
    db = H.device_batch(b, dev)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=dev)
    return config, model, variables, db


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DURF_DIST_BACKEND='gloo')
    sys.path.insert(0, ROOT)
    from durf_amd import train_boxpose
    r, w, local = train_boxpose.init_distributed()
    dev = torch.device('cuda', local)
    config, model, variables, db = _setup(dev)
    shard = train_boxpose.shard_batch(db, r, w)
    state = train_boxpose.create_train_state(variables)
    state, stats, _, _ = train_boxpose.train_step(model, config, 0, state, shard, 5e-4, 3.0, 10.0, db['init'][0:1])
    torch.cuda.synchronize()
    if r == 0:
        torch.save(dict(flat=state.variables.flat.cpu(), loss=stats.loss.cpu()), os.path.join(out_dir, 'dp.pt'))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_matches_shard_average(cuda, tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), 'dp.pt'))
    # single process: per-shard gradients, mean, then the same clip + Adam
    from durf_amd import ops, train_boxpose
    config, model, variables, db = _setup(cuda)
    grads, losses = [], []
    for r in range(2):
        shard = train_boxpose.shard_batch(db, r, 2)
        g, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, shard, 3.0, 10.0, db['init'][0:1])
        out = train_boxpose._assemble_stats(config, shard, raw, db['init'][0:1], ops.STATS_ASSEMBLE)
        grads.append(g)
        losses.append(out[0].clone())
    gsum = grads[0] + grads[1]
    state = train_boxpose.create_train_state(variables)
    ops.clip_adam(variables.flat, state.m, state.v, gsum, 0.5, float(config.grad_max_val),
                  float(config.grad_max_norm), 5e-4, 0)
    torch.cuda.synchronize()
    torch.testing.assert_close(got['flat'], variables.flat.cpu(), rtol=0, atol=0)     # deterministic kernels
    torch.testing.assert_close(got['loss'], ((losses[0] + losses[1]) / 2).cpu(), rtol=1e-6, atol=0)


def _bench_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DURF_DIST_BACKEND='gloo')
    sys.path.insert(0, ROOT)
    import bench
    from durf_amd import train_boxpose
    r, w, local = train_boxpose.init_distributed()
    dev = torch.device('cuda', local)
    wl = bench.setup_workload('cfg3', dev, r, w, rays=256)           # bench.py's own path: global batch -> this rank's shard
    state, stats_seen = wl['state'], []
    rng = 1000 * r
    for i in range(3):
        state, stats, rng, _ = train_boxpose.train_step(wl['model'], wl['config'], rng, state, wl['batch'], 5e-4, 3.0,
                                                        wl['alpha'], wl['prev'], reduce_stats=(i == 2))
        stats_seen.append(float(stats.loss))
    torch.cuda.synchronize()
    torch.save(dict(flat=state.variables.flat.cpu(), losses=stats_seen, first_pixel=wl['batch']['pixels'][0].cpu()),
               os.path.join(out_dir, 'bench_r%d.pt' % r))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_through_bench_workload_with_distinct_shards(cuda, tmp_path):
    """bench.py's sharding (every rank builds the seeded global batch and keeps its contiguous shard) + the stats
    all-reduce cadence: both ranks end with identical parameters; the rank-local losses of the unreduced steps differ
    (distinct shards), the reduced one is identical on both ranks."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'bench_r0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'bench_r1.pt'))
    assert torch.equal(a['flat'], b['flat']), 'replicas must stay bit-identical after the gradient all-reduce'
    assert not torch.equal(a['first_pixel'], b['first_pixel']), 'the ranks train on distinct shards'
    assert a['losses'][0] != b['losses'][0] and a['losses'][1] != b['losses'][1]      # shard-local scalars
    assert a['losses'][2] == b['losses'][2]                                            # all-reduced when logged


def _rccl_worker(_i, out_dir, force, bucket, pose_opt, instream=0, one_call=False):
    """one process; force: through a world-size-1 `nccl` (= RCCL) group, else the plain single-GPU path; instream: the
    all-reduce issued by the library in the compute stream (csrc/comm.hip); one_call: the step as durf_train_step"""
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'DURF_DIST_BACKEND', 'DURF_RDZV_FILE'):
        os.environ.pop(k, None)
    os.environ['DURF_FORCE_DIST'] = '1' if force else '0'
    os.environ['DURF_BUCKET_ALLREDUCE'] = '1' if bucket else '0'
    if instream == 2:       # the DEFAULT: no switch set -- the self-verified in-stream route must be what an nccl group takes
        os.environ.pop('DURF_INSTREAM_ALLREDUCE', None)
    else:
        os.environ['DURF_INSTREAM_ALLREDUCE'] = '1' if instream else '0'
    sys.path.insert(0, ROOT)
    import bench
    from durf_amd import train_boxpose
    r, w, local = train_boxpose.init_distributed()
    import torch.distributed as dist
    assert dist.is_initialized() == bool(force)
    if force:
        assert dist.get_backend() == 'nccl' and train_boxpose._dist() is not None
    dev = torch.device('cuda', local)
    wl = bench.setup_workload('cfg4' if pose_opt else 'cfg3', dev, 0, 1, rays=256)
    state, losses = wl['state'], []
    rng = 0
    step = train_boxpose.train_step_one_call if one_call else train_boxpose.train_step
    if instream == 2:       # what bench.py and train_loop run: the fast host path, chosen by the init-time self-check
        step = train_boxpose.best_step_fn(wl['model'], state.variables)
        assert step is train_boxpose.train_step_one_call, train_boxpose.COLLECTIVE_INFO
        info = train_boxpose.COLLECTIVE_INFO
        assert info['route'].startswith('in-stream RCCL') and info['ranks_seen'] == 1 and info['world'] == 1, info
    for i in range(3):
        state, stats, rng, _ = step(wl['model'], wl['config'], rng, state, wl['batch'], 5e-4, 3.0,
                                    wl['alpha'], wl['prev'], reduce_stats=(i == 2))
        losses.append(float(stats.loss))
    torch.cuda.synchronize()
    if instream:
        assert train_boxpose._INSTREAM.get('comm') is not None, 'the in-stream communicator was not created'
    elif force:
        assert train_boxpose._INSTREAM.get('comm') is None and 'c10d' in train_boxpose.COLLECTIVE_INFO['route']
    torch.save(dict(flat=state.variables.flat.cpu(), m=state.m.cpu(), losses=losses),
               os.path.join(out_dir, 'rccl_%d_%d_%d_%d.pt' % (force, bucket, instream, one_call)))
    if force:
        dist.barrier()
        train_boxpose.shutdown_instream()
        dist.destroy_process_group()


@pytest.mark.parametrize('pose_opt', [False, True])
def test_train_step_through_rccl_world_size_one(cuda, tmp_path, pose_opt):
    """The data-parallel code path with its PRODUCTION backend: DURF_FORCE_DIST=1 builds a world-size-1 `nccl` group
    (RCCL: the one-GPU box cannot host more ranks), so train_step issues the asynchronous gradient all-reduce, waits
    for it on the compute stream, folds 1 / world into clip + Adam and all-reduces the logged scalars on the logging
    step.  Three steps must leave parameters, Adam moments and losses BIT-identical to the plain path -- also with
    the objects' gradients all-reduced ahead of the background MLP's (DURF_BUCKET_ALLREDUCE=1: two collectives), with the
    all-reduce issued by the library itself in the compute stream (DURF_INSTREAM_ALLREDUCE=1: csrc/comm.hip, its own RCCL
    communicator from a unique id passed through the store), and with the whole step as ONE C call on that communicator
    (durf_train_step with args.comm: what a host that is not Python would run per rank)."""
    for force, bucket, instream, one_call in ((0, 0, 0, False), (1, 0, 0, False), (1, 1, 0, False), (1, 0, 1, False),
                                              (1, 0, 1, True), (1, 0, 2, False)):
        mp.spawn(_rccl_worker, args=(str(tmp_path), force, bucket, pose_opt, instream, one_call), nprocs=1, join=True)
    base = torch.load(os.path.join(str(tmp_path), 'rccl_0_0_0_0.pt'))
    # (rccl_1_0_2_0: round 6's default -- no switch set, the route picked by the self-check at communicator creation)
    for tag in ('rccl_1_0_0_0.pt', 'rccl_1_1_0_0.pt', 'rccl_1_0_1_0.pt', 'rccl_1_0_1_1.pt', 'rccl_1_0_2_0.pt'):
        got = torch.load(os.path.join(str(tmp_path), tag))
        assert torch.equal(got['flat'], base['flat']) and torch.equal(got['m'], base['m']), tag
        assert got['losses'] == base['losses'], tag


def _bucket_worker(rank, world, port, out_dir, bucket, weight_decay=0.0, multi_hit=False, side_streams=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DURF_DIST_BACKEND='gloo', DURF_BUCKET_ALLREDUCE='1' if bucket else '0')
    if side_streams:        # the object backward on the side stream (what batches >= 2048 rays do by themselves): the
        os.environ['DURF_OVERLAP_OBJECTS'] = '2'       # objects' weight-gradient launch has to join it first
    sys.path.insert(0, ROOT)
    import bench
    from durf_amd import train_boxpose
    r, w, local = train_boxpose.init_distributed()
    dev = torch.device('cuda', local)
    wl = bench.setup_workload('cfg3', dev, r, w, rays=256)
    wl['config'].weight_decay_mult = weight_decay
    if multi_hit:                               # the first ray's box hits copied from two rays that hit different boxes:
        from durf_amd import synthetic          # a batch with rays that hit two boxes (poisoned gradient segments)
        bn = synthetic.make_batch(256 * w, wl['K'], far=wl['far'], seed=synthetic.SEED, noise_boxes=0.0, allow_multi_hit=True,
                                  hit_range=(0.2, 0.4))
        full = synthetic.device_batch(bn, dev)
        wl['batch'] = train_boxpose.shard_batch(full, r, w)
        wl['prev'] = full['init'][0:1]
    state, rng = wl['state'], 1000 * r
    for i in range(2):
        state, stats, rng, _ = train_boxpose.train_step(wl['model'], wl['config'], rng, state, wl['batch'], 5e-4, 3.0,
                                                        wl['alpha'], wl['prev'], reduce_stats=False)
    torch.cuda.synchronize()
    if r == 0:
        torch.save(dict(flat=state.variables.flat.cpu(), multi=int(stats.multi_hit_rays)),
                   os.path.join(out_dir, 'bucket_%d.pt' % bucket))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('weight_decay,multi_hit,side_streams',
                         [(0.0, False, False), (1e-2, False, False), (1e-2, True, False), (1e-2, True, True)])
def test_bucketed_allreduce_gives_the_same_parameters(cuda, tmp_path, weight_decay, multi_hit, side_streams):
    """two ranks (gloo): objects' gradient slice all-reduced ahead of [box_centers | MLP_0] == one all-reduce of the
    flat buffer, bit for bit (same element-wise sums, same kernels) -- also with weight decay (the slice must carry its
    own term before it leaves, and nothing may write it afterwards) and with rays that hit two boxes (the poisoned
    segments), and with the object launches on the side stream (round-4 advice: the bucketed branch issued the objects'
    weight gradients on the main stream without joining the side stream's object backward)"""
    for bucket in (0, 1):
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_bucket_worker, args=(2, port, str(tmp_path), bucket, weight_decay, multi_hit, side_streams), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'bucket_0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'bucket_1.pt'))
    assert torch.equal(a['flat'], b['flat'])
    assert (a['multi'] > 0) == multi_hit


def test_bench_gpus_2_end_to_end_on_one_gpu(cuda):
    """`python bench.py --gpus 2` as the driver starts it, with the two ranks sharing the one GPU of this box over gloo
    (DURF_DIST_BACKEND=gloo; RCCL needs a GPU per rank): the launcher, the file-store rendezvous, distinct shards, the
    gradient all-reduce inside the timed steps, max-over-ranks timing, and ONE JSON line -- the last line of stdout --
    whose fields say what ran.  No scaling number is read off this run."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['DURF_DIST_BACKEND'] = 'gloo'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
                        '--rays', '512', '--no-calibration'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    out = json.loads(lines[-1])                                 # the result is the LAST line of stdout
    assert sum(ln.startswith('{"metric"') for ln in lines) == 1  # and rank 0 alone prints it
    assert out['metric'] == 'train_rays_per_sec' and out['unit'] == 'rays/s' and out['higher_is_better'] is True
    assert out['n_gpus'] == 2 and out['steps'] == 4 and out['warmup'] == 2 and out['scaling'] == 'weak'
    c = out['config']
    assert c['rays_per_gpu'] == 512 and c['global_batch'] == 1024 and c['parallelism'] == 'dp2' and c['name'] == 'cfg3'
    col = c['collective']
    assert col['summary'] == 'gloo all-reduce, world size 2' and col['world'] == 2 and col['backend'] == 'gloo'
    assert 'c10d' in col['route'] and 'gloo' in col['why']       # the decision path ran and said why it stayed on c10d
    assert out['cpu_baseline'] is None                          # rank 0 at N = 1 only
    assert abs(out['value'] - 1024 * 4 / (out['ms_per_step'] * 4e-3)) <= 1e-6 * out['value']    # whole-job rays / max-over-ranks time
    r = out['roofline']
    assert r['kernel'].startswith('mlp_') and 0.0 < r['frac'] < 1.0 and out['loss'] == out['loss']


def _oracle_worker(rank, world, port, out_dir, K, B):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DURF_DIST_BACKEND='gloo')
    sys.path.insert(0, ROOT)
    from durf_amd import obbpose_model, synthetic, train_boxpose, utils
    from tests import helpers as H
    import torch.distributed as dist
    r, w, local = train_boxpose.init_distributed()
    dev = torch.device('cuda', local)
    utils.clear_gin()
    utils.parse_gin(GIN)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=43)
    db = H.device_batch(b, dev)
    model, variables = obbpose_model.construct_mipnerf(3, db, device=dev)
    shard = train_boxpose.shard_batch(db, r, w)
    grad, _, _ = train_boxpose.loss_and_grad(model, config, 0, variables, shard, 3.0, 10.0, db['init'][0:1])
    dist.all_reduce(grad)                       # lax.pmean(grad) (train_boxpose.py:253): the sum here, 1 / world below
    grad /= w
    torch.cuda.synchronize()
    if r == 0:
        torch.save(grad.cpu(), os.path.join(out_dir, 'dp_grad.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('K,B', [(2, 256), (0, 192)])
def test_two_rank_gradient_against_the_oracles_pmap_emulation(cuda, tmp_path, K, B):
    """The multi-rank step against the ORACLE, with the real kernels (round-5 review: every GPU dist test was product vs
    product): two ranks (gloo, sharing the box's one GPU) compute their shards' gradients with the HIP path -- per-shard loss
    normalisation, as each pmap replica has it (train_boxpose.py:94-102, 370-374) -- and all-reduce them; the mean must be the
    gradient the oracle's pmap emulation gives for the same two shards (lax.pmean, :253), to the bf16 path's gradient gate."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_oracle_worker, args=(2, port, str(tmp_path), K, B), nprocs=2, join=True)
    got = torch.load(os.path.join(str(tmp_path), 'dp_grad.pt'))
    sys.path.insert(0, ROOT)
    from durf_amd import obbpose_model, synthetic, utils
    from oracle import durf_ref as R
    from tests import helpers as H
    utils.clear_gin()
    utils.parse_gin(GIN)
    b = synthetic.make_batch(B, K, seed=43)
    db = H.device_batch(b, cuda)
    _, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
    params = H.oracle_params_from_variables(variables)
    ob = H.oracle_batch(b)
    n = B // 2

    def shard(i):
        sl = slice(n * i, n * (i + 1))
        out = dict(ob)
        out['rays'] = R.BoxRays(*[x[sl] for x in ob['rays']])
        for k in ('pixels', 'depth', 'sky'):
            out[k] = ob[k][sl]
        return out
    cfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=0.0)
    grads = R.train_step(params, R.new_opt_state(params), None, cfg, dict(num_samples=32), 5e-4, 3.0, 10.0, ob['init'][0:1],
                         shards=[shard(0), shard(1)], mlp_hook=R.mlp_apply_bf16)[3]
    want = torch.cat([g.reshape(-1) for g in grads])
    lay = variables.layout
    rel = lambda a, c: float((a - c).norm() / c.norm())
    for name in lay.mlp_names():
        width, _ = lay.mlp_dims(name)
        sl = slice(lay.mlp_off[name], lay.mlp_off[name] + lay.mlp_size[width])
        assert float(want[sl].norm()) > 0
        assert rel(got[sl].double(), want[sl].double()) < 5e-2, (name, rel(got[sl].double(), want[sl].double()))
