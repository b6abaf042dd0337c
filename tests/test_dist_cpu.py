"""Data-parallel path on CPU: 2 processes over gloo (127.0.0.1).  The kernels need a GPU, so
each rank computes its shard's gradient with the oracle; what is under test is the product's
plumbing -- init_distributed, shard_batch, ONE all-reduce of the flat gradient buffer with the
mean taken afterwards (train_boxpose.py:253) -- against the oracle's pmap emulation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from durf_amd import synthetic, train_boxpose, utils
    from oracle import durf_ref as R
    from tests import helpers as H
    torch.set_num_threads(2)
    r, w, _ = train_boxpose.init_distributed(backend='gloo')
    assert (r, w) == (rank, world) and dist.get_world_size() == world
    b = synthetic.make_batch(32, 1, seed=8)
    full = {k: (torch.tensor(v, dtype=torch.float64) if isinstance(v, np.ndarray) else v) for k, v in b.items() if k != 'rays'}
    full['rays'] = utils.BoxRays(**{k: torch.tensor(v, dtype=torch.float64) for k, v in b['rays'].items()})
    shard = train_boxpose.shard_batch(full, rank, world)
    assert shard['pixels'].shape[0] == 16
    ob = dict(shard)
    ob['rays'] = R.BoxRays(*shard['rays'])
    params = R.init_params(2, ob['init'], 1, dtype=torch.float64)
    cfg = dict(R.CONFIG_DEFAULTS, randomized=False)
    grads = R.train_step(params, R.new_opt_state(params), ob, cfg, dict(num_samples=8), 5e-4, 3.0, 10.0,
                         ob['init'][0:1])[3]
    flat = torch.cat([g.reshape(-1) for g in grads])          # the flat gradient buffer
    assert train_boxpose._dist() is not None
    dist.all_reduce(flat)                                     # what train_step does ...
    flat /= world                                             # ... followed by inv_world in durf_clip_adam
    if rank == 0:
        torch.save(flat, os.path.join(out_dir, 'dp_grad.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), 'dp_grad.pt'))
    sys.path.insert(0, ROOT)
    from durf_amd import synthetic
    from oracle import durf_ref as R
    from tests import helpers as H
    b = synthetic.make_batch(32, 1, seed=8)
    ob = H.oracle_batch(b, torch.float64)

    def shard(i):
        s = slice(16 * i, 16 * (i + 1))
        out = dict(ob)
        out['rays'] = R.BoxRays(*[r[s] for r in ob['rays']])
        for k in ('pixels', 'depth', 'sky'):
            out[k] = ob[k][s]
        return out
    params = R.init_params(2, ob['init'], 1, dtype=torch.float64)
    cfg = dict(R.CONFIG_DEFAULTS, randomized=False)
    grads = R.train_step(params, R.new_opt_state(params), None, cfg, dict(num_samples=8), 5e-4, 3.0, 10.0,
                         ob['init'][0:1], shards=[shard(0), shard(1)])[3]
    want = torch.cat([g.reshape(-1) for g in grads])
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-15)
