"""Host logic of the thin training driver (durf_amd.train_boxpose.train_loop / main, mirroring the reference's
main(), train_boxpose.py:324-580) and of the sharded render_image, on CPU: the kernels are replaced by a recording
step function; what is under test is schedules, pose feedback, logging cadence, checkpoint cadence and resume."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from durf_amd import checkpoints, math as dmath, obbpose_model, synthetic, train_boxpose, utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Data:
    def __init__(self, T=5, K=2, n=10 ** 9):
        b = synthetic.make_batch(16, K, T=T, seed=3)
        self.b = synthetic.device_batch(b, 'cpu')
        self.i, self.n, self.T = 0, n, T

    def peek(self):
        return dict(self.b, ts=0)

    def __iter__(self):
        return self

    def __next__(self):
        if self.i >= self.n:
            raise StopIteration
        self.i += 1
        return dict(self.b, ts=(3 * self.i) % self.T)


def _fake_step(calls):
    def step(model, config, rng, state, batch, lr, eps, alpha, prev, reduce_stats=True):
        calls.append(dict(step=state.step + 1, lr=lr, eps=eps, alpha=alpha, prev=prev.clone(), ts=batch['ts'],
                          reduce_stats=reduce_stats))
        K = state.variables.layout.K
        pose = torch.full((K, 3), float(state.step + 1))            # recognisable pose estimate of this step
        new = train_boxpose.TrainState(state.variables, state.m, state.v, state.step + 1)
        stats = utils.Stats(loss=torch.tensor(1.0 / (state.step + 1)), psnr=torch.tensor(20.0), grad_norm=torch.tensor(0.5))
        return new, stats, rng + 1, pose
    return step


def _setup(max_steps=25, **kw):
    utils.clear_gin()
    config = utils.Config(max_steps=max_steps, print_every=10, save_every=10, batch_size=16, lr_delay_steps=5,
                          alpha_init=0.0, alpha_final=10.0, alpha_delay_steps=4, alpha_max_steps=20, eps_max_steps=max_steps, **kw)
    data = _Data()
    model, variables = obbpose_model.construct_mipnerf(0, data.peek(), device='cpu')
    return config, data, model, train_boxpose.create_train_state(variables)


def test_train_loop_schedules_feedback_and_cadence(tmp_path):
    config, data, model, state = _setup()
    calls, lines = [], []
    state, hist = train_boxpose.train_loop(model, config, state, data, train_dir=str(tmp_path), step_fn=_fake_step(calls),
                                           log=lines.append)
    assert [c['step'] for c in calls] == list(range(1, 26)) and state.step == 25        # range(init_step, max_steps + 1)
    for c in calls:                                                                      # train_boxpose.py:425-427
        s = c['step']
        assert c['lr'] == dmath.learning_rate_decay(s, config.lr_init, config.lr_final, config.max_steps,
                                                    config.lr_delay_steps, config.lr_delay_mult)
        assert c['eps'] == dmath.learning_rate_decay(s, config.eps_init, config.eps_final, config.eps_max_steps,
                                                     config.eps_delay_steps, config.lr_delay_mult)
        assert c['alpha'] == dmath.freq_alpha_rate(s, 0.0, 10.0, 4, 20)
        assert c['reduce_stats'] == (s % 10 == 0)                                        # stats all-reduce only when logged
    # pose feedback (:429-437): prev = prevs[ts+1] if ts == 0 else prevs[ts-1]; prevs[ts,:,:3] = pose of that step
    prevs = data.peek()['init'].clone()
    for c in calls:
        ts = c['ts']
        nb = ts + 1 if ts == 0 else ts - 1
        assert torch.equal(c['prev'], prevs[nb:nb + 1]), c['step']
        prevs[ts, :, :3] = float(c['step'])
    assert [s for s, _ in hist] == [10, 20] and len(lines) == 2 and '10/25' in lines[0].replace(' ', '')
    # checkpoints: every save_every steps plus the final one (:528-532,577-580)
    assert checkpoints._steps(str(tmp_path)) == [10, 20, 25]


def test_train_loop_resumes_from_checkpoint(tmp_path):
    config, data, model, state = _setup(max_steps=20)
    calls = []
    state, _ = train_boxpose.train_loop(model, config, state, data, train_dir=str(tmp_path), step_fn=_fake_step(calls),
                                        log=lambda s: None)
    assert checkpoints._steps(str(tmp_path)) == [10, 20]
    os.remove(os.path.join(str(tmp_path), 'checkpoint_20'))
    config2, data2, model2, fresh = _setup(max_steps=22)
    calls2 = []
    state2, _ = train_boxpose.train_loop(model2, config2, fresh, data2, train_dir=str(tmp_path), step_fn=_fake_step(calls2),
                                         log=lambda s: None)
    assert [c['step'] for c in calls2] == list(range(11, 23))                            # init_step = restored step + 1
    assert checkpoints._steps(str(tmp_path)) == [10, 20, 22]


def test_main_rejects_indivisible_batch(monkeypatch):
    monkeypatch.setattr(train_boxpose, 'init_distributed', lambda: (0, 3, 0))
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    with pytest.raises(ValueError, match='Batch size must be divisible by the number of devices'):
        train_boxpose.main(['--gin_param', 'Config.batch_size = 512'])


# ---- sharded render_image (obbpose_model.py:421-479 + the all_gather of train_boxpose.py:379) ------------------
def _render_fn(rng, batch):
    r = batch['rays']
    rgb = r.origins * 2.0 + r.directions
    dist_ = (r.origins * r.directions).sum(-1)
    acc = r.radii.reshape(-1) + 1.0
    return [(rgb, dist_, acc)]


def _rays(H, W):
    g = torch.Generator().manual_seed(0)
    f = lambda c: torch.randn(H, W, c, generator=g)
    return utils.BoxRays(f(3), f(3), f(3), f(1), f(1), f(1), f(1))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    train_boxpose.init_distributed(backend='gloo')
    seen = []

    def fn(rng, batch):
        seen.append(batch['rays'].origins.shape[0])
        return _render_fn(rng, batch)
    rgb, d, a = obbpose_model.render_image(fn, _rays(7, 9), None, None, 0, 0, 10.0, chunk=20)    # 63 rays: 20+20+20+3
    torch.save(dict(rgb=rgb, d=d, a=a, seen=seen), os.path.join(out, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_render_image_shards_chunks_across_ranks(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    want = obbpose_model.render_image(_render_fn, _rays(7, 9), None, None, 0, 0, 10.0, chunk=20)
    for r in range(3):
        got = torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r))
        assert got['seen'] == [7, 7, 7, 1]             # chunks of 20 -> padded to 21 = 3 x 7; the last 3 rays -> 3 x 1
        for k, w in zip(('rgb', 'd', 'a'), want):
            assert torch.equal(got[k], w), k           # every rank ends up with the full image
