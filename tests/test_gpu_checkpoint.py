"""Checkpoint round trip on the device (train_boxpose.py:404-406,529-532): train 3 steps, save in the
flax-msgpack layout, restore into a FRESH state (different seed), and the restored state must render the same image bit
for bit and take a 4th step bit-identical to the uninterrupted run.  (The file format itself is unpinned against a real
flax checkpoint -- flax is not installable here; tests/test_checkpoints.py covers the layout.)"""
import pytest
import torch

import bench
from durf_amd import checkpoints, obbpose_model, synthetic, train_boxpose, utils

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['cfg3', 'cfg4'])
def test_save_restore_resumes_bit_identically(cuda, tmp_path, name):
    w = bench.setup_workload(name, cuda, rays=256)
    config, model, batch, prev, alpha = w['config'], w['model'], w['batch'], w['prev'], w['alpha']
    state, rng = w['state'], 0
    for i in range(3):
        state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev)
    path = checkpoints.save_checkpoint(str(tmp_path), state, state.step, keep=1)
    # the FILE the trained device state produced, byte for byte against the independent encoder (oracle/flax_msgpack_ref.py:
    # written from the msgpack and flax-serialization specifications, no `msgpack` package), and read back by it
    from oracle import flax_msgpack_ref as F
    from tests import helpers
    sv, as_oracle = state.variables, helpers.oracle_params_from_variables
    want = F.serialize(F.state_dict(as_oracle(sv), as_oracle(sv.like(state.m)), as_oracle(sv.like(state.v)), state.step))
    with open(path, 'rb') as f:
        written = f.read()
    assert written == want, 'checkpoint bytes differ from the independent encoder'
    tree = F.restore(written)
    assert tree['optimizer']['state']['step'].item() == 3
    # the uninterrupted run: step 4 and a test render
    H, W = 16, 24
    tb = synthetic.device_batch(synthetic.make_batch(H * W, w['K'], seed=5, far=w['far'], allow_multi_hit=True), cuda)
    rays = utils.namedtuple_map(lambda r: r.reshape(H, W, -1), tb['rays'])
    render = lambda st: obbpose_model.render_image(train_boxpose.make_render_fn(model, config, st.variables), rays,
                                                   batch['init'], batch['ext'], batch['ts'], 0, alpha, chunk=128)
    want_img = render(state)
    flat3, m3, v3 = state.variables.flat.clone(), state.m.clone(), state.v.clone()
    want4, wstats, _, _ = train_boxpose.train_step(model, config, rng, state, batch, 5e-4, 3.0, alpha, prev)
    want_flat4 = want4.variables.flat.clone()
    # a fresh state from another seed, then restore
    _, fresh_vars = obbpose_model.construct_mipnerf(12345, batch, device=cuda)
    fresh = train_boxpose.create_train_state(fresh_vars)
    assert not torch.equal(fresh.variables.flat, flat3)
    got = checkpoints.restore_checkpoint(str(tmp_path), fresh)
    assert got.step == 3
    assert torch.equal(got.variables.flat, flat3) and torch.equal(got.m, m3) and torch.equal(got.v, v3)
    got_img = render(got)
    for a, b in zip(got_img, want_img):
        assert torch.equal(a, b), 'restored state renders the same image'
    got4, gstats, _, _ = train_boxpose.train_step(model, config, rng, got, batch, 5e-4, 3.0, alpha, prev)
    assert torch.equal(got4.variables.flat, want_flat4), 'step 4 after the restore == step 4 of the uninterrupted run'
    assert float(gstats.loss) == float(wstats.loss) and got4.step == 4
