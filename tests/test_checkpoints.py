"""flax-msgpack checkpoint layout (durf_amd/checkpoints.py): tree names, ndarray extension encoding,
round trip, keep=, and resume step -- CPU only."""
import os

import msgpack
import numpy as np
import pytest
import torch

from durf_amd import checkpoints, obbpose_model, train_boxpose


def _state(K=2, seed=0):
    lay = obbpose_model.ParamLayout(5, K)
    g = torch.Generator().manual_seed(seed)
    variables = obbpose_model.Variables(torch.randn(lay.total, generator=g), lay)
    st = train_boxpose.create_train_state(variables)
    st.m.copy_(torch.randn(lay.total, generator=g))
    st.v.copy_(torch.rand(lay.total, generator=g))
    st.step = 1234
    return st


def test_state_dict_tree_matches_reference_layout():
    sd = checkpoints.to_state_dict(_state())
    assert list(sd) == ['optimizer']
    opt = sd['optimizer']
    assert set(opt) == {'target', 'state'} and set(opt['state']) == {'step', 'param_states'}
    p = opt['target']['params']
    assert set(p) == {'box_centers', 'MLP_0', 'BoxMLP_0', 'BoxMLP_1'}
    assert p['box_centers'].shape == (5, 2, 6)
    assert set(p['MLP_0']) == {'Dense_%d' % i for i in range(12)}
    assert p['MLP_0']['Dense_5']['kernel'].shape == (316, 256) and p['MLP_0']['Dense_10']['kernel'].shape == (283, 128)
    assert p['BoxMLP_1']['Dense_0']['kernel'].shape == (63, 128) and p['BoxMLP_1']['Dense_11']['bias'].shape == (3,)
    ps = opt['state']['param_states']['params']['MLP_0']['Dense_3']['kernel']
    assert set(ps) == {'grad_ema', 'grad_sq_ema'} and ps['grad_ema'].shape == (256, 256)
    assert opt['state']['step'] == 1234


def test_ndarray_extension_encoding():
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    blob = checkpoints.msgpack_serialize({'x': a, 's': np.float32(2.5), 'n': 7})
    raw = msgpack.unpackb(blob, raw=False)                      # without the ext hook: ExtType(1, ...)
    assert isinstance(raw['x'], msgpack.ExtType) and raw['x'].code == 1 and raw['s'].code == 3
    shape, dtype, buf = msgpack.unpackb(raw['x'].data, raw=False)
    assert shape == [2, 3] and dtype == 'float32' and buf == a.tobytes()
    back = checkpoints.msgpack_restore(blob)
    np.testing.assert_array_equal(back['x'], a)
    assert back['s'] == np.float32(2.5) and back['n'] == 7


def test_save_restore_round_trip_and_keep(tmp_path):
    st = _state(seed=1)
    d = str(tmp_path / 'ckpt')
    for step in (100, 200, 300):
        st.step = step
        checkpoints.save_checkpoint(d, st, step, keep=2)
    assert sorted(os.listdir(d)) == ['checkpoint_200', 'checkpoint_300']
    fresh = _state(seed=9)
    fresh = checkpoints.restore_checkpoint(d, fresh)
    assert fresh.step == 300
    for a, b in ((fresh.variables.flat, st.variables.flat), (fresh.m, st.m), (fresh.v, st.v)):
        assert torch.equal(a, b)
    # empty directory: target returned unchanged (flax semantics), init_step = step + 1 = 1
    other = _state(seed=3)
    other.step = 0
    assert checkpoints.restore_checkpoint(str(tmp_path / 'none'), other).step == 0


def test_restore_rejects_other_topology(tmp_path):
    d = str(tmp_path / 'c')
    checkpoints.save_checkpoint(d, _state(K=1), 1)
    with pytest.raises(ValueError):
        checkpoints.restore_checkpoint(d, _state(K=2))
