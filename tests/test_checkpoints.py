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


# ---- the product's writer against an implementation written from the two specifications (oracle/flax_msgpack_ref.py) ----
def _oracle_side(st):
    """the same numbers in the ORACLE's parameter structure ({'box_centers', 'MLP_0': [[kernel, bias] x 12], ...})"""
    from tests import helpers as H
    v = st.variables
    return (H.oracle_params_from_variables(v), H.oracle_params_from_variables(v.like(st.m)),
            H.oracle_params_from_variables(v.like(st.v)))


def _same_tree(a, b, path=''):
    assert type(a) is type(b) or (np.isscalar(a) and np.isscalar(b)), path
    if isinstance(a, dict):
        assert list(a) == list(b), path                      # key ORDER is part of the bytes
        for k in a:
            _same_tree(a[k], b[k], path + '/' + str(k))
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), path
    else:
        assert a == b, path


@pytest.mark.parametrize('K', [0, 2])
def test_writer_agrees_byte_for_byte_with_the_independent_encoder(K):
    from oracle import flax_msgpack_ref as F
    st = _state(K=K, seed=5)
    params, m, v = _oracle_side(st)
    want = F.serialize(F.state_dict(params, m, v, st.step))          # no `msgpack` package, tree built from the oracle's structure
    got = checkpoints.msgpack_serialize(checkpoints.to_state_dict(st))
    assert len(got) == len(want) and got == want
    # each side reads what the other wrote
    _same_tree(F.restore(got), checkpoints.msgpack_restore(want))
    back = checkpoints.from_state_dict(_state(K=K, seed=6), F.restore(got))
    assert back.step == st.step and torch.equal(back.variables.flat, st.variables.flat)
    assert torch.equal(back.m, st.m) and torch.equal(back.v, st.v)


def test_independent_encoder_against_the_msgpack_package_on_every_format_family():
    """the wire-format half of oracle/flax_msgpack_ref.py against msgpack-python on values that reach every format byte
    it emits (fix / 8 / 16 / 32-bit lengths of str, bin, array, map, ext; every integer width; float64; nil / bool)"""
    from oracle import flax_msgpack_ref as F
    rng = np.random.default_rng(0)
    ints = [0, 1, 127, 128, 255, 256, 65535, 65536, 2 ** 32 - 1, 2 ** 32, 2 ** 63, -1, -32, -33, -128, -129, -32768, -32769,
            -2 ** 31, -2 ** 31 - 1, -2 ** 63]
    strs = ['', 'a' * 31, 'b' * 32, 'c' * 255, 'd' * 256, 'e' * 70000, 'grüß']
    bins = [b'', b'x' * 255, b'y' * 256, b'z' * 65535, b'w' * 65536]
    arrays = [[], list(range(15)), list(range(16)), list(range(70000))]
    maps = [{}, {str(i): i for i in range(15)}, {str(i): i for i in range(16)}, {i: [i] for i in range(66000)}]
    values = ints + strs + bins + arrays + maps + [None, True, False, 0.0, -1.5, 1e300, [1, [2, [3, {'k': b'v'}]]]]
    for x in values:
        assert F.serialize(x) == msgpack.packb(x, use_bin_type=True, strict_types=True), repr(x)[:60]
        assert F.restore(F.serialize(x)) == x
    for n in (0, 1, 2, 3, 4, 8, 16, 17, 255, 256, 65535, 65536):             # ext payload sizes through ndarray shapes
        a = rng.standard_normal(n).astype(np.float32)
        assert F.serialize({'a': a}) == checkpoints.msgpack_serialize({'a': a})
        np.testing.assert_array_equal(F.restore(F.serialize({'a': a}))['a'], a)
    for s in (np.float32(2.5), np.int32(-7), np.float64(1e-300)):
        assert F.serialize([s]) == checkpoints.msgpack_serialize([s])
        assert F.restore(F.serialize([s]))[0] == s
