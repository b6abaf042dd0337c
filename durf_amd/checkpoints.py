"""Checkpoint import/export in the layout `flax.training.checkpoints` writes for the reference's
TrainState (train_boxpose.py:404,530-532,580), so a durf checkpoint can be loaded
into this build and vice versa.

flax (0.2.2 - 0.5.x, the `flax.optim` era pinned by requirements_jax.txt) is not installed here, so the
format is restated from its published serializer and is UNPINNED against a real checkpoint:
  * file  <dir>/checkpoint_<step>  = msgpack of `flax.serialization.to_state_dict(state)`;
  * state dict of utils.TrainState(optimizer=flax.optim.Optimizer):
        {'optimizer': {'target': {'params': {box_centers, MLP_0/Dense_i/{kernel,bias}, BoxMLP_k/...}},
                       'state': {'step': 0-d int32 ndarray (flax.optim keeps OptimizerState.step as an array),
                                 'param_states': {'params': {<same tree>: {'grad_ema', 'grad_sq_ema'}}}}}}
  * every ndarray is msgpack ExtType(1, packb((shape, dtype.name, bytes))), numpy scalars ExtType(3, same).
Only tests/ exercise it today; train scripts call save_checkpoint / restore_checkpoint like the
reference does.
"""
import os
import re

import msgpack
import numpy as np
import torch

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
PREFIX = 'checkpoint_'


def _pack_array(a):
    a = np.ascontiguousarray(a)
    return msgpack.packb((list(a.shape), a.dtype.name, a.tobytes('C')), use_bin_type=True)


def _ext_pack(x):
    if isinstance(x, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _pack_array(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _pack_array(np.asarray(x)))
    if isinstance(x, torch.Tensor):
        return msgpack.ExtType(_EXT_NDARRAY, _pack_array(x.detach().cpu().numpy()))
    raise TypeError('cannot serialise %r' % type(x))


def _ext_unpack(code, data):
    if code in (_EXT_NDARRAY, _EXT_NPSCALAR):
        shape, dtype, buf = msgpack.unpackb(data, raw=False)
        a = np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
        return a[()] if code == _EXT_NPSCALAR else a.copy()
    if code == _EXT_COMPLEX:
        re_, im = msgpack.unpackb(data, raw=False)
        return complex(re_, im)
    return msgpack.ExtType(code, data)


def msgpack_serialize(tree):
    return msgpack.packb(tree, default=_ext_pack, strict_types=True, use_bin_type=True)


def msgpack_restore(blob):
    return msgpack.unpackb(blob, ext_hook=_ext_unpack, raw=False, strict_map_key=False)


def _tree_like_params(variables, flat):
    """flax param tree (numpy leaves) of a flat buffer laid out like `variables`."""
    v = variables.like(flat.detach().cpu().contiguous())

    def conv(d):
        return {k: (conv(x) if isinstance(x, dict) else x.numpy().copy()) for k, x in d.items()}
    return conv(v['params'])


def to_state_dict(state):
    """TrainState (train_boxpose.TrainState) -> the reference's state dict."""
    params = _tree_like_params(state.variables, state.variables.flat)
    m = _tree_like_params(state.variables, state.m)
    v = _tree_like_params(state.variables, state.v)

    def zip_states(a, b):
        if isinstance(a, dict):
            return {k: zip_states(a[k], b[k]) for k in a}
        return {'grad_ema': a, 'grad_sq_ema': b}
    return {'optimizer': {'target': {'params': params},
                          'state': {'step': np.asarray(int(state.step), np.int32), 'param_states': {'params': zip_states(m, v)}}}}


def _fill(variables_like, tree, what):
    """copy a flax param tree into a Variables view (shape-checked)"""
    def rec(dst, src, path):
        for k, x in dst.items():
            if k not in src:
                raise ValueError('checkpoint is missing %s/%s' % (path, k))
            if isinstance(x, dict):
                rec(x, src[k], path + '/' + k)
            else:
                a = src[k] if what is None else src[k][what]
                a = np.asarray(a)
                if tuple(a.shape) != tuple(x.shape):
                    raise ValueError('shape of %s/%s: checkpoint %s, model %s' % (path, k, a.shape, tuple(x.shape)))
                x.copy_(torch.from_numpy(a.astype(np.float32)))
    rec(variables_like['params'], tree, 'params')


def from_state_dict(state, sd):
    """fill `state` (same layout) from a reference state dict; returns state"""
    opt = sd['optimizer']
    dev = state.variables.flat.device
    for buf, tree, what in ((state.variables.flat, opt['target']['params'], None),
                            (state.m, opt['state']['param_states']['params'], 'grad_ema'),
                            (state.v, opt['state']['param_states']['params'], 'grad_sq_ema')):
        cpu = torch.zeros(buf.shape, dtype=torch.float32)
        _fill(state.variables.like(cpu), tree, what)
        buf.copy_(cpu.to(dev))
    state.step = int(np.asarray(opt['state']['step']).reshape(-1)[0])
    return state


def _steps(ckpt_dir):
    out = []
    if os.path.isdir(ckpt_dir):
        for f in os.listdir(ckpt_dir):
            m = re.fullmatch(re.escape(PREFIX) + r'(\d+)', f)
            if m:
                out.append(int(m.group(1)))
    return sorted(out)


def latest_checkpoint(ckpt_dir):
    s = _steps(ckpt_dir)
    return os.path.join(ckpt_dir, PREFIX + str(s[-1])) if s else None


def save_checkpoint(ckpt_dir, state, step, keep=1):
    """flax.training.checkpoints.save_checkpoint(ckpt_dir, state, step, keep=) (train_boxpose.py:531,580):
    atomic write of checkpoint_<step>, then only the newest `keep` files stay."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, PREFIX + str(int(step)))
    tmp = path + '.tmp'
    with open(tmp, 'wb') as f:
        f.write(msgpack_serialize(to_state_dict(state)))
    os.replace(tmp, path)
    steps = _steps(ckpt_dir)
    for s in steps[:-keep] if keep > 0 else []:
        os.remove(os.path.join(ckpt_dir, PREFIX + str(s)))
    return path


def restore_checkpoint(ckpt_dir, state):
    """flax.training.checkpoints.restore_checkpoint(ckpt_dir, target) (train_boxpose.py:404): newest
    checkpoint_<step> in a directory (or the file itself); `state` unchanged when there is none."""
    path = ckpt_dir if os.path.isfile(ckpt_dir) else latest_checkpoint(ckpt_dir)
    if path is None:
        return state
    with open(path, 'rb') as f:
        sd = msgpack_restore(f.read())
    return from_state_dict(state, sd)
