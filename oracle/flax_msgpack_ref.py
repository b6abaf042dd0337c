"""TEST INFRASTRUCTURE -- an independent restatement of the checkpoint bytes the reference writes.

Only tests/ may import this file.  It restates, byte by byte and without the `msgpack` package the product uses, what
`flax.training.checkpoints.save_checkpoint(train_dir, state, step, keep=)` (train_boxpose.py:404,531,580) puts in a
`checkpoint_<step>` file for the reference's `utils.TrainState(optimizer=flax.optim.Adam(...).create(variables))`
(internal/utils.py:37-39, train_boxpose.py:343-344):

  file      = flax.serialization.msgpack_serialize(flax.serialization.to_state_dict(state))
  state dict (flax.serialization, `flax.optim` era pinned by the reference's requirements: 0.3.x):
      dataclass / struct.dataclass -> {field name: to_state_dict(value)} in field order
          TrainState            -> {'optimizer': ...}
          optim.Optimizer       -> {'target': ..., 'state': ...}      (fields: optimizer_def [not serialised], state, target;
                                                                        Optimizer.__init__ registers ty_to_state_dict that
                                                                        emits 'target' first, then 'state')
          optim.OptimizerState  -> {'step': ..., 'param_states': ...}
          _AdamParamState       -> {'grad_ema': ..., 'grad_sq_ema': ...}
      FrozenDict / dict         -> {key: to_state_dict(value)} in iteration order
      jnp / np ndarray          -> the array (np.asarray)
  msgpack_serialize(tree) = msgpack.packb(tree, default=_msgpack_ext_pack, strict_types=True)
      ndarray    -> ExtType(1, packb((shape tuple, dtype.name, arr.tobytes('C')), use_bin_type=True))
      np scalar  -> ExtType(3, same triple of np.asarray(x))

The msgpack wire format itself is restated from the msgpack specification (format bytes in `_pack`), with the
smallest-encoding rules msgpack-python applies.  PARITY UNPINNED: no file written by a real flax install is available
in this container (flax is not installable here); what this file pins is that the product's writer
(durf_amd/checkpoints.py, built on the `msgpack` package) and an implementation written from the two specifications
agree byte for byte, and that each can read what the other wrote (tests/test_checkpoints.py)."""
import struct

import numpy as np

EXT_NDARRAY, EXT_COMPLEX, EXT_NPSCALAR = 1, 2, 3


# ---------------------------------------------------------------------------
# msgpack wire format (specification: format families nil / bool / int / float / str / bin / array / map / ext)
# ---------------------------------------------------------------------------
def _pack_int(x, out):
    if 0 <= x <= 0x7f:
        out.append(struct.pack('B', x))                        # positive fixint
    elif -32 <= x < 0:
        out.append(struct.pack('b', x))                        # negative fixint
    elif 0 <= x <= 0xff:
        out.append(b'\xcc' + struct.pack('B', x))
    elif 0 <= x <= 0xffff:
        out.append(b'\xcd' + struct.pack('>H', x))
    elif 0 <= x <= 0xffffffff:
        out.append(b'\xce' + struct.pack('>I', x))
    elif 0 <= x:
        out.append(b'\xcf' + struct.pack('>Q', x))
    elif x >= -0x80:
        out.append(b'\xd0' + struct.pack('b', x))
    elif x >= -0x8000:
        out.append(b'\xd1' + struct.pack('>h', x))
    elif x >= -0x80000000:
        out.append(b'\xd2' + struct.pack('>i', x))
    else:
        out.append(b'\xd3' + struct.pack('>q', x))


def _pack_ext(code, data, out):
    n = len(data)
    fix = {1: b'\xd4', 2: b'\xd5', 4: b'\xd6', 8: b'\xd7', 16: b'\xd8'}
    if n in fix:
        out.append(fix[n])
    elif n <= 0xff:
        out.append(b'\xc7' + struct.pack('B', n))
    elif n <= 0xffff:
        out.append(b'\xc8' + struct.pack('>H', n))
    else:
        out.append(b'\xc9' + struct.pack('>I', n))
    out.append(struct.pack('b', code))
    out.append(data)


def _array_triple(a):
    """flax.serialization._ndarray_to_bytes: packb((shape, dtype.name, bytes), use_bin_type=True)"""
    a = np.ascontiguousarray(a)
    out = []
    _pack((tuple(int(s) for s in a.shape), a.dtype.name, a.tobytes('C')), out)
    return b''.join(out)


def _pack(x, out):
    # (numpy types first: np.float64 IS a Python float, and packb(strict_types=True) sends subclasses to the ext hook)
    if isinstance(x, np.ndarray):                              # flax: _msgpack_ext_pack
        _pack_ext(EXT_NDARRAY, _array_triple(x), out)
    elif isinstance(x, np.generic):
        _pack_ext(EXT_NPSCALAR, _array_triple(np.asarray(x)), out)
    elif x is None:
        out.append(b'\xc0')
    elif isinstance(x, bool):
        out.append(b'\xc3' if x else b'\xc2')
    elif isinstance(x, int):
        _pack_int(x, out)
    elif isinstance(x, float):
        out.append(b'\xcb' + struct.pack('>d', x))             # float 64
    elif isinstance(x, str):
        b = x.encode('utf-8')
        n = len(b)
        if n <= 31:
            out.append(struct.pack('B', 0xa0 | n))             # fixstr
        elif n <= 0xff:
            out.append(b'\xd9' + struct.pack('B', n))          # str 8 (use_bin_type)
        elif n <= 0xffff:
            out.append(b'\xda' + struct.pack('>H', n))
        else:
            out.append(b'\xdb' + struct.pack('>I', n))
        out.append(b)
    elif isinstance(x, (bytes, bytearray)):
        n = len(x)
        if n <= 0xff:
            out.append(b'\xc4' + struct.pack('B', n))          # bin 8
        elif n <= 0xffff:
            out.append(b'\xc5' + struct.pack('>H', n))
        else:
            out.append(b'\xc6' + struct.pack('>I', n))
        out.append(bytes(x))
    elif isinstance(x, (list, tuple)):
        n = len(x)
        if n <= 15:
            out.append(struct.pack('B', 0x90 | n))             # fixarray
        elif n <= 0xffff:
            out.append(b'\xdc' + struct.pack('>H', n))
        else:
            out.append(b'\xdd' + struct.pack('>I', n))
        for e in x:
            _pack(e, out)
    elif isinstance(x, dict):
        n = len(x)
        if n <= 15:
            out.append(struct.pack('B', 0x80 | n))             # fixmap
        elif n <= 0xffff:
            out.append(b'\xde' + struct.pack('>H', n))
        else:
            out.append(b'\xdf' + struct.pack('>I', n))
        for k, v in x.items():
            _pack(k, out)
            _pack(v, out)
    else:
        raise TypeError('cannot serialise %r' % type(x))


def serialize(tree):
    """flax.serialization.msgpack_serialize"""
    out = []
    _pack(tree, out)
    return b''.join(out)


class _Reader:
    def __init__(self, buf):
        self.b, self.i = memoryview(buf), 0

    def take(self, n):
        v = self.b[self.i:self.i + n]
        if len(v) != n:
            raise ValueError('truncated msgpack stream')
        self.i += n
        return bytes(v)

    def num(self, fmt):
        return struct.unpack(fmt, self.take(struct.calcsize(fmt)))[0]


def _ext(code, data):
    if code in (EXT_NDARRAY, EXT_NPSCALAR):
        shape, dtype, buf = _unpack(_Reader(data))
        a = np.frombuffer(buf, dtype=np.dtype(dtype)).reshape(shape)
        return a[()] if code == EXT_NPSCALAR else a.copy()
    if code == EXT_COMPLEX:
        re_, im = _unpack(_Reader(data))
        return complex(re_, im)
    raise ValueError('unknown extension type %d' % code)


def _unpack(r):
    t = r.num('B')
    if t <= 0x7f:
        return t
    if t >= 0xe0:
        return t - 0x100
    if 0x80 <= t <= 0x8f:
        return {_unpack(r): _unpack(r) for _ in range(t & 0x0f)}
    if 0x90 <= t <= 0x9f:
        return [_unpack(r) for _ in range(t & 0x0f)]
    if 0xa0 <= t <= 0xbf:
        return r.take(t & 0x1f).decode('utf-8')
    if t == 0xc0:
        return None
    if t in (0xc2, 0xc3):
        return t == 0xc3
    if t in (0xc4, 0xc5, 0xc6):
        return r.take(r.num({0xc4: 'B', 0xc5: '>H', 0xc6: '>I'}[t]))
    if t in (0xc7, 0xc8, 0xc9):
        n = r.num({0xc7: 'B', 0xc8: '>H', 0xc9: '>I'}[t])
        code = r.num('b')
        return _ext(code, r.take(n))
    if t == 0xca:
        return r.num('>f')
    if t == 0xcb:
        return r.num('>d')
    if t in (0xcc, 0xcd, 0xce, 0xcf):
        return r.num({0xcc: 'B', 0xcd: '>H', 0xce: '>I', 0xcf: '>Q'}[t])
    if t in (0xd0, 0xd1, 0xd2, 0xd3):
        return r.num({0xd0: 'b', 0xd1: '>h', 0xd2: '>i', 0xd3: '>q'}[t])
    if t in (0xd4, 0xd5, 0xd6, 0xd7, 0xd8):
        code = r.num('b')
        return _ext(code, r.take({0xd4: 1, 0xd5: 2, 0xd6: 4, 0xd7: 8, 0xd8: 16}[t]))
    if t in (0xd9, 0xda, 0xdb):
        return r.take(r.num({0xd9: 'B', 0xda: '>H', 0xdb: '>I'}[t])).decode('utf-8')
    if t in (0xdc, 0xdd):
        return [_unpack(r) for _ in range(r.num('>H' if t == 0xdc else '>I'))]
    if t in (0xde, 0xdf):
        return {_unpack(r): _unpack(r) for _ in range(r.num('>H' if t == 0xde else '>I'))}
    raise ValueError('reserved msgpack format byte 0x%02x' % t)


def restore(blob):
    """flax.serialization.msgpack_restore"""
    r = _Reader(blob)
    tree = _unpack(r)
    if r.i != len(blob):
        raise ValueError('trailing bytes after the msgpack object')
    return tree


# ---------------------------------------------------------------------------
# the reference's TrainState as a state dict, from the ORACLE's parameter structure (oracle/durf_ref.py: params =
# {'box_centers': [T,K,6], 'MLP_0': [[kernel, bias] x 12], 'BoxMLP_k': ...}; Adam moments in the same structure)
# ---------------------------------------------------------------------------
def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t), dtype=np.float32)


def _mlp_names(params):
    return ['MLP_0'] + sorted([k for k in params if k.startswith('BoxMLP_')], key=lambda s: int(s.split('_')[1]))


def params_tree(params):
    """flax variable tree of the model (obbpose_model.py:35-39,88: box_centers; MLP_0 / BoxMLP_k: Dense_0..11)"""
    tree = {'box_centers': _np(params['box_centers'])}
    for name in _mlp_names(params):
        tree[name] = {'Dense_%d' % i: {'kernel': _np(k), 'bias': _np(b)} for i, (k, b) in enumerate(params[name])}
    return tree


def state_dict(params, m, v, step):
    """to_state_dict(TrainState(optimizer=Adam.create(variables))) with Adam moments m / v structured like params"""
    def moments(name):
        return {'Dense_%d' % i: {'kernel': {'grad_ema': _np(mk), 'grad_sq_ema': _np(vk)},
                                 'bias': {'grad_ema': _np(mb), 'grad_sq_ema': _np(vb)}}
                for i, ((mk, mb), (vk, vb)) in enumerate(zip(m[name], v[name]))}
    ps = {'box_centers': {'grad_ema': _np(m['box_centers']), 'grad_sq_ema': _np(v['box_centers'])}}
    for name in _mlp_names(params):
        ps[name] = moments(name)
    return {'optimizer': {'target': {'params': params_tree(params)},
                          'state': {'step': np.asarray(int(step), np.int32), 'param_states': {'params': ps}}}}
