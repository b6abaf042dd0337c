"""Stage-level operators: torch tensors in, torch tensors out, every one a call through the
C ABI of libdurf_hip.so on the current HIP stream.  PyTorch is used for device memory and
streams only.  Reference lines each op replaces are cited in include/durf_hip.h."""
import ctypes as C
import math
import os

import torch

from . import _lib

ENC_DIM = 64
VIEW_DIM = 32


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-op timing with HIP events recorded on the launch stream (bench.py's live
# roofline measurement).  TIMERS = None disables it (default: zero overhead).
TIMERS = None
TIMED_NAMES = None       # optional set of op names to time (None: every wrapped op)
TIMERS_ACTIVE = True     # bench.py samples: an event record stalls the stream for ~3-6 us on either side of the launch it
                         # brackets (the next packet waits for the marker's signal), so only every n-th step is timed


class _Timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.on = TIMERS is not None and TIMERS_ACTIVE and (TIMED_NAMES is None or self.name in TIMED_NAMES)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            TIMERS.setdefault(self.name, []).append((self.e0, self.e1))


EVENT_POOL = []          # timing events that exist already (created by a record outside any timed region): _step_timing takes
                         # them from here first, because handing an event to the C call needs its handle, torch creates the
                         # handle on the first record, and a record costs the stream 3-6 us (bench.py parks its pre-warm events here)


def recycle_timers():
    """the events of the collected TIMERS go back to EVENT_POOL (call after timer_totals())"""
    for pairs in (TIMERS or {}).values():
        for pair in pairs:
            EVENT_POOL.extend(pair)
    if TIMERS is not None:
        TIMERS.clear()


def _timing_event():
    if EVENT_POOL:
        return EVENT_POOL.pop()
    e = torch.cuda.Event(enable_timing=True)
    e.record()                              # (torch creates the underlying event on its first record; the call re-records it)
    return e


def timer_totals():
    """{name: (calls, total_seconds)}; synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) * 1e-3) for k, v in (TIMERS or {}).items()}


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.is_contiguous())
    return t


def ray_setup(origins, dirs, pose, ext):
    """-> origins_s[B,3], dirs_s[B,3], hit[B,K] int32, zo[B]"""
    B, K = origins.shape[0], pose.shape[0]
    dev = origins.device
    o_s = torch.empty(B, 3, device=dev)
    d_s = torch.empty(B, 3, device=dev)
    hit = torch.empty(B, K, dtype=torch.int32, device=dev)
    zo = torch.empty(B, device=dev)
    _lib.check(_lib.lib().durf_ray_setup(_stream(), B, K, _p(_f32(origins)), _p(_f32(dirs)),
                                         _p(_f32(pose)), _p(_f32(ext)), _p(o_s), _p(d_s), _p(hit),
                                         _p(zo)), 'durf_ray_setup')
    return o_s, d_s, hit, zo


def ray_prologue(origins, dirs, pose, ext, viewdirs, near, far, N, t_rand=None, lindisp=False, pose_copy=None, zero=None,
                 seed=None, pack=None):
    """ray_setup + view_enc (bf16) + sample_t as ONE launch (durf_ray_prologue)
    -> origins_s[B,3], dirs_s[B,3], hit[B,K] int32, zo[B], view[B,32] bf16, t_vals[B,N+1]
    pose_copy [K,6]: receives a snapshot of `pose`; zero: a contiguous fp32 tensor the launch zero fills (the gradient)
    seed (int, instead of t_rand): the launch draws the step's stratified-sampling noise itself (Philox under this key) and
    also returns u_rand [3,B,N+1], the resampling draws behind levels 0, 1, 2 (Philox words 1, 2, 3) -> (..., t_vals, u_rand)
    pack = (bkgd_params, K, obj_params, obj_param_stride, want_bwd): pack_weights_all's work rides in the same launch
    (durf_ray_prologue_pack); its result ((bkgd_fwd, bkgd_bwd), (obj_fwd, obj_bwd) or None) is appended to the tuple"""
    B, K = origins.shape[0], pose.shape[0]
    dev = origins.device
    if zero is not None:
        assert zero.is_contiguous() and zero.dtype == torch.float32
    o_s = torch.empty(B, 3, device=dev)
    d_s = torch.empty(B, 3, device=dev)
    hit = torch.empty(B, K, dtype=torch.int32, device=dev)
    zo = torch.empty(B, device=dev)
    view = torch.empty(B, VIEW_DIM, dtype=torch.bfloat16, device=dev)
    t = torch.empty(B, N + 1, device=dev)
    u_out = None
    if seed is not None:
        assert t_rand is None
        u_out = torch.empty(3, B, N + 1, device=dev)
    lo, hi = (0, 0) if seed is None else split_seed(seed)
    common = (_stream(), B, K, N, _p(_f32(origins)), _p(_f32(dirs)), _p(_f32(pose)),
              _p(_f32(ext)), _p(o_s), _p(d_s), _p(hit), _p(zo), _p(_f32(viewdirs)), _p(view),
              _p(_f32(near)), _p(_f32(far)), _p(None if t_rand is None else _f32(t_rand)),
              int(lindisp), _p(t), _p(pose_copy if K else None), _p(zero),
              0 if zero is None else zero.numel(), lo, hi, _p(u_out))
    out = (o_s, d_s, hit, zo, view, t) if seed is None else (o_s, d_s, hit, zo, view, t, u_out)
    if pack is None:
        _lib.check(_lib.lib().durf_ray_prologue(*common), 'durf_ray_prologue')
        return out
    bkgd_params, Kp, obj_params, obj_param_stride, want_bwd = pack
    L = _lib.lib()
    u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
    bf = u8(L.durf_wpack_fwd_bytes(W_BKGD_))
    bb = u8(L.durf_wpack_bwd_bytes(W_BKGD_)) if want_bwd else None
    of = ob = None
    if Kp:
        of = u8(Kp * int(L.durf_wpack_fwd_bytes(W_OBJ_)))
        ob = u8(Kp * int(L.durf_wpack_bwd_bytes(W_OBJ_))) if want_bwd else None
    _lib.check(L.durf_ray_prologue_pack(*common, _p(_f32(bkgd_params)), IN_BKGD, _p(bf), _p(bb), int(Kp),
                                        _p(obj_params) if Kp else None, int(obj_param_stride), IN_OBJ_, _p(of), _p(ob), None, 0),
               'durf_ray_prologue_pack')
    return out + (((bf, bb), ((of, ob) if Kp else None)),)


def split_seed(seed):
    """host PRNG key (any Python int) -> the two 32-bit key words of the in-kernel Philox stream"""
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    return s & 0xFFFFFFFF, s >> 32


def compact_all(hit, N):
    """compact_hits + compact_classes as ONE launch -> (idx, count, slot), (idx2, count4, slot2, dyn)"""
    B, K = hit.shape
    dev = hit.device
    i32 = lambda *sh: torch.empty(*sh, dtype=torch.int32, device=dev)
    idx, count, slot = i32(K, B), i32(K), i32(B, K)
    idx2, count4, slot2, dyn = i32(2, B), i32(5), i32(B, 2), i32(B)
    _lib.check(_lib.lib().durf_compact_all(_stream(), B, K, N, _p(hit), _p(idx), _p(count), _p(slot), _p(idx2),
                                           _p(count4), _p(slot2), _p(dyn)), 'durf_compact_all')
    return (idx, count, slot), (idx2, count4, slot2, dyn)


_CONST = {}


def const_tensor(device, shape, dtype=torch.float32, value=0):
    """A cached, READ-ONLY constant tensor (the K = 0 model's empty hit lists, zero dyn_mask / box_rot / multi-hit count):
    a torch.zeros / torch.full per step is a fill launch each -- five of them were 5 % of a cfg1 step."""
    key = (str(device), tuple(shape), dtype, value)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full(tuple(shape), value, dtype=dtype, device=device)
    return t


def compact_hits(hit):
    """-> idx[K,B] int32, count[K] int32, slot[B,K] int32"""
    B, K = hit.shape
    dev = hit.device
    if K == 0:
        return (const_tensor(dev, (1, B), torch.int32), const_tensor(dev, (1,), torch.int32),
                const_tensor(dev, (B, 1), torch.int32, -1))
    # the kernel writes every slot and count entry, and idx[k, :count[k]] is all anyone reads
    idx = torch.empty(K, B, dtype=torch.int32, device=dev)
    count = torch.empty(K, dtype=torch.int32, device=dev)
    slot = torch.empty(B, K, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().durf_compact_hits(_stream(), B, K, _p(hit), _p(idx), _p(count), _p(slot)),
               'durf_compact_hits')
    return idx, count, slot


def compact_classes(hit, N):
    """-> idx[2,B], count[5] (class-0 rays, class-1 rays, valid compacted rows, multi-hit rays, bit mask of the boxes
    they hit), slot[B,2], dyn[B]:
    the ray classes of the de-duplicated background evaluation (class 1 = rays that hit exactly one box)"""
    B, K = hit.shape
    dev = hit.device
    idx = torch.empty(2, B, dtype=torch.int32, device=dev)
    count = torch.empty(5, dtype=torch.int32, device=dev)
    slot = torch.empty(B, 2, dtype=torch.int32, device=dev)
    dyn = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().durf_compact_classes(_stream(), B, K, N, _p(hit), _p(idx), _p(count), _p(slot), _p(dyn)),
               'durf_compact_classes')
    return idx, count, slot, dyn


def sample_t(near, far, N, t_rand=None, lindisp=False):
    B = near.shape[0]
    t = torch.empty(B, N + 1, device=near.device)
    _lib.check(_lib.lib().durf_sample_t(_stream(), B, N, _p(_f32(near)), _p(_f32(far)),
                                        _p(None if t_rand is None else _f32(t_rand)), int(lindisp), _p(t)),
               'durf_sample_t')
    return t


def view_enc(viewdirs, want_f32=False):
    B = viewdirs.shape[0]
    out = torch.empty(B, VIEW_DIM, dtype=torch.bfloat16, device=viewdirs.device)
    o32 = torch.empty(B, 27, device=viewdirs.device) if want_f32 else None
    _lib.check(_lib.lib().durf_view_enc(_stream(), B, _p(_f32(viewdirs)), _p(out), _p(o32)),
               'durf_view_enc')
    return (out, o32) if want_f32 else out


def tile_rows(rows):
    return (rows + 31) // 32 * 32


# include/durf_hip.h DURF_DISPATCH_*: the kernel variants the launchers choose by size (tests/test_gpu_dispatch_matrix.py)
DISPATCH = dict(FWD256_8W=0x1, FWD256_4W=0x2, FWD128_SAMPLE=0x4, FWD128_MSPLIT=0x8, BWD256_8W=0x10, BWD256_4W=0x20,
                BWD128_SAMPLE=0x40, BWD128_MSPLIT=0x80, DW256_256WG=0x100, DW256_512WG=0x200, DW128_128WG=0x400,
                DW128_256WG=0x800, FWD_ENC=0x1000, FWD_RAW_FULL=0x2000, FWD_TAIL=0x4000, F32_DW_TILE=0x8000,
                F32_DW_B2=0x10000, BWD_POSE=0x20000, FWD_MIX=0x40000, BWD_MIX=0x80000)


def dispatch_reset():
    _lib.lib().durf_dispatch_reset()


def dispatch_seen():
    """names of the size-selected kernel variants launched since dispatch_reset() (durf_dispatch_seen)"""
    m = int(_lib.lib().durf_dispatch_seen())
    return {k for k, b in DISPATCH.items() if m & b}


ENC_CONTRACT, ENC_NO_INTEGRATION, ENC_CYLINDER = 1, 2, 4
FWD_RAW_FULL = 8          # mlp_fwd_enc only (include/durf_hip.h DURF_FWD_RAW_FULL)


def encode_bkgd(t_vals, origins_s, dirs_s, radii, hit, contraction=True, tile=True, f32=False,
                disable_integration=False, cylinder=False, idx=None, count=None):
    """hit=None: no object masking of the samples (MipNerfModel.dynamics=False);
    idx / count: encode only rays idx[:count] into compacted rows (bf16 tiles)"""
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    K = 0 if hit is None else hit.shape[1]
    dev = t_vals.device
    ot = torch.empty(tile_rows(B * N), ENC_DIM, dtype=torch.bfloat16, device=dev) if tile else None
    of = torch.empty(B * N, 60, device=dev) if f32 else None
    with _Timed('encode_bkgd'):
        _lib.check(_lib.lib().durf_encode_bkgd(_stream(), B, N, _p(_f32(t_vals)), _p(_f32(origins_s)),
                                               _p(_f32(dirs_s)), _p(_f32(radii)), _p(hit), K,
                                               (ENC_CONTRACT if contraction else 0) | (ENC_NO_INTEGRATION if disable_integration else 0) |
                                               (ENC_CYLINDER if cylinder else 0),
                                               _p(ot), _p(of), _p(idx), _p(count)), 'durf_encode_bkgd')
    return ot, of


def barf_weights(alpha, max_deg=10):
    """mip.py:217-218 evaluated on the host in float32 arithmetic."""
    import numpy as np
    k = np.arange(max_deg, dtype=np.float32)
    a = np.clip(np.float32(alpha) - k, 0, 1).astype(np.float32) * np.float32(math.pi)
    return ((np.float32(1) - np.cos(a, dtype=np.float32)) / np.float32(2)).astype(np.float32)


def encode_obj(max_rays, idx_k, count_k, t_vals, origins_s, dirs_s, radii, alpha, tile=True, f32=False,
               disable_integration=False, cylinder=False):
    N = t_vals.shape[1] - 1
    dev = t_vals.device
    ot = torch.empty(tile_rows(max_rays * N), ENC_DIM, dtype=torch.bfloat16, device=dev) if tile else None
    of = torch.zeros(max_rays * N, 63, device=dev) if f32 else None
    w = barf_weights(alpha)
    wa = (C.c_float * 10)(*[float(x) for x in w])
    _lib.check(_lib.lib().durf_encode_obj(_stream(), max_rays, N, _p(idx_k), _p(count_k),
                                          _p(_f32(t_vals)), _p(_f32(origins_s)), _p(_f32(dirs_s)),
                                          _p(_f32(radii)), wa,
                                          (ENC_NO_INTEGRATION if disable_integration else 0) | (ENC_CYLINDER if cylinder else 0),
                                          _p(ot), _p(of)), 'durf_encode_obj')
    return ot, of


def mlp_param_count(width, in_dim):
    return int(_lib.lib().durf_mlp_param_count(width, in_dim))


def mlp_layer_offset(width, in_dim, layer, bias):
    return int(_lib.lib().durf_mlp_layer_offset(width, in_dim, layer, int(bias)))


def pack_weights(width, in_dim, mlp_params, want_bwd=False):
    """fp32 flax-layout params of one MLP -> bf16 fragment streams (fwd[, bwd])."""
    dev = mlp_params.device
    wf = torch.empty(int(_lib.lib().durf_wpack_fwd_bytes(width)), dtype=torch.uint8, device=dev)
    wb = None
    if want_bwd:
        wb = torch.empty(int(_lib.lib().durf_wpack_bwd_bytes(width)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().durf_pack_weights(_stream(), width, in_dim, _p(_f32(mlp_params)), _p(wf),
                                            _p(wb)), 'durf_pack_weights')
    return (wf, wb) if want_bwd else wf


def pack_weights_all(bkgd_params, K, obj_params, obj_param_stride, want_bwd=False):
    """Every weight stream of the model in one launch (durf_pack_weights_all): the background MLP and the K object MLPs
    (obj_params: BoxMLP_0 .. BoxMLP_{K-1} back to back, obj_param_stride floats apart).  Returns
    ((bkgd_fwd, bkgd_bwd), (obj_fwd, obj_bwd)); the bwd streams are None without want_bwd, the obj pair is None at K = 0."""
    L = _lib.lib()
    dev = bkgd_params.device
    u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
    bf = u8(L.durf_wpack_fwd_bytes(W_BKGD_))
    bb = u8(L.durf_wpack_bwd_bytes(W_BKGD_)) if want_bwd else None
    of = ob = None
    if K:
        of = u8(K * int(L.durf_wpack_fwd_bytes(W_OBJ_)))
        ob = u8(K * int(L.durf_wpack_bwd_bytes(W_OBJ_))) if want_bwd else None
    _lib.check(L.durf_pack_weights_all(_stream(), _p(_f32(bkgd_params)), IN_BKGD, _p(bf), _p(bb), int(K),
                                       _p(obj_params) if K else None, int(obj_param_stride), IN_OBJ_, _p(of), _p(ob)),
               'durf_pack_weights_all')
    return (bf, bb), ((of, ob) if K else None)


def mlp_stash_bytes(width, rows):
    return int(_lib.lib().durf_mlp_stash_bytes(width, rows))


def mlp_mask_bytes(rows):
    return int(_lib.lib().durf_mlp_mask_bytes(rows))


def mlp_fwd(width, rows, N, enc_tile, view_bf16, wpack_fwd, ray_idx=None, count=None, stash=None,
            raw=None, relu_mask=None, tail_idx=None, tail_count=None):
    """tail_idx / tail_count: once-per-ray rows of a de-duplicated batch after the count*N compacted rows"""
    dev = enc_tile.device
    if raw is None:
        raw = torch.empty(rows, 4, device=dev)
    with _Timed('mlp_fwd_%d%s' % (width, '_train' if stash is not None else '')):
        _lib.check(_lib.lib().durf_mlp_fwd(_stream(), width, rows, N, _p(enc_tile), _p(view_bf16),
                                           _p(ray_idx), _p(count), _p(wpack_fwd), _p(raw), _p(stash),
                                           _p(relu_mask), _p(tail_idx), _p(tail_count)),
                   'durf_mlp_fwd')
    return raw


# DURF_FUSED_ENCODE=0: the background encoding as its own launch in front of the forward (A/B switch; same results).
FUSED_ENCODE = os.environ.get('DURF_FUSED_ENCODE', '1') != '0'
# False: a de-duplicated forward writes compacted raw rows and expand_raw makes the full layout (the parity test of the
# scattered store toggles it: tests/test_gpu_fused_encode.py; not an environment switch any more)
FWD_SCATTER_RAW = True


def mlp_fwd_enc(rows, N, t_vals, origins_s, dirs_s, radii, hit, view_bf16, wpack_fwd, contraction=True,
                disable_integration=False, cylinder=False, ray_idx=None, count=None, stash=None, raw=None, relu_mask=None,
                tail_idx=None, tail_count=None, view_tile=None, raw_full=False):
    """durf_mlp_fwd_enc: the background forward that encodes its own tiles (encode_bkgd + mlp_fwd(256) as one launch)
    -> (raw, enc_tile); enc_tile is what encode_bkgd would have returned (the weight-gradient GEMMs read it).
    view_tile (training): a [tile_rows, 32] bf16 buffer the launch fills with expand_view's output; raw_full (with
    ray_idx / tail_idx): raw comes back in the full [B*N,4] layout -- expand_raw's output, without that launch"""
    dev = t_vals.device
    K = 0 if hit is None else hit.shape[1]
    if raw is None:
        raw = torch.empty(rows, 4, device=dev)
    enc_tile = torch.empty(tile_rows(rows), ENC_DIM, dtype=torch.bfloat16, device=dev)
    flags = ((ENC_CONTRACT if contraction else 0) | (ENC_NO_INTEGRATION if disable_integration else 0) |
             (ENC_CYLINDER if cylinder else 0) | (FWD_RAW_FULL if raw_full else 0))
    with _Timed('mlp_fwd_256%s' % ('_train' if stash is not None else '')):
        _lib.check(_lib.lib().durf_mlp_fwd_enc(_stream(), rows, N, _p(_f32(t_vals)), _p(_f32(origins_s)), _p(_f32(dirs_s)),
                                               _p(_f32(radii)), _p(hit), K, flags, _p(enc_tile), _p(view_bf16),
                                               _p(ray_idx), _p(count), _p(wpack_fwd), _p(raw), _p(stash), _p(relu_mask),
                                               _p(tail_idx), _p(tail_count), _p(view_tile)), 'durf_mlp_fwd_enc')
    return raw, enc_tile


# The K object MLPs touch ~10 % of the rays: their launches are small and latency-bound.  DURF_OVERLAP_OBJECTS issues the
# object work of a stage on a side HIP stream, forked after the background encode / the loss kernel and joined before its
# results are consumed:
#   2 forward + backward + weight gradients;   0 everything on one stream;   1 the object forward;   3 forward + backward.
# The persistent background forward / backward (one workgroup per CU) leave the object launches no CU until their own
# tail, so 1 and 3 measure like 0; what pays is that the objects' weight-gradient launch, queued behind their backward on
# the side stream, starts in the tail of the background backward instead of after it.  Round 3, three interleaved runs per
# mode on one box, k rays/s: cfg3 0: 949 / 948 / 950, 2: 960 / 961 / 954;  cfg2 0: 958 / 960 / 947, 2: 972 / 962 / 961;
# cfg5 0: 704 / 712 / 709, 2: 721 / 702 / 721;  cfg4 (objects on the fp32 kernels, main stream): no change.  The
# background weight-gradient launch then shares its first ~100 us with the objects' (1444-1452 -> 1506-1515 us, HIP events),
# which is what bench.py's roofline line reports.  Moving ONLY the objects' weight gradients to the side stream, started
# together with the background ones, is destructive (background launch 1725-1777 us, 898-910 k rays/s): measured, dropped.
# The same switch costs at small batches, where every kernel is one latency-bound round and a fork / join is one more
# dependency in the chain (cfg3 shape, k rays/s, 0 vs 2: 512 rays 595 -> 545-559, 1024 rays 750-756 -> 745, 2048 rays 883
# -> 902-908, 4096 rays above): 'auto' (default) = 2 from 2048 x 128 sample rows per step (4 rounds of background blocks), else 0.
_MODE = os.environ.get('DURF_OVERLAP_OBJECTS', 'auto')
OVERLAP_MIN_ROWS = 2048 * 128


def set_overlap_mode(mode):
    """'auto' / '0' .. '3' for the Python-issued launches; the one-call C step reads DURF_OVERLAP_OBJECTS per call and knows
    'auto', '0' and '2' only -- it runs '1' and '3' (experiment modes of this file) as '2' (csrc/side_stream.h)"""
    global _MODE
    _MODE = mode
    os.environ['DURF_OVERLAP_OBJECTS'] = mode


def overlap_mode(rows):
    """'0' .. '3' for a step (or a render chunk) of `rows` sample rows per level"""
    if _MODE == 'auto':
        return '2' if rows >= OVERLAP_MIN_ROWS else '0'
    return _MODE


def overlap_forward(rows):
    return overlap_mode(rows) in ('1', '2', '3')


def overlap_backward(rows):
    return overlap_mode(rows) in ('2', '3')


def overlap_dw(rows):
    return overlap_mode(rows) == '2'


_SIDE = {}


def side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class on_side:
    """with on_side(device, enabled): ... -- launches inside go to the side stream (no-op when disabled).  fork():
    the side stream waits for everything issued so far on the current stream; join(): the reverse."""

    def __init__(self, device, enabled=True):
        self.enabled = bool(enabled) and device.type == 'cuda'
        self.side = side_stream(device) if self.enabled else None
        self.ctx = None

    def fork(self):
        if self.enabled:
            self.side.wait_stream(torch.cuda.current_stream())

    def join(self):
        if self.enabled:
            torch.cuda.current_stream().wait_stream(self.side)

    def __enter__(self):
        if self.enabled:
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.enabled:
            self.ctx.__exit__(*a)


# Evaluate the background MLP once per box-hit ray instead of once per sample (exact: see durf_expand_raw in
# include/durf_hip.h).  Module switch for A/B measurements and tests.
DEDUP_HIT_RAYS = os.environ.get('DURF_DEDUP_HIT_RAYS', '1') != '0'


def expand_raw(B, N, raw_c, slot2, count2, raw_tail=None):
    """compacted raw (count2[0]*N rows of the sample-by-sample rays, then one row per box-hit ray) -> [B*N,4];
    raw_tail [B,4]: fp32 re-evaluation of the box-hit rays' rows (mlp_fwd_f32 with enc=None), used instead of them"""
    raw_full = torch.empty(B * N, 4, device=raw_c.device)
    # the tail starts at row count2[0]*N: hand the kernel a pointer to that row (device-side count -> device-side offset
    # is not available on the host, so the kernel takes the base and the slot of the tail separately)
    _lib.check(_lib.lib().durf_expand_raw(_stream(), B, N, _p(_f32(raw_c)), _p(count2), _p(slot2), _p(raw_full),
                                          _p(raw_tail)), 'durf_expand_raw')
    return raw_full


BKGD_GREY, BKGD_WHITE, BKGD_RAND = 0, 1, 2


def composite_fwd(raw_bkgd, raw_obj, slot, t_vals, dirs_s, density_bias=-1.0, bkgd_mode=BKGD_GREY,
                  want_t=True):
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    K = len(raw_obj)
    dev = t_vals.device
    rgb = torch.empty(B, 3, device=dev)
    depth = torch.empty(B, device=dev)
    acc = torch.empty(B, device=dev)
    weights = torch.empty(B, N, device=dev)
    t_mids = torch.empty(B, N, device=dev) if want_t else None
    t_dists = torch.empty(B, N, device=dev) if want_t else None
    ptrs = (C.c_void_p * max(K, 1))(*[r.data_ptr() for r in raw_obj])
    with _Timed('composite_fwd'):
        _lib.check(_lib.lib().durf_composite_fwd(_stream(), B, N, K, _p(_f32(raw_bkgd)), ptrs, _p(slot),
                                                 _p(_f32(t_vals)), _p(_f32(dirs_s)), density_bias,
                                                 bkgd_mode, _p(rgb), _p(depth), _p(acc), _p(weights),
                                                 _p(t_mids), _p(t_dists)), 'durf_composite_fwd')
    return rgb, depth, acc, weights, t_mids, t_dists


def composite_resample(raw_bkgd, raw_obj, slot, t_vals, dirs_s, density_bias=-1.0, bkgd_mode=BKGD_GREY,
                       padding=0.01, u_rand=None, want_t=True, prep=None):
    """composite_fwd + resample in one launch (bit-identical to the two calls) -> (rgb, depth, acc, weights,
    t_mids, t_dists, t_vals_next).  prep = dict(lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, level,
    disable_multiscale, norms [L,5]) also fills norms[level] (when level == 0) and norms[level + 1]: durf_loss_prep's
    job for both levels."""
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    K = len(raw_obj)
    dev = t_vals.device
    rgb = torch.empty(B, 3, device=dev)
    depth = torch.empty(B, device=dev)
    acc = torch.empty(B, device=dev)
    weights = torch.empty(B, N, device=dev)
    t_mids = torch.empty(B, N, device=dev) if want_t else None
    t_dists = torch.empty(B, N, device=dev) if want_t else None
    t_next = torch.empty(B, N + 1, device=dev)
    ptrs = (C.c_void_p * max(K, 1))(*[r.data_ptr() for r in raw_obj])
    pa = [None] * 5 + [0.0, 0.0, 0, 0] + [None] * 4
    if prep is not None:
        lvl = int(prep['level'])
        norms = prep['norms']
        buf = torch.empty(2, PREP_ROWS, B, device=dev)
        pa = [_p(_f32(prep['lossmult'])), _p(_f32(prep['gt_depth'])), _p(_f32(prep['sky'])), _p(prep['dyn']),
              _p(_f32(prep['zo'])), float(prep['eps']), float(prep['box_loss_mult']), lvl,
              int(prep['disable_multiscale']),
              _p(buf[0]) if lvl == 0 else None, _p(norms[lvl]) if lvl == 0 else None, _p(buf[1]), _p(norms[lvl + 1])]
    with _Timed('composite_resample'):
        _lib.check(_lib.lib().durf_composite_resample(
            _stream(), B, N, K, _p(_f32(raw_bkgd)), ptrs, _p(slot), _p(_f32(t_vals)), _p(_f32(dirs_s)), density_bias,
            bkgd_mode, _p(rgb), _p(depth), _p(acc), _p(weights), _p(t_mids), _p(t_dists), padding,
            _p(None if u_rand is None else _f32(u_rand)), _p(t_next), *pa), 'durf_composite_resample')
    return rgb, depth, acc, weights, t_mids, t_dists, t_next


def resample(t_vals, weights, padding=0.01, u_rand=None):
    B, N = weights.shape
    out = torch.empty(B, N + 1, device=t_vals.device)
    _lib.check(_lib.lib().durf_resample(_stream(), B, N, _p(_f32(t_vals)), _p(_f32(weights)), padding,
                                        _p(None if u_rand is None else _f32(u_rand)), _p(out)),
               'durf_resample')
    return out


def sorted_piecewise_constant_pdf(bins, weights, u_rand=None):
    """math.sorted_piecewise_constant_pdf(key, bins, weights, num_samples=N+1, randomized=u_rand is not None)"""
    B, N = weights.shape
    out = torch.empty(B, N + 1, device=bins.device)
    _lib.check(_lib.lib().durf_sorted_piecewise_constant_pdf(_stream(), B, N, _p(_f32(bins)), _p(_f32(weights)),
                                                             _p(None if u_rand is None else _f32(u_rand)), _p(out)),
               'durf_sorted_piecewise_constant_pdf')
    return out


# ---------------------------------------------------------------------------
# training ops
# ---------------------------------------------------------------------------
PREP_ROWS, TERM_ROWS = 5, 7
TERM_NAMES = ('rgb', 'obj', 'depth', 'near', 'empty', 'sky', 'dist')


def loss_prep(t_vals, lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, level, disable_multiscale=False,
              norm=None):
    """-> norm[5] device floats: sum m, sum depth_mask, sum sky_mask, min near-dist^2, sum dyn."""
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    dev = t_vals.device
    prep = torch.empty(PREP_ROWS, B, device=dev)
    if norm is None:
        norm = torch.empty(PREP_ROWS, device=dev)
    _lib.check(_lib.lib().durf_loss_prep(_stream(), B, N, _p(_f32(t_vals)), _p(_f32(lossmult)),
                                         _p(_f32(gt_depth)), _p(_f32(sky)), _p(dyn), _p(_f32(zo)),
                                         eps, box_loss_mult, level, int(disable_multiscale), _p(prep),
                                         _p(norm)), 'durf_loss_prep')
    return norm


def loss_bwd(raw_bkgd, raw_obj, slot, t_vals, dirs_s, pixels, lossmult, gt_depth, sky, dyn, zo, norm, eps,
             mults, box_loss_mult, level, bg, density_bias=-1.0, disable_multiscale=False, sums=None, render_out=None,
             draw_ray_sum=None, defer_sums=False):
    """-> draw [B*N,4], term_sums[7] (rgb, obj, depth, near, empty, sky, dist numerators).
    defer_sums: no reduction launch here; returns (draw, terms [7,B]) for train_stats(..., terms=...) to reduce.
    render_out = (rgb [B,3], depth [B], acc [B], weights [B,N], t_mids [B,N], t_dists [B,N]) tensors to fill with the level's rendered
    outputs (what composite_fwd returns; a training step then skips that launch for the last level)."""
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    K = len(raw_obj)
    dev = t_vals.device
    draw = torch.empty(B * N, 4, device=dev)
    terms = torch.empty(TERM_ROWS, B, device=dev)
    if sums is None and not defer_sums:
        sums = torch.empty(TERM_ROWS, device=dev)
    ptrs = (C.c_void_p * max(K, 1))(*[r.data_ptr() for r in raw_obj])
    m = (C.c_float * 6)(*[float(x) for x in mults])
    _lib.check(_lib.lib().durf_loss_bwd(_stream(), B, N, K, _p(_f32(raw_bkgd)), ptrs, _p(slot),
                                        _p(_f32(t_vals)), _p(_f32(dirs_s)), _p(_f32(pixels)),
                                        _p(_f32(lossmult)), _p(_f32(gt_depth)), _p(_f32(sky)), _p(dyn),
                                        _p(_f32(zo)), _p(norm), eps, m, box_loss_mult, level,
                                        int(disable_multiscale), bg, density_bias, _p(draw), _p(terms),
                                        None if defer_sums else _p(sums), *[_p(t) for t in (render_out or (None,) * 6)],
                                        _p(draw_ray_sum)), 'durf_loss_bwd')
    return draw, (terms if defer_sums else sums)


class LossLevel(C.Structure):
    """durf_loss_level (include/durf_hip.h), field for field"""
    _fields_ = ([('raw_bkgd', C.c_void_p), ('raw_obj', C.c_void_p * 16), ('t_vals', C.c_void_p), ('norm', C.c_void_p),
                 ('mults', C.c_float * 6), ('level', C.c_int)] +
                [(n, C.c_void_p) for n in ('draw', 'terms', 'rgb_out', 'depth_out', 'acc_out', 'weights_out', 't_mids_out',
                                           't_dists_out', 'draw_ray_sum')])


def loss_bwd_levels(levels, slot, dirs_s, pixels, lossmult, gt_depth, sky, dyn, zo, eps, box_loss_mult, bg, density_bias=-1.0,
                    disable_multiscale=False):
    """loss_bwd(defer_sums=True) of EVERY level as one launch (durf_loss_bwd_levels).  levels: one dict per level with
    raw_bkgd, raw_obj (list), t_vals, norm, mults, level[, render_out (6 tensors)][, draw_ray_sum]
    -> [(draw [B*N,4], terms [7,B])] per level"""
    L = len(levels)
    B, N = levels[0]['t_vals'].shape[0], levels[0]['t_vals'].shape[1] - 1
    K = len(levels[0]['raw_obj'])
    dev = levels[0]['t_vals'].device
    arr = (LossLevel * L)()
    out, keep = [], []
    for i, lv in enumerate(levels):
        a = arr[i]
        draw = torch.empty(B * N, 4, device=dev)
        terms = torch.empty(TERM_ROWS, B, device=dev)
        a.raw_bkgd, a.t_vals, a.norm = _p(_f32(lv['raw_bkgd'])), _p(_f32(lv['t_vals'])), _p(lv['norm'])
        for k, r in enumerate(lv['raw_obj']):
            a.raw_obj[k] = r.data_ptr()
        a.mults = (C.c_float * 6)(*[float(x) for x in lv['mults']])
        a.level = int(lv['level'])
        a.draw, a.terms = _p(draw), _p(terms)
        ro = lv.get('render_out') or (None,) * 6
        a.rgb_out, a.depth_out, a.acc_out, a.weights_out, a.t_mids_out, a.t_dists_out = (_p(t) for t in ro)
        a.draw_ray_sum = _p(lv.get('draw_ray_sum'))
        keep.append(lv)
        out.append((draw, terms))
    _lib.check(_lib.lib().durf_loss_bwd_levels(_stream(), B, N, K, L, C.cast(arr, C.c_void_p), _p(slot), _p(_f32(dirs_s)),
                                               _p(_f32(pixels)), _p(_f32(lossmult)), _p(_f32(gt_depth)), _p(_f32(sky)), _p(dyn),
                                               _p(_f32(zo)), float(eps), float(box_loss_mult), int(disable_multiscale),
                                               float(bg), float(density_bias)), 'durf_loss_bwd_levels')
    return out


STAT_ROWS = ('losses', 'obj_losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses', 'tv_losses',
             'offsets', 'offset_x', 'offset_y', 'offset_z', 'offset_yaw', 'psnrs', 'obj_psnrs')
STATS_ASSEMBLE, STATS_PSNR = 1, 2


def train_stats(norms, sums, weight_l2, pose6, prev6, target6, t_vals_levels, mults, mode, out=None, terms=None):
    """Scalars of utils.Stats in one launch; see durf_train_stats.  -> out [2 + 17 L]
    terms: per-level [7,B] tensors of loss_bwd(defer_sums=True); reduced into `sums` by the same launch."""
    L = norms.shape[0]
    K = 0 if pose6 is None else pose6.shape[0]
    N = t_vals_levels[0].shape[1] - 1
    if out is None:
        out = torch.empty(2 + 17 * L, device=norms.device)
    ptrs = (C.c_void_p * L)(*[t.data_ptr() for t in t_vals_levels])
    m = (C.c_float * 6)(*[float(x) for x in mults])
    _lib.check(_lib.lib().durf_train_stats(_stream(), L, K, N, _p(norms), _p(sums), _p(weight_l2),
                                           _p(pose6) if K else None, _p(prev6) if K else None,
                                           _p(target6) if K else None, ptrs, m, mode, _p(out),
                                           (C.c_void_p * L)(*[t.data_ptr() for t in terms]) if terms is not None else None,
                                           int(terms[0].shape[1]) if terms is not None else 0), 'durf_train_stats')
    return out


def stats_scrub(norms, sums, weight_l2, pose6, prev6, target6, t_vals_levels, mults, mode, terms, grad, inv_world, max_val,
                poison=None):
    """train_stats + the scrub pass of clip_adam (+ the multi-hit outcome: poison = (cls_count, box_floats, K, mlp0_floats,
    obj_floats) of poison_multi_hit, single device only) as ONE launch (durf_stats_scrub) -> (out [2 + 17 L], scratch);
    adam_apply(.., scratch) finishes the update"""
    L = norms.shape[0]
    K = 0 if pose6 is None else pose6.shape[0]
    N = t_vals_levels[0].shape[1] - 1
    dev = norms.device
    out = torch.empty(2 + 17 * L, device=dev)
    n = grad.numel()
    scratch = torch.empty(int(_lib.lib().durf_optim_scratch_floats(n)), device=dev)
    ptrs = (C.c_void_p * L)(*[t.data_ptr() for t in t_vals_levels])
    m = (C.c_float * 6)(*[float(x) for x in mults])
    cls, bf, Kb, m0, of = poison if poison is not None else (None, 0, 0, 0, 0)
    _lib.check(_lib.lib().durf_stats_scrub(_stream(), L, K, N, _p(norms), _p(sums), _p(weight_l2),
                                           _p(pose6) if K else None, _p(prev6) if K else None, _p(target6) if K else None,
                                           ptrs, m, mode, _p(out),
                                           (C.c_void_p * L)(*[t.data_ptr() for t in terms]) if terms is not None else None,
                                           int(terms[0].shape[1]) if terms is not None else 0, n, _p(_f32(grad)),
                                           float(inv_world), float(max_val), _p(scratch), _p(cls), int(bf), int(Kb), int(m0),
                                           int(of)), 'durf_stats_scrub')
    return out, scratch


def adam_apply(params, m, v, grad, max_norm, lr, step, scratch):
    """the Adam pass of clip_adam behind stats_scrub's scrub pass -> stats[4] as clip_adam"""
    _bump_generation(params)
    stats = torch.empty(4, device=params.device)
    _lib.check(_lib.lib().durf_adam_apply(_stream(), params.numel(), _p(_f32(params)), _p(_f32(m)), _p(_f32(v)),
                                          _p(_f32(grad)), float(max_norm), float(lr), int(step), _p(scratch), _p(stats)),
               'durf_adam_apply')
    return stats


def weight_decay(params, grad, mult, lo=0, hi=None, want_l2=True):
    """Config.weight_decay_mult (train_boxpose.py:73-75): grad[lo:hi) += (2 mult / n) params[lo:hi); -> weight_l2 [1] =
    mult * mean(params^2) over the whole buffer (None unless want_l2)"""
    n = params.numel()
    hi = n if hi is None else hi
    L = _lib.lib()
    out = torch.empty(1, device=params.device) if want_l2 else None
    scratch = torch.empty(int(L.durf_optim_scratch_floats(n)), device=params.device) if want_l2 else None
    _lib.check(L.durf_weight_decay(_stream(), n, _p(_f32(params)), _p(_f32(grad)), int(lo), int(hi), float(mult), _p(scratch),
                                   _p(out)), 'durf_weight_decay')
    return out


def density_noise(raw, scale, normal=None, seed=None, level=0):
    """MipNerfModel.density_noise (obbpose_model.py:236-240): raw[:, 3] += scale * z in place; z = `normal` [rows] or the
    library's own draws under `seed` (Philox block (row, 1 + level, 0, 0) through Box-Muller; oracle/philox_ref.py)"""
    assert raw.dim() == 2 and raw.shape[1] == 4 and raw.is_contiguous()
    assert (normal is None) != (seed is None), 'either the draws or the key they are made under'
    if normal is not None:
        normal = _f32(normal.reshape(-1).contiguous())
        assert normal.numel() == raw.shape[0]
    lo, hi = (0, 0) if seed is None else split_seed(seed)
    _lib.check(_lib.lib().durf_density_noise(_stream(), raw.shape[0], _p(_f32(raw)), float(scale), _p(normal), lo, hi, int(level)),
               'durf_density_noise')
    return raw


class Comm:
    """A communicator of the library's own in-stream all-reduce (csrc/comm.hip: RCCL resolved at run time)"""

    def __init__(self, world, rank, uid):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), COMM_ID_BYTES)
        _lib.check(_lib.lib().durf_comm_init(int(world), int(rank), C.cast(buf, C.c_void_p), C.byref(h)), 'durf_comm_init')
        self.handle, self.world, self.rank = h, int(world), int(rank)

    def all_reduce_sum(self, t):
        """in place, on the current stream (fp32, contiguous)"""
        assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(_lib.lib().durf_allreduce_sum(_stream(), self.handle, _p(t), t.numel()), 'durf_allreduce_sum')

    def destroy(self):
        if self.handle:
            _lib.check(_lib.lib().durf_comm_destroy(self.handle), 'durf_comm_destroy')
            self.handle = None


COMM_ID_BYTES = 128


def comm_available():
    return bool(_lib.lib().durf_comm_available())


def comm_unique_id():
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _lib.check(_lib.lib().durf_comm_unique_id(C.cast(buf, C.c_void_p)), 'durf_comm_unique_id')
    return buf.raw


def stats_views(out, L):
    """dict of named views into the durf_train_stats buffer"""
    d = {'loss': out[0], 'sampling_stats': out[1 + 15 * L:1 + 17 * L], 'weight_l2': out[1 + 17 * L]}
    for i, name in enumerate(STAT_ROWS):
        d[name] = out[1 + i * L:1 + (i + 1) * L]
    return d


def mlp_bwd(width, rows, N, draw, wpack_bwd, relu_mask, ray_idx=None, count=None, want_d_enc=False,
            tail_idx=None, tail_count=None, draw_ray_sum=None):
    """-> dz (same layout as the stash), dz_out tile [rows,16][, d_enc [rows,64] fp32]"""
    dev = draw.device
    dz = torch.empty(mlp_stash_bytes(width, rows), dtype=torch.uint8, device=dev)
    dz_out = torch.empty(tile_rows(rows), 16, dtype=torch.bfloat16, device=dev)
    d_enc = torch.zeros(rows, ENC_DIM, device=dev) if want_d_enc else None
    with _Timed('mlp_bwd_%d' % width):
        _lib.check(_lib.lib().durf_mlp_bwd(_stream(), width, rows, N, _p(_f32(draw)), _p(ray_idx), _p(count),
                                           _p(wpack_bwd), _p(relu_mask), _p(dz), _p(dz_out), _p(d_enc),
                                           _p(tail_idx), _p(tail_count), _p(draw_ray_sum)),
                   'durf_mlp_bwd')
    return (dz, dz_out, d_enc) if want_d_enc else (dz, dz_out)


def expand_view(rows, N, view_bf16, ray_idx=None, count=None, tail_idx=None, tail_count=None):
    out = torch.empty(tile_rows(rows), VIEW_DIM, dtype=torch.bfloat16, device=view_bf16.device)
    _lib.check(_lib.lib().durf_expand_view(_stream(), rows, N, _p(view_bf16), _p(ray_idx), _p(count),
                                           _p(out), _p(tail_idx), _p(tail_count)), 'durf_expand_view')
    return out


def dw_buffers(width, device):
    part = torch.empty(int(_lib.lib().durf_dw_part_floats(width)), device=device)
    bpart = torch.empty(int(_lib.lib().durf_dw_bpart_floats(width)), device=device)
    return part, bpart


def mlp_dw(width, rows, N, enc_tiles, view_tiles, stashes, dzs, dz_outs, part, bpart, count=None):
    """Weight-gradient partials of one MLP over the samples of every level (lists: one entry per level)."""
    L = len(stashes)
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    with _Timed('mlp_dw_%d' % width):
        _lib.check(_lib.lib().durf_mlp_dw(_stream(), width, rows, N, _p(count), L, arr(enc_tiles), arr(view_tiles),
                                          arr(stashes), arr(dzs), arr(dz_outs), _p(part), _p(bpart)), 'durf_mlp_dw')


def _levels_args(rows_l, n_l, count_l):
    L = len(rows_l)
    return (L, (C.c_size_t * L)(*[int(r) for r in rows_l]), (C.c_int * L)(*[int(n) for n in n_l]),
            (C.c_void_p * L)(*[None if c is None else c.data_ptr() for c in count_l]))


def mlp_dw_levels(width, rows_l, n_l, count_l, enc_tiles, view_tiles, stashes, dzs, dz_outs, part, bpart):
    """mlp_dw with per-segment geometry: row capacity, rows per ray and device ray count of every segment"""
    L, rows_a, n_a, cnt_a = _levels_args(rows_l, n_l, count_l)
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    with _Timed('mlp_dw_%d' % width):
        _lib.check(_lib.lib().durf_mlp_dw_levels(_stream(), width, L, rows_a, n_a, cnt_a, arr(enc_tiles), arr(view_tiles),
                                                 arr(stashes), arr(dzs), arr(dz_outs), _p(part), _p(bpart)),
                   'durf_mlp_dw_levels')


def mlp_dw_finalize_levels(width, in_dim, rows_l, n_l, count_l, part, bpart, grad_mlp, mlp_params):
    L, rows_a, n_a, cnt_a = _levels_args(rows_l, n_l, count_l)
    with _Timed('mlp_dw_finalize_%d' % width):
        _lib.check(_lib.lib().durf_mlp_dw_finalize_levels(_stream(), width, in_dim, L, rows_a, n_a, cnt_a, _p(part),
                                                          _p(bpart), _p(grad_mlp), _p(_f32(mlp_params))),
                   'durf_mlp_dw_finalize_levels')


def mlp_dw_finalize(width, in_dim, rows, N, nlevels, part, bpart, grad_mlp, mlp_params, count=None):
    """rows, N, nlevels, count: as in the mlp_dw call that wrote the partials; mlp_params: the MLP's fp32 parameters
    (the linear bottleneck layer's gradients are derived from them, see durf_mlp_dw_finalize)"""
    with _Timed('mlp_dw_finalize_%d' % width):
        _lib.check(_lib.lib().durf_mlp_dw_finalize(_stream(), width, in_dim, rows, N, _p(count), nlevels, _p(part),
                                                   _p(bpart), _p(grad_mlp), _p(_f32(mlp_params))), 'durf_mlp_dw_finalize')


# ---------------------------------------------------------------------------
# the K object MLPs of one level as one call each (csrc/objects.hip)
# ---------------------------------------------------------------------------
W_OBJ_, IN_OBJ_ = 128, 63
W_BKGD_, IN_BKGD = 256, 60


class ObjSlabs:
    """[K, ...] slabs of one level for the batched object calls (strides fixed by the library)."""

    def __init__(self, K, B, N, device, train):
        L = _lib.lib()
        rows = B * N
        u8 = lambda n: torch.empty(K * int(n), dtype=torch.uint8, device=device)
        self.K, self.B, self.N = K, B, N
        self.enc = u8(L.durf_obj_enc_stride(B, N))
        self.raw = torch.empty(K, rows, 4, device=device)
        self.stash = u8(mlp_stash_bytes(W_OBJ_, rows)) if train else None
        self.mask = u8(mlp_mask_bytes(rows)) if train else None
        self.dz = self.dz_out = self.d_enc = None

    def raws(self):
        return [self.raw[k] for k in range(self.K)]


def pack_weights_batch(K, obj_params, param_stride, want_bwd=False):
    """obj_params: flat fp32 params of BoxMLP_0 .. BoxMLP_{K-1}, back to back"""
    L = _lib.lib()
    dev = obj_params.device
    wf = torch.empty(K * int(L.durf_wpack_fwd_bytes(W_OBJ_)), dtype=torch.uint8, device=dev)
    wb = torch.empty(K * int(L.durf_wpack_bwd_bytes(W_OBJ_)), dtype=torch.uint8, device=dev) if want_bwd else None
    _lib.check(L.durf_pack_weights_batch(_stream(), W_OBJ_, IN_OBJ_, K, _p(obj_params), param_stride, _p(wf), _p(wb)),
               'durf_pack_weights_batch')
    return wf, wb


def obj_fwd_batch(slabs, idx, count, t_vals, origins_s, dirs_s, radii, alpha, view_bf16, wf, view_tile=None,
                  disable_integration=False, cylinder=False):
    w = barf_weights(alpha)
    wa = (C.c_float * 10)(*[float(x) for x in w])
    with _Timed('obj_fwd_batch'):
        _lib.check(_lib.lib().durf_obj_fwd_batch(
            _stream(), slabs.K, slabs.B, slabs.N, _p(idx), _p(count), _p(_f32(t_vals)), _p(_f32(origins_s)),
            _p(_f32(dirs_s)), _p(_f32(radii)), wa,
            (ENC_NO_INTEGRATION if disable_integration else 0) | (ENC_CYLINDER if cylinder else 0), _p(view_bf16),
            _p(wf), _p(slabs.enc), _p(slabs.raw), _p(slabs.stash), _p(slabs.mask), _p(view_tile)), 'durf_obj_fwd_batch')


def obj_mix(rows):
    """whether a training step of `rows` sample rows per level issues its bf16 object MLPs as items of the background MLP's
    persistent launches (durf_mlp_fwd_enc_obj / durf_mlp_bwd_obj: one stream, the M-split regime; DURF_OBJ_MIX=0: A/B switch)"""
    return (os.environ.get('DURF_OBJ_MIX', '1') != '0' and os.environ.get('DURF_OBJ_MSPLIT', '1') != '0' and
            rows < OVERLAP_MIN_ROWS and overlap_mode(rows) == '0')


def mlp_fwd_enc_obj(rows, N, t_vals, origins_s, dirs_s, radii, hit, view_bf16, wpack_fwd, slabs, obj_idx, obj_count, alpha, obj_wf,
                    ray_idx, count, tail_idx, tail_count, stash, relu_mask, contraction=True, disable_integration=False,
                    cylinder=False, view_tile=None, raw_full=True, obj_view_tile=None):
    """mlp_fwd_enc (the background MLP on the de-duplicated ray classes) + obj_fwd_batch (the K object MLPs into `slabs`) as
    ONE launch (durf_mlp_fwd_enc_obj: the heterogeneous persistent grid of a small training step) -> (raw, enc_tile);
    every output bit-identical to the two calls"""
    dev = t_vals.device
    K = hit.shape[1]
    raw = torch.empty(rows, 4, device=dev)
    enc_tile = torch.empty(tile_rows(rows), ENC_DIM, dtype=torch.bfloat16, device=dev)
    flags = ((ENC_CONTRACT if contraction else 0) | (ENC_NO_INTEGRATION if disable_integration else 0) |
             (ENC_CYLINDER if cylinder else 0) | (FWD_RAW_FULL if raw_full else 0))
    wa = (C.c_float * 10)(*[float(x) for x in barf_weights(alpha)])
    with _Timed('mlp_fwd_256%s' % ('_train' if stash is not None else '')):
        _lib.check(_lib.lib().durf_mlp_fwd_enc_obj(
            _stream(), rows, N, _p(_f32(t_vals)), _p(_f32(origins_s)), _p(_f32(dirs_s)), _p(_f32(radii)), _p(hit), K, flags,
            _p(enc_tile), _p(view_bf16), _p(ray_idx), _p(count), _p(wpack_fwd), _p(raw), _p(stash), _p(relu_mask), _p(tail_idx),
            _p(tail_count), _p(view_tile), slabs.B, _p(obj_idx), _p(obj_count), wa,
            (ENC_NO_INTEGRATION if disable_integration else 0) | (ENC_CYLINDER if cylinder else 0), _p(obj_wf), _p(slabs.enc),
            _p(slabs.raw), _p(slabs.stash), _p(slabs.mask), _p(obj_view_tile)), 'durf_mlp_fwd_enc_obj')
    return raw, enc_tile


def mlp_bwd_obj(rows, N, draw, wpack_bwd, relu_mask, ray_idx, count, tail_idx, tail_count, draw_ray_sum, slabs_levels, obj_idx,
                obj_count, obj_draws, obj_wb):
    """mlp_bwd(256) of the background MLP + obj_bwd_batch_levels of the K object MLPs (slabs_levels / obj_draws: per level)
    as ONE launch (durf_mlp_bwd_obj) -> (dz, dz_out) of the background MLP; the slabs receive dz / dz_out; bit-identical"""
    L = _lib.lib()
    dev = draw.device
    dz = torch.empty(mlp_stash_bytes(256, rows), dtype=torch.uint8, device=dev)
    dz_out = torch.empty(tile_rows(rows), 16, dtype=torch.bfloat16, device=dev)
    s0 = slabs_levels[0]
    K, B, nl = s0.K, s0.B, len(slabs_levels)
    for sl in slabs_levels:
        sl.dz = torch.empty(K * mlp_stash_bytes(W_OBJ_, B * N), dtype=torch.uint8, device=dev)
        sl.dz_out = torch.empty(K * int(L.durf_obj_dzout_stride(B, N)), dtype=torch.uint8, device=dev)
        sl.d_enc = None
    arr = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    with _Timed('mlp_bwd_256'):
        _lib.check(L.durf_mlp_bwd_obj(_stream(), rows, N, _p(_f32(draw)), _p(ray_idx), _p(count), _p(wpack_bwd), _p(relu_mask),
                                      _p(dz), _p(dz_out), _p(tail_idx), _p(tail_count), _p(draw_ray_sum), K, B, nl, _p(obj_idx),
                                      _p(obj_count), arr([_f32(d) for d in obj_draws]), _p(obj_wb),
                                      arr([s.mask for s in slabs_levels]), arr([s.dz for s in slabs_levels]),
                                      arr([s.dz_out for s in slabs_levels])), 'durf_mlp_bwd_obj')
    return dz, dz_out


def obj_view_tiles(K, B, N, device):
    return torch.empty(K * int(_lib.lib().durf_obj_view_stride(B, N)), dtype=torch.uint8, device=device)


def obj_bwd_batch(slabs, idx, count, draw, wb, want_d_enc=False):
    L = _lib.lib()
    dev = draw.device
    K, B, N = slabs.K, slabs.B, slabs.N
    slabs.dz = torch.empty(K * mlp_stash_bytes(W_OBJ_, B * N), dtype=torch.uint8, device=dev)
    slabs.dz_out = torch.empty(K * int(L.durf_obj_dzout_stride(B, N)), dtype=torch.uint8, device=dev)
    slabs.d_enc = torch.empty(K, B * N, ENC_DIM, device=dev) if want_d_enc else None     # every valid row is written
    with _Timed('obj_bwd_batch'):
        _lib.check(L.durf_obj_bwd_batch(_stream(), K, B, N, _p(idx), _p(count), _p(_f32(draw)), _p(wb), _p(slabs.mask),
                                        _p(slabs.dz), _p(slabs.dz_out), _p(slabs.d_enc)), 'durf_obj_bwd_batch')


def obj_bwd_batch_levels(slabs_levels, idx, count, draws, wb):
    """obj_bwd_batch (no d(enc)) for every level of a step in one call (durf_obj_bwd_batch_levels: one launch at small batches);
    slabs_levels / draws: per level, in the order they are to be issued"""
    L = _lib.lib()
    s0 = slabs_levels[0]
    K, B, N = s0.K, s0.B, s0.N
    nl = len(slabs_levels)
    dev = draws[0].device
    for sl in slabs_levels:
        sl.dz = torch.empty(K * mlp_stash_bytes(W_OBJ_, B * N), dtype=torch.uint8, device=dev)
        sl.dz_out = torch.empty(K * int(L.durf_obj_dzout_stride(B, N)), dtype=torch.uint8, device=dev)
        sl.d_enc = None
    arr = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    with _Timed('obj_bwd_batch'):
        _lib.check(L.durf_obj_bwd_batch_levels(_stream(), K, B, N, nl, _p(idx), _p(count), arr([_f32(d) for d in draws]), _p(wb),
                                               arr([s.mask for s in slabs_levels]), arr([s.dz for s in slabs_levels]),
                                               arr([s.dz_out for s in slabs_levels])), 'durf_obj_bwd_batch_levels')


def obj_dw_batch(slabs_levels, view_tile, count, grad_obj, grad_stride, obj_params):
    """weight gradients of all K object MLPs over every level -> grad_obj (flat, K x grad_stride floats);
    obj_params: their fp32 parameters with the same stride"""
    L = _lib.lib()
    s0 = slabs_levels[0]
    K, B, N = s0.K, s0.B, s0.N
    nl = len(slabs_levels)
    dev = grad_obj.device
    part = torch.empty(K * int(L.durf_dw_part_floats(W_OBJ_)), device=dev)
    bpart = torch.empty(K * int(L.durf_dw_bpart_floats(W_OBJ_)), device=dev)
    arr = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    with _Timed('obj_dw_batch'):
        _lib.check(L.durf_obj_dw_batch(_stream(), K, B, N, _p(count), nl, arr([s.enc for s in slabs_levels]),
                                       arr([view_tile] * nl), arr([s.stash for s in slabs_levels]),
                                       arr([s.dz for s in slabs_levels]), arr([s.dz_out for s in slabs_levels]),
                                       IN_OBJ_, _p(part), _p(bpart), _p(grad_obj), grad_stride, _p(_f32(obj_params))),
                   'durf_obj_dw_batch')


def obj_dw_partials(slabs_levels, view_tile, count):
    """the split-K half of obj_dw_batch: returns (part, bpart) for dw_finalize_all"""
    L = _lib.lib()
    s0 = slabs_levels[0]
    K, B, N = s0.K, s0.B, s0.N
    nl = len(slabs_levels)
    dev = s0.enc.device
    part = torch.empty(K * int(L.durf_dw_part_floats(W_OBJ_)), device=dev)
    bpart = torch.empty(K * int(L.durf_dw_bpart_floats(W_OBJ_)), device=dev)
    arr = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    with _Timed('obj_dw_batch'):
        _lib.check(L.durf_obj_dw_partials(_stream(), K, B, N, _p(count), nl, arr([s.enc for s in slabs_levels]),
                                          arr([view_tile] * nl), arr([s.stash for s in slabs_levels]),
                                          arr([s.dz for s in slabs_levels]), arr([s.dz_out for s in slabs_levels]),
                                          _p(part), _p(bpart)), 'durf_obj_dw_partials')
    return part, bpart


def dw_finalize_all(rows_l, n_l, count_l, part, bpart, grad_bkgd, bkgd_params, obj=None):
    """Finalize the background MLP's weight gradients (segments as in mlp_dw_levels) and, with
    obj = (K, B, N, count, nlevels, part, bpart, grad_obj, grad_stride, obj_params), those of the K object MLPs in the
    same pair of launches (durf_dw_finalize_all)."""
    L, rows_a, n_a, cnt_a = _levels_args(rows_l, n_l, count_l)
    if obj is None:
        oa = (0, 0, 0, None, 1, IN_OBJ_, None, None, None, 0, None)
    else:
        K, B, N, cnt, nl, po, bo, go, gs, pr = obj
        oa = (int(K), int(B), int(N), _p(cnt), int(nl), IN_OBJ_, _p(po), _p(bo), _p(go), int(gs), _p(_f32(pr)))
    with _Timed('mlp_dw_finalize_256'):
        _lib.check(_lib.lib().durf_dw_finalize_all(_stream(), IN_BKGD, L, rows_a, n_a, cnt_a, _p(part), _p(bpart),
                                                   _p(grad_bkgd), _p(_f32(bkgd_params)), *oa), 'durf_dw_finalize_all')


def poison_multi_hit(grad, cls_count, box_floats, K, mlp0_floats, obj_floats, upto=None):
    """reference semantics of rays that hit two boxes (durf_poison_multi_hit): NaN into the gradient segments they touch;
    upto: only the first `upto` floats of the flat buffer are touched"""
    n = grad.numel() if upto is None else int(upto)
    _lib.check(_lib.lib().durf_poison_multi_hit(_stream(), n, _p(_f32(grad)), _p(cls_count), int(box_floats), int(K),
                                                int(mlp0_floats), int(obj_floats)), 'durf_poison_multi_hit')


# Parameter updates go through ctypes data pointers, which torch's tensor version counters do not see.  Whatever caches a
# function of the parameters (MipNerfModel.prefetch_const_trunk) keys it on this generation count instead: every
# in-place update issued from this module bumps the count of the flat buffer it wrote.
_PARAM_GENERATION = {}


def param_generation(params):
    return _PARAM_GENERATION.get((params.data_ptr(), params.numel()), 0)


def _bump_generation(params):
    key = (params.data_ptr(), params.numel())
    _PARAM_GENERATION[key] = _PARAM_GENERATION.get(key, 0) + 1
    if len(_PARAM_GENERATION) > 256:                 # (buffers come and go in tests; the table need not grow with them)
        for k in list(_PARAM_GENERATION)[:128]:
            if k != key:
                del _PARAM_GENERATION[k]


def clip_adam(params, m, v, grad, inv_world, max_val, max_norm, lr, step):
    """In-place Adam step on the flat buffers; returns stats[4] (grad_norm, grad_abs_max,
    clip multiplier, grad_norm_clipped) as a device tensor."""
    _bump_generation(params)
    n = params.numel()
    dev = params.device
    scratch = torch.empty(int(_lib.lib().durf_optim_scratch_floats(n)), device=dev)
    stats = torch.empty(4, device=dev)
    _lib.check(_lib.lib().durf_clip_adam(_stream(), n, _p(_f32(params)), _p(_f32(m)), _p(_f32(v)),
                                         _p(_f32(grad)), inv_world, max_val, max_norm, lr, int(step),
                                         _p(scratch), _p(stats)), 'durf_clip_adam')
    return stats


def encode_obj_bwd(k_obj, idx_k, count_k, d_enc, t_vals, origins_s, dirs_s, radii, origins, dirs, pose, alpha,
                   sums, scratch=None, precise=False, enc_flags=0):
    """accumulates the 21 pose sums of object k (one level) into sums[k] (sums: [K,21], caller-zeroed)"""
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    if scratch is None:
        scratch = torch.empty(21 * B, device=t_vals.device)
    w = barf_weights(alpha)
    wa = (C.c_float * 10)(*[float(x) for x in w])
    _lib.check(_lib.lib().durf_encode_obj_bwd(_stream(), B, N, k_obj, _p(idx_k), _p(count_k), _p(_f32(d_enc)),
                                              _p(_f32(t_vals)), _p(_f32(origins_s)), _p(_f32(dirs_s)),
                                              _p(_f32(radii)), _p(_f32(origins)), _p(_f32(dirs)), _p(_f32(pose)),
                                              wa, _p(scratch), _p(_f32(sums)), int(precise), int(enc_flags)), 'durf_encode_obj_bwd')


def encode_obj_bwd_batch(K, idx, count, d_enc, t_vals, origins_s, dirs_s, radii, origins, dirs, pose, alpha, sums,
                         precise=False, enc_flags=0):
    """all K objects of one level in one launch pair: idx [K,B], count [K], d_enc [K, B*N, 64] (obj_bwd_batch's slab)"""
    B, N = t_vals.shape[0], t_vals.shape[1] - 1
    scratch = torch.empty(K * 21 * B, device=t_vals.device)
    w = barf_weights(alpha)
    wa = (C.c_float * 10)(*[float(x) for x in w])
    _lib.check(_lib.lib().durf_encode_obj_bwd_batch(_stream(), int(K), B, N, _p(idx), _p(count), _p(_f32(d_enc)),
                                                    _p(_f32(t_vals)), _p(_f32(origins_s)), _p(_f32(dirs_s)),
                                                    _p(_f32(radii)), _p(_f32(origins)), _p(_f32(dirs)), _p(_f32(pose)),
                                                    wa, _p(scratch), _p(_f32(sums)), int(precise), int(enc_flags)),
               'durf_encode_obj_bwd_batch')


def pose_finish(pose, sums, want_pos, want_rot, grad6):
    K = pose.shape[0]
    _lib.check(_lib.lib().durf_pose_finish(_stream(), K, _p(_f32(pose)), _p(_f32(sums)), int(want_pos),
                                           int(want_rot), _p(_f32(grad6))), 'durf_pose_finish')


# ---------------------------------------------------------------------------
# exact-fp32 MLP (csrc/mlp_f32.hip): the object branch of a step with box-pose optimisation (MipNerfModel.object_precision)
# and the parity instrument behind MipNerfModel(mlp_precision='f32')
# ---------------------------------------------------------------------------
def mlp_f32_pack(width, in_dim, mlp_params, K=1, param_stride=0, x3=False):
    """fp32 weight streams of K MLPs (durf_mlp_f32_pack): the wide Dense kernels as the chunks the fp32 forward / the
    backward (transposed) consume, in order; re-packed whenever the parameters change.  x3 (W = 128): every weight as a
    (hi, lo) bf16 pair, for the 'bf16x3' object kernels (durf_mlp_f32_pack_x3)"""
    L = _lib.lib()
    out = torch.empty(int(K) * int(L.durf_mlp_f32_wstream_floats(width)), device=mlp_params.device)
    if x3:
        assert width == W_OBJ_ and in_dim == IN_OBJ_
        _lib.check(L.durf_mlp_f32_pack_x3(_stream(), int(K), _p(_f32(mlp_params)), int(param_stride), _p(out)), 'durf_mlp_f32_pack_x3')
        return out
    _lib.check(L.durf_mlp_f32_pack(_stream(), width, in_dim, int(K), _p(_f32(mlp_params)), int(param_stride), _p(out)),
               'durf_mlp_f32_pack')
    return out


def mlp_fwd_f32(width, in_dim, rows, N, enc_f32, view27, mlp_params, ray_idx=None, count=None, want_act=False,
                wstream=None):
    """-> raw [rows,4][, act (opaque record buffer for mlp_bwd_f32 / mlp_dw_f32)].  enc_f32 [rows,in_dim] row-major, or
    None: every row is the background MLP's constant encoding of a box-hit ray (width 256); view27 [B,27]"""
    dev = view27.device
    L = _lib.lib()
    if wstream is None:
        wstream = mlp_f32_pack(width, in_dim, mlp_params)
    raw = torch.zeros(rows, 4, device=dev)
    act = torch.empty(tile_rows(rows) * int(L.durf_mlp_f32_act_floats(width, in_dim)), device=dev) if want_act else None
    with _Timed('mlp_fwd_f32_%d' % width):
        _lib.check(L.durf_mlp_fwd_f32(_stream(), width, in_dim, rows, N, _p(None if enc_f32 is None else _f32(enc_f32)),
                                      _p(_f32(view27)), _p(ray_idx), _p(count), _p(_f32(mlp_params)), _p(wstream), _p(raw),
                                      _p(act)), 'durf_mlp_fwd_f32')
    return (raw, act) if want_act else raw


def mlp_bwd_f32(width, in_dim, rows, N, draw, mlp_params, act, ray_idx=None, count=None, want_d_enc=False, wstream=None):
    """-> dz (opaque record buffer)[, d_enc [rows,64]]"""
    dev = draw.device
    L = _lib.lib()
    if wstream is None:
        wstream = mlp_f32_pack(width, in_dim, mlp_params)
    dz = torch.empty(tile_rows(rows) * int(L.durf_mlp_f32_dz_floats(width, in_dim)), device=dev)
    d_enc = torch.zeros(rows, ENC_DIM, device=dev) if want_d_enc else None
    with _Timed('mlp_bwd_f32_%d' % width):
        _lib.check(L.durf_mlp_bwd_f32(_stream(), width, in_dim, rows, N, _p(_f32(draw)), _p(ray_idx), _p(count),
                                      _p(_f32(mlp_params)), _p(wstream), _p(_f32(act)), _p(dz), _p(d_enc)),
                   'durf_mlp_bwd_f32')
    return (dz, d_enc) if want_d_enc else dz


def mlp_dw_f32(width, in_dim, rows, N, act, dz, grad_mlp, count=None, nsplit=32):
    """weight gradients of one MLP over `rows` samples, ADDED to grad_mlp (flax layout)"""
    dev = act.device
    L = _lib.lib()
    scratch = torch.empty(int(L.durf_mlp_f32_dw_scratch_floats(width, in_dim, nsplit)), device=dev)
    g = torch.empty_like(grad_mlp)
    with _Timed('mlp_dw_f32_%d' % width):
        _lib.check(L.durf_mlp_dw_f32(_stream(), width, in_dim, rows, N, _p(count), _p(_f32(act)), _p(_f32(dz)), nsplit,
                                     _p(scratch), _p(g)), 'durf_mlp_dw_f32')
    grad_mlp += g


def bkgd_const_trunk_f32(bkgd_params):
    """Dense_0 .. Dense_9 of the background MLP on the constant encoding every box-hit ray feeds it ([0 x 30, 1 x 30]), in
    fp32 -> trunk [257] (bottleneck, density): depends on the parameters only, once per step"""
    trunk = torch.empty(257, device=bkgd_params.device)
    _lib.check(_lib.lib().durf_bkgd_const_trunk_f32(_stream(), _p(_f32(bkgd_params)), _p(trunk)), 'durf_bkgd_const_trunk_f32')
    return trunk


def bkgd_hit_rays_f32(B, view27, bkgd_params, idx1, count1, trunk=None):
    """the background MLP's one evaluation of every box-hit ray (ray class 1: idx1 / count1), in fp32 -> raw_tail [B,4]
    (row j = ray idx1[j]): the view layer and the rgb head per ray on top of bkgd_const_trunk_f32's output"""
    dev = view27.device
    if trunk is None:
        trunk = bkgd_const_trunk_f32(bkgd_params)
    raw_tail = torch.empty(B, 4, device=dev)
    with _Timed('bkgd_hit_rays_f32'):
        _lib.check(_lib.lib().durf_bkgd_hit_rays_f32(_stream(), B, _p(_f32(view27)), _p(_f32(bkgd_params)), _p(idx1),
                                                     _p(count1), _p(trunk), _p(raw_tail)), 'durf_bkgd_hit_rays_f32')
    return raw_tail


class ObjSlabsF32:
    """[K, ...] slabs of one level for the batched fp32 object calls (durf_objf32_*; strides fixed by the library)"""

    def __init__(self, K, B, N, device, train):
        L = _lib.lib()
        self.K, self.B, self.N = K, B, N
        self.enc = None                       # only filled by objf32_fwd_batch(fused_encode=False)
        self.raw = torch.empty(K, B * N, 4, device=device)
        self.act = torch.empty(K * int(L.durf_objf32_act_stride(B, N)), device=device) if train else None
        self.dz = self.d_enc = None

    def raws(self):
        return [self.raw[k] for k in range(self.K)]


F32_X3 = 16               # durf_objf32_fwd_batch flag (include/durf_hip.h DURF_F32_X3)


def objf32_fwd_batch(slabs, idx, count, t_vals, origins_s, dirs_s, radii, alpha, view27, obj_params, param_stride, wstream,
                     disable_integration=False, cylinder=False, fused_encode=True, x3=False):
    """accurate fp32 encodings + fp32 forward of all K object MLPs of one level: ONE launch (the forward encodes its own
    tiles); fused_encode=False: durf_encode_obj_f32_batch into slabs.enc first, then the forward reads it (bit-identical)"""
    w = barf_weights(alpha)
    wa = (C.c_float * 10)(*[float(x) for x in w])
    L = _lib.lib()
    flags = (ENC_NO_INTEGRATION if disable_integration else 0) | (ENC_CYLINDER if cylinder else 0) | (F32_X3 if x3 else 0)
    assert fused_encode or not x3
    with _Timed('objf32_fwd_batch'):
        if not fused_encode:
            slabs.enc = torch.empty(slabs.K, slabs.B * slabs.N, IN_OBJ_, device=t_vals.device)
            _lib.check(L.durf_encode_obj_f32_batch(
                _stream(), slabs.K, slabs.B, slabs.N, _p(idx), _p(count), _p(_f32(t_vals)), _p(_f32(origins_s)),
                _p(_f32(dirs_s)), _p(_f32(radii)), wa, flags & ~F32_X3, _p(slabs.enc)), 'durf_encode_obj_f32_batch')
        _lib.check(L.durf_objf32_fwd_batch(_stream(), slabs.K, slabs.B, slabs.N, _p(idx), _p(count),
                                           _p(None if fused_encode else slabs.enc), _p(_f32(view27)), _p(_f32(obj_params)),
                                           int(param_stride), _p(wstream), _p(slabs.raw), _p(slabs.act), _p(_f32(t_vals)),
                                           _p(_f32(origins_s)), _p(_f32(dirs_s)), _p(_f32(radii)), wa, flags),
                   'durf_objf32_fwd_batch')


def objf32_bwd_batch(slabs, idx, count, draw, obj_params, param_stride, wstream, want_d_enc=False, x3=False):
    L = _lib.lib()
    dev = draw.device
    K, B, N = slabs.K, slabs.B, slabs.N
    slabs.dz = torch.empty(K * int(L.durf_objf32_dz_stride(B, N)), device=dev)
    slabs.d_enc = torch.empty(K, B * N, ENC_DIM, device=dev) if want_d_enc else None     # every valid row is written
    with _Timed('objf32_bwd_batch'):
        fn = L.durf_objf32_bwd_batch_x3 if x3 else L.durf_objf32_bwd_batch
        _lib.check(fn(_stream(), K, B, N, _p(idx), _p(count), _p(_f32(draw)), _p(_f32(obj_params)), int(param_stride), _p(wstream),
                      _p(slabs.act), _p(slabs.dz), _p(slabs.d_enc)), 'durf_objf32_bwd_batch')


def objf32_dw_batch(slabs_levels, count, grad_obj, grad_stride, nsplit=8):
    """weight gradients of all K object MLPs over every level -> grad_obj (flat, K x grad_stride floats), overwritten"""
    L = _lib.lib()
    s0 = slabs_levels[0]
    K, B, N = s0.K, s0.B, s0.N
    nl = len(slabs_levels)
    scratch = torch.empty(K * int(L.durf_mlp_f32_dw_scratch_floats(W_OBJ_, IN_OBJ_, nsplit)), device=grad_obj.device)
    arr = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    with _Timed('objf32_dw_batch'):
        _lib.check(L.durf_objf32_dw_batch(_stream(), K, B, N, _p(count), nl, arr([s.act for s in slabs_levels]),
                                          arr([s.dz for s in slabs_levels]), int(nsplit), _p(scratch), _p(grad_obj),
                                          int(grad_stride)), 'durf_objf32_dw_batch')


# ---------------------------------------------------------------------------------------------------------------------
# the whole inference forward as one C call (csrc/forward.hip, include/durf_hip.h durf_forward)
# ---------------------------------------------------------------------------------------------------------------------
FORWARD_MAX_LEVELS = 4
_vp4 = C.c_void_p * FORWARD_MAX_LEVELS


class ForwardArgs(C.Structure):
    """durf_forward_args (include/durf_hip.h), field for field"""
    _fields_ = ([(n, C.c_int) for n in ('B', 'N', 'K', 'num_levels', 'enc_flags', 'lindisp', 'bkgd_mode')] +
                [('density_bias', C.c_float), ('resample_padding', C.c_float), ('barf_w', C.c_float * 10)] +
                [(n, C.c_void_p) for n in ('origins', 'directions', 'viewdirs', 'radii', 'near', 'far', 'pose', 'ext',
                                           'bkgd_params', 'obj_params')] +
                [('obj_param_stride', C.c_size_t), ('t_rand', C.c_void_p), ('u_rand', C.c_void_p)] +
                [(n, _vp4) for n in ('rgb', 'depth', 'acc', 'weights', 't_vals', 't_mids', 't_dists')] +
                [('dyn_mask', C.c_void_p), ('zo', C.c_void_p), ('draw_noise', C.c_int), ('seed_lo', C.c_uint32),
                 ('seed_hi', C.c_uint32), ('density_noise', C.c_float), ('density_rand', _vp4)])


class TrainArgs(C.Structure):
    """durf_train_args (include/durf_hip.h), field for field"""
    _fields_ = ([('f', ForwardArgs)] +
                [(n, C.c_void_p) for n in ('lossmult', 'pixels', 'gt_depth', 'sky', 'target6', 'prev6')] +
                [('eps', C.c_float), ('box_loss_mult', C.c_float), ('bg', C.c_float), ('disable_multiscale', C.c_int),
                 ('level_mults', (C.c_float * 6) * FORWARD_MAX_LEVELS), ('stat_mults', C.c_float * 6), ('params', C.c_void_p)] +
                [(n, C.c_size_t) for n in ('n_params', 'box_floats', 'mlp0_floats', 'obj_floats')] +
                [(n, C.c_void_p) for n in ('grad', 'stats', 'adam_m', 'adam_v')] +
                [('lr', C.c_float), ('max_val', C.c_float), ('max_norm', C.c_float), ('step', C.c_int), ('grad_stats', C.c_void_p),
                 ('flags', C.c_int), ('want_pos', C.c_int), ('want_rot', C.c_int), ('tv_loss_mult', C.c_float),
                 ('comm', C.c_void_p), ('world', C.c_int), ('reduce_stats', C.c_int), ('weight_decay_mult', C.c_float),
                 ('pose_used', C.c_void_p), ('cls_count', C.c_void_p), ('timing', C.c_void_p),
                 ('const_trunk', C.c_void_p), ('const_trunk_valid', C.c_int), ('prefetch_const_trunk', C.c_int)])


TRAIN_OBJ_FP32, TRAIN_POSE_OPT, TRAIN_OBJ_X3 = 1, 2, 4          # durf_train_args.flags
TIMED_FWD, TIMED_BWD, TIMED_COMPOSITE, TIMED_DW, TIMED_STAGES = 0, 4, 8, 12, 13      # durf_step_timing slots


class StepTiming(C.Structure):
    """durf_step_timing (include/durf_hip.h)"""
    _fields_ = [('begin', C.c_void_p * TIMED_STAGES), ('end', C.c_void_p * TIMED_STAGES)]


def _step_timing(num_levels, keep):
    """the live timers of a one-call step (bench.py's roofline): HIP events the C call records around the launches the
    Python-issued path brackets with _Timed -- the same names in TIMERS.  None unless timers are armed for this step."""
    if TIMERS is None or not TIMERS_ACTIVE:
        return None
    want = lambda name: TIMED_NAMES is None or name in TIMED_NAMES
    slots = []
    if want('mlp_fwd_256_train'):
        slots += [('mlp_fwd_256_train', TIMED_FWD + l) for l in range(num_levels)]
    if want('mlp_bwd_256'):
        slots += [('mlp_bwd_256', TIMED_BWD + l) for l in range(num_levels)]
    if want('composite_resample'):
        slots += [('composite_resample', TIMED_COMPOSITE + l) for l in range(num_levels - 1)]
    if want('mlp_dw_256'):
        slots += [('mlp_dw_256', TIMED_DW)]
    if not slots:
        return None
    tm = StepTiming()
    for name, i in slots:
        pair = (_timing_event(), _timing_event())
        tm.begin[i], tm.end[i] = pair[0].cuda_event, pair[1].cuda_event
        TIMERS.setdefault(name, []).append(pair)
    keep.append(tm)
    return tm


def _fill_forward_args(a, rays, pose, ext, bkgd_params, obj_params, obj_param_stride, N, num_levels, alpha, enc_flags, lindisp,
                       bkgd_mode, density_bias, resample_padding, t_rand, u_rand, outs, dyn, zo, keep, seed=None,
                       density_noise=0.0, density_rand=None):
    B, K = rays.origins.shape[0], pose.shape[0]
    a.B, a.N, a.K, a.num_levels, a.enc_flags, a.lindisp, a.bkgd_mode = B, N, K, num_levels, enc_flags, int(lindisp), bkgd_mode
    a.density_bias, a.resample_padding = density_bias, resample_padding
    a.barf_w = (C.c_float * 10)(*[float(x) for x in barf_weights(alpha)])
    flat = [t.reshape(-1).contiguous() for t in (rays.radii, rays.near, rays.far)]
    keep.extend(flat)
    a.origins, a.directions, a.viewdirs = _p(_f32(rays.origins)), _p(_f32(rays.directions)), _p(_f32(rays.viewdirs))
    a.radii, a.near, a.far = (_p(_f32(t)) for t in flat)
    a.pose, a.ext = (_p(_f32(pose)) if K else None), (_p(_f32(ext)) if K else None)
    a.bkgd_params = _p(_f32(bkgd_params))
    a.obj_params, a.obj_param_stride = (_p(_f32(obj_params)) if K else None), int(obj_param_stride)
    a.t_rand, a.u_rand = _p(t_rand), _p(u_rand)
    a.draw_noise = int(seed is not None)
    a.seed_lo, a.seed_hi = (0, 0) if seed is None else split_seed(seed)
    a.density_noise = float(density_noise)
    if density_noise and density_rand is not None:          # the caller's standard-normal draws, [B,N] per level
        dr = [_f32(t.reshape(-1).contiguous()) for t in density_rand]
        assert len(dr) == num_levels and all(t.numel() == B * N for t in dr)
        keep.extend(dr)
        a.density_rand = _vp4(*([t.data_ptr() for t in dr] + [None] * (FORWARD_MAX_LEVELS - num_levels)))
    if outs is not None:          # (durf_render_image keeps the per-chunk outputs in its workspace)
        for i, name in enumerate(('rgb', 'depth', 'acc', 'weights', 't_vals', 't_mids', 't_dists')):
            setattr(a, name, _vp4(*([o[i].data_ptr() for o in outs] + [None] * (FORWARD_MAX_LEVELS - num_levels))))
    a.dyn_mask, a.zo = _p(dyn), _p(zo)


def forward_call(rays, pose, ext, bkgd_params, obj_params, obj_param_stride, N, num_levels, alpha, enc_flags, lindisp=False,
                 bkgd_mode=BKGD_GREY, density_bias=-1.0, resample_padding=0.01, t_rand=None, u_rand=None, seed=None,
                 density_noise=0.0, density_rand=None):
    """MipNerfModel.__call__ in inference as ONE library call (durf_forward): -> list[num_levels] of
    (rgb, depth, acc, weights, t_vals, t_mids, t_dists), dyn_mask [B,1] int32, zo [B]"""
    B, K = rays.origins.shape[0], pose.shape[0]
    dev = rays.origins.device
    L = _lib.lib()
    f = lambda *sh: torch.empty(*sh, device=dev)
    outs = [(f(B, 3), f(B), f(B), f(B, N), f(B, N + 1), f(B, N), f(B, N)) for _ in range(num_levels)]
    dyn, zo = torch.empty(B, 1, dtype=torch.int32, device=dev), f(B)
    a = ForwardArgs()
    keep = []
    _fill_forward_args(a, rays, pose, ext, bkgd_params, obj_params, obj_param_stride, N, num_levels, alpha, enc_flags, lindisp,
                       bkgd_mode, density_bias, resample_padding, t_rand, u_rand, outs, dyn, zo, keep, seed=seed,
                       density_noise=density_noise, density_rand=density_rand)
    ws = _workspace(dev, int(L.durf_forward_workspace_bytes(B, N, K)))
    with _Timed('forward_call'):
        _lib.check(L.durf_forward(_stream(), C.byref(a), _p(ws), ws.numel()), 'durf_forward')
    return outs, dyn, zo


def render_image_call(rays, pose, ext, bkgd_params, obj_params, obj_param_stride, N, num_levels, alpha, enc_flags, chunk,
                      lindisp=False, bkgd_mode=BKGD_GREY, density_bias=-1.0, resample_padding=0.01):
    """render_image on one device as ONE library call (durf_render_image: the chunk loop in C over the ray buffer resident on
    the device) -> rgb [n,3], distance [n], acc [n] of the last level; rays: flattened [n, .] fields of the whole image"""
    n, K = rays.origins.shape[0], pose.shape[0]
    dev = rays.origins.device
    L = _lib.lib()
    rgb, dist_, acc = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    a = ForwardArgs()
    keep = []
    rays = type(rays)(*[t.contiguous() for t in rays])
    _fill_forward_args(a, rays, pose, ext, bkgd_params, obj_params, obj_param_stride, N, num_levels, alpha, enc_flags, lindisp,
                       bkgd_mode, density_bias, resample_padding, None, None, None, None, None, keep)
    a.B = min(chunk, n)
    ws = _workspace(dev, int(L.durf_render_image_workspace_bytes(min(chunk, n), N, K, num_levels)))
    with _Timed('render_image_call'):
        _lib.check(L.durf_render_image(_stream(), C.byref(a), n, min(chunk, n), _p(rgb), _p(dist_), _p(acc), _p(ws), ws.numel()),
                   'durf_render_image')
    return rgb, dist_, acc


_WORKSPACE = {}


def _workspace(dev, nbytes):
    """the one-call entry points' workspace: ONE buffer per device, kept across calls and grown when a call needs more (a
    fresh torch.empty of several hundred MB per step leaves it to the caching allocator to hand the same block back; when it
    does not, the step stalls on a device allocation -- seen as one 0.5 ms step in ~8 passes of a 0.4 ms workload)"""
    # (per device AND per stream: two one-call entry points issued on different streams of one device -- an eval render beside
    # training, a second model -- must not share intermediates; calls on one stream are ordered and may)
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream if dev.type == 'cuda' else 0)
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes:
        _WORKSPACE[key] = ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        assert ws.data_ptr() % 256 == 0
    return ws


def release_workspace():
    """drop the cached workspaces of the one-call entry points (a large render chunk otherwise pins its buffer -- hundreds
    of MB -- for the life of the process); the next call allocates afresh"""
    _WORKSPACE.clear()


def train_call(rays, pose, ext, params_flat, m, v, box_floats, mlp0_floats, obj_floats, N, num_levels, alpha, enc_flags,
               lossmult, pixels, gt_depth, sky, target6, prev6, eps, box_loss_mult, bg, disable_multiscale, level_mults,
               stat_mults, lr, max_val, max_norm, step, lindisp=False, bkgd_mode=BKGD_GREY, density_bias=-1.0,
               resample_padding=0.01, t_rand=None, u_rand=None, update=True, obj_fp32=False, obj_x3=False, want_pos=False, want_rot=False,
               tv_loss_mult=0.0, seed=None, comm=None, world=1, reduce_stats=False, density_noise=0.0, density_rand=None,
               weight_decay_mult=0.0, const_trunk=None, const_trunk_valid=False):
    """One shard's training step as ONE library call (durf_train_step; update=False: durf_loss_backward, parameters
    untouched) -> (per-level outputs, dyn_mask, zo, grad, stats buffer, grad_stats or None, pose_used [K,6] or None -- the
    poses the step rendered with -- and the class counts [8] int32 or None: [3] = rays that hit two boxes).
    obj_fp32: the object branch on the exact-fp32 kernels; want_pos / want_rot: box-pose optimisation behind it (`pose` must
    then be a view of this timestep's rows of box_centers inside params_flat).  const_trunk [264] (obj_fp32): the caller-kept
    constant trunk -- used when const_trunk_valid (the caller vouches for the parameters), refilled for the NEXT step behind the
    update (durf_train_args.prefetch_const_trunk)"""
    if update:
        _bump_generation(params_flat)
    B, K = rays.origins.shape[0], pose.shape[0]
    dev = rays.origins.device
    L = _lib.lib()
    f = lambda *sh: torch.empty(*sh, device=dev)
    outs = [(f(B, 3), f(B), f(B), f(B, N), f(B, N + 1), f(B, N), f(B, N)) for _ in range(num_levels)]
    dyn, zo = torch.empty(B, 1, dtype=torch.int32, device=dev), f(B)
    grad, stats, gstats = torch.empty_like(params_flat), f(2 + 17 * num_levels), f(4)
    a = TrainArgs()
    keep = []
    o0 = box_floats + mlp0_floats
    _fill_forward_args(a.f, rays, pose, ext, params_flat[box_floats:o0], params_flat[o0:] if K else None, obj_floats, N, num_levels,
                       alpha, enc_flags, lindisp, bkgd_mode, density_bias, resample_padding, t_rand, u_rand, outs, dyn, zo, keep, seed=seed,
                       density_noise=density_noise, density_rand=density_rand)
    a.weight_decay_mult = float(weight_decay_mult)
    hold = [t.reshape(-1).contiguous() for t in (lossmult, gt_depth, sky)] + [pixels.contiguous()]
    a.lossmult, a.gt_depth, a.sky, a.pixels = (_p(_f32(t)) for t in hold)
    hold += [target6.contiguous(), prev6.contiguous()] if K else []
    a.target6, a.prev6 = (_p(_f32(hold[-2])), _p(_f32(hold[-1]))) if K else (None, None)
    a.eps, a.box_loss_mult, a.bg, a.disable_multiscale = float(eps), float(box_loss_mult), float(bg), int(disable_multiscale)
    for lvl in range(num_levels):
        for i in range(6):
            a.level_mults[lvl][i] = float(level_mults[lvl][i])
    a.stat_mults = (C.c_float * 6)(*[float(x) for x in stat_mults])
    a.params, a.n_params = _p(_f32(params_flat)), params_flat.numel()
    a.box_floats, a.mlp0_floats, a.obj_floats = int(box_floats), int(mlp0_floats), int(obj_floats)
    a.grad, a.stats, a.grad_stats = _p(grad), _p(stats), _p(gstats)
    a.adam_m, a.adam_v = _p(_f32(m)), _p(_f32(v))
    a.lr, a.max_val, a.max_norm, a.step = float(lr), float(max_val), float(max_norm), int(step)
    pose_opt = bool(K) and (want_pos or want_rot)
    if pose_opt and not obj_fp32:
        raise NotImplementedError('durf_train_step: DURF_TRAIN_POSE_OPT needs DURF_TRAIN_OBJ_FP32 (csrc/train.hip)')
    a.flags = ((TRAIN_OBJ_FP32 if (K and obj_fp32) else 0) | (TRAIN_POSE_OPT if pose_opt else 0) |
               (TRAIN_OBJ_X3 if (K and obj_fp32 and obj_x3) else 0))
    a.want_pos, a.want_rot, a.tv_loss_mult = int(bool(want_pos)), int(bool(want_rot)), float(tv_loss_mult)
    a.comm, a.world, a.reduce_stats = (comm.handle if comm is not None else None), int(world), int(bool(reduce_stats))
    pose_used = torch.empty_like(pose) if K else None
    cls = torch.empty(8, dtype=torch.int32, device=dev) if K else None
    a.pose_used, a.cls_count = _p(pose_used), _p(cls)
    if const_trunk is not None and K and obj_fp32:
        assert const_trunk.numel() >= 264 and const_trunk.is_contiguous()
        a.const_trunk, a.const_trunk_valid, a.prefetch_const_trunk = _p(_f32(const_trunk)), int(bool(const_trunk_valid)), int(bool(update))
    tm = _step_timing(num_levels, keep) if update else None
    a.timing = C.cast(C.pointer(tm), C.c_void_p) if tm is not None else None
    ws = _workspace(dev, int(L.durf_train_workspace_bytes_flags(B, N, K, num_levels, params_flat.numel(), a.flags)))
    with _Timed('train_call'):
        fn = L.durf_train_step if update else L.durf_loss_backward
        _lib.check(fn(_stream(), C.byref(a), _p(ws), ws.numel()), 'durf_train_step' if update else 'durf_loss_backward')
    return outs, dyn, zo, grad, stats, (gstats if update else None), pose_used, cls
