"""One optimisation step on MI355X -- host-side mirror of train_boxpose.py:49-321.

`train_step(model, config, rng, state, batch, lr, eps, alpha, prev)` keeps the reference's
signature and return value `(new_state, stats, rng, pose)`.  Forward, losses, backward and
clip+Adam all run as HIP kernels through the C ABI; the data-parallel exchange is ONE
all-reduce (RCCL over xGMI via torch.distributed) of the flat gradient buffer, replacing
`jax.lax.pmean(grad, 'batch')` (train_boxpose.py:253), plus a ~40-float all-reduce of the
scalar stats replacing `pmean(stats)` (:255; the reference also averages the logged
weights/samples tensors, which nothing downstream needs).
"""
import dataclasses
import math
import struct
from typing import Any

import torch

from . import math as dmath
from . import obbpose_model as om
from . import ops
from . import utils


@dataclasses.dataclass
class TrainState:
    """utils.TrainState (internal/utils.py:37-39) + flax.optim.Adam state, flattened."""
    variables: Any          # obbpose_model.Variables (flat fp32 params)
    m: torch.Tensor         # Adam first moment, flat
    v: torch.Tensor         # Adam second moment, flat
    step: int = 0           # optimizer.state.step


def create_train_state(variables):
    """flax.optim.Adam(lr).create(variables) (train_boxpose.py:343-344)."""
    return TrainState(variables, torch.zeros_like(variables.flat), torch.zeros_like(variables.flat), 0)


def _force_dist():
    """DURF_FORCE_DIST=1: take the data-parallel code path (RCCL all-reduce of the gradient, stream ordering behind
    it, inv_world, stats cadence) even with ONE rank -- a world-size-1 `nccl` group; how the collective path is
    exercised and timed on a one-GPU box (tests/test_gpu_dist.py, bench.py --force-dist)"""
    import os
    return os.environ.get('DURF_FORCE_DIST', '0') != '0'


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force_dist()):
        return dist
    return None


# DURF_BUCKET_ALLREDUCE=1 (multi-rank only): the K object MLPs' gradients are final before the background MLP's
# weight-gradient launch (the longest kernel of a step) starts, so their slice of the flat buffer is all-reduced behind
# it and only [box_centers | MLP_0] after it -- two collectives instead of one, the first one free.  Costs a separate
# finalize launch for the objects (~30 us); off by default until it has been measured on a multi-GPU node.
def _bucketed():
    import os
    return os.environ.get('DURF_BUCKET_ALLREDUCE', '0') != '0'


# The gradient all-reduce of a multi-rank step (lax.pmean(grad), train_boxpose.py:253-255,370-374) is issued by the library
# itself IN the compute stream (csrc/comm.hip: its own RCCL communicator, the unique id handed round through
# torch.distributed's store) instead of by torch.distributed on its communication stream -- no event hop there and back
# (~22 us of idle GPU per step on a world-size-1 group, profiles/r05_rccl_instream.txt: 3.3 % of cfg5's per-rank step) --
# and the step then runs as the ONE C call durf_train_step(args.comm), the sequence a host that is not Python gets.
# Round 6: this is the DEFAULT route of an `nccl` process group, SELF-VERIFIED when the communicator is created: a known
# vector (rank + 1, and a rank-dependent ramp) is all-reduced through the library's communicator AND through c10d, both are
# compared with the closed form and with each other, and the ranks agree on the verdict (a MIN all-reduce through c10d): any
# failure on any rank -> every rank logs once and stays on the c10d route, in the same process.  DURF_INSTREAM_ALLREDUCE=0
# opts out.  COLLECTIVE_INFO records what happened (bench.py prints it in its JSON line: `collective.route`, `ranks_seen`).
_INSTREAM = {}
COLLECTIVE_INFO = {}


def _selfcheck_vector(rank, n, dev):
    return (rank + 1) + torch.arange(n, device=dev, dtype=torch.float32) * (rank + 1) / 1024.0


def _instream_comm(dist):
    import os
    if 'comm' in _INSTREAM:
        return _INSTREAM['comm']
    _INSTREAM['comm'] = None
    info = COLLECTIVE_INFO
    info.update(backend=dist.get_backend(), world=dist.get_world_size(), route='torch.distributed (c10d) all-reduce on its own stream')
    if os.environ.get('DURF_INSTREAM_ALLREDUCE', '1') == '0':
        info['why'] = 'DURF_INSTREAM_ALLREDUCE=0'
        return None
    if dist.get_backend() != 'nccl':
        info['why'] = 'backend %s: the in-stream route is RCCL only' % dist.get_backend()
        return None
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', torch.cuda.current_device())
    comm, err = None, None
    try:
        if not ops.comm_available():
            raise RuntimeError('no RCCL resolvable by the library: ' + ops._lib.lib().durf_last_error().decode())
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set('durf_rccl_unique_id', ops.comm_unique_id())
        comm = ops.Comm(world, rank, store.get('durf_rccl_unique_id'))
        n = 4096
        mine = _selfcheck_vector(rank, n, dev)
        a, b = mine.clone(), mine.clone()
        comm.all_reduce_sum(a)                      # the library's communicator, in the compute stream
        dist.all_reduce(b)                          # c10d
        want = sum(_selfcheck_vector(r, n, dev).double() for r in range(world))
        torch.cuda.synchronize()
        tol = 1e-6 * float(want.abs().max())
        if not (torch.equal(a, b) or float((a - b).abs().max()) <= tol):
            raise RuntimeError('the library\'s all-reduce and c10d\'s disagree (max |diff| %g)' % float((a - b).abs().max()))
        if float((a.double() - want).abs().max()) > tol:
            raise RuntimeError('the all-reduced self-check vector is not the sum over %d ranks' % world)
        info['ranks_seen'] = int(round(float(a[0]) * 2.0 / (world + 1))) if world > 0 else 0      # sum(rank + 1) = w (w + 1) / 2
    except Exception as e:                          # noqa: BLE001 -- whatever went wrong, the c10d route still works
        err = '%s: %s' % (type(e).__name__, e)
    ok = torch.tensor([0.0 if err else 1.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # every rank takes the same route
    if float(ok) < 1.0:
        info['why'] = err or 'another rank failed the self-check'
        if comm is not None:
            try:
                comm.destroy()
            except Exception:                       # noqa: BLE001
                pass
        if rank == 0 or err:
            import sys
            print('durf: in-stream all-reduce not used (%s); staying on torch.distributed' % info['why'], file=sys.stderr)
        return None
    info['route'] = 'in-stream RCCL all-reduce issued by libdurf_hip.so (self-verified against c10d at init)'
    _INSTREAM['comm'] = comm
    return comm


def shutdown_instream():
    """destroy the in-stream communicator (before torch.distributed's process group goes)"""
    c = _INSTREAM.pop('comm', None)
    if c is not None:
        torch.cuda.synchronize()
        c.destroy()


def _cpus_near_gpu(local, n_local, avail, sysfs='/sys/class/drm'):
    """the host cores rank `local` of `n_local` ranks on this node should run on: the cores sysfs lists as local to its GPU
    (the local_cpulist of the local-th AMD render device by PCI address), shared out among the ranks that list the same
    cores; without that information an even split of the cores the process may use.  Pure function of its arguments + sysfs."""
    import os
    avail = sorted(avail)
    lists = []
    try:
        devs = []
        for c in sorted(os.listdir(sysfs)):
            d = os.path.join(sysfs, c, 'device')
            if not c.startswith('renderD') or not os.path.exists(os.path.join(d, 'vendor')):
                continue
            if open(os.path.join(d, 'vendor')).read().strip() != '0x1002':
                continue
            devs.append((os.path.basename(os.path.realpath(d)), open(os.path.join(d, 'local_cpulist')).read().strip()))
        for _, txt in sorted(devs):
            cpus = set()
            for part in filter(None, txt.split(',')):
                lo, _, hi = part.partition('-')
                cpus.update(range(int(lo), int(hi or lo) + 1))
            lists.append(sorted(cpus & set(avail)))
    except (OSError, ValueError):
        lists = []
    if len(lists) >= n_local and lists[local]:
        mates = [r for r in range(n_local) if lists[r] == lists[local]]
        mine = lists[local]
        k, per = mates.index(local), max(len(mine) // len(mates), 1)
        return mine[k * per:(k + 1) * per] or mine
    per = max(len(avail) // max(n_local, 1), 1)
    return avail[local * per:(local + 1) * per] or avail


def pin_host_thread(local, n_local):
    """os.sched_setaffinity for this rank, before its first GPU call (DURF_PIN_CPUS=0: leave the affinity alone)"""
    import os
    if n_local <= 1 or os.environ.get('DURF_PIN_CPUS', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        cpus = _cpus_near_gpu(local, n_local, os.sched_getaffinity(0))
        if len(cpus) < 4:           # (the runtime's own threads -- RCCL's proxy, HIP's signal handler -- inherit the mask: a rank
            return None             # squeezed onto fewer than four cores is worse off than an unpinned one)
        os.sched_setaffinity(0, cpus)
        COLLECTIVE_INFO['host_cpus'] = '%d-%d (%d)' % (cpus[0], cpus[-1], len(cpus))
        return cpus
    except OSError:
        return None


def level_multipliers(config, level, num_levels):
    """Multipliers of one level's terms in the total loss (train_boxpose.py:211-220), in the
    order durf_loss_bwd expects: rgb, sky, depth, near, empty, distortion."""
    last = level == num_levels - 1
    return [1.0 if last else config.coarse_loss_mult,
            (10.0 if last else 1.0) * config.sky_loss_mult,
            (1.0 if last else 0.1) * config.depth_loss_mult,
            (1.0 if last else 0.1) * config.near_loss_mult,
            (1.0 if last else 0.1) * config.empty_loss_mult,
            0.000001]


def loss_and_grad(model, config, rng, variables, batch, eps, alpha, prev, noise=None, objects_ready=None, defer_poison=False):
    """value_and_grad(loss_fn) (train_boxpose.py:67-252) for this rank's shard.
    Returns (grad_flat, raw stats dict of device tensors, pose).
    objects_ready(grad_slice): called as soon as the K object MLPs' gradients are final (before the background MLP's
    weight-gradient launch is issued) -- the hook of the bucketed all-reduce; forces the objects' own finalize launch.
    defer_poison: the multi-hit outcome (ops.poison_multi_hit) is NOT applied here; raw['poison'] carries its arguments for
    the optimizer's first launch (ops.stats_scrub; single device only -- an all-reduce has to see the NaNs)."""
    pose_opt = not (model.no_pose_opt and model.no_yaw_opt)
    # use_viewdirs=False: the step runs on the 12-Dense embedding of the 10-Dense tree (durf_amd/noview.py); the gradient of
    # the real parameters is read back from it below, before anything that works on the real buffer (weight decay)
    real = variables
    variables = model._kernel_variables(variables)
    rays = batch['rays']
    L = model.num_levels
    dev = variables.flat.device
    lossmult = rays.lossmult.reshape(-1).contiguous()
    gt_depth = batch['depth'].reshape(-1).contiguous()
    sky = batch['sky'].reshape(-1).contiguous()
    norms = torch.empty(L, ops.PREP_ROWS, device=dev)
    # with >= 2 levels the forward's fused per-ray launches also compute the loss normalisers (durf_loss_prep's job)
    prep = dict(lossmult=lossmult, gt_depth=gt_depth, sky=sky, eps=float(eps), box_loss_mult=float(config.box_loss_mult),
                disable_multiscale=config.disable_multiscale_loss, norms=norms) if L >= 2 else None
    # zero filled by the forward's first launch (durf_ray_prologue): the gradient and, behind it, the pose sums of a
    # pose-optimisation step (one buffer: no fill launch of their own)
    n_par = variables.flat.numel()
    n_sums = 21 * max(int(variables.layout.K), 1) if (pose_opt and variables.layout.K) else 0
    zbuf = torch.empty(n_par + n_sums, device=variables.flat.device)
    grad = zbuf[:n_par]
    ret, ctx = model._forward(variables, rng, rays, batch['init'], batch['ext'], batch['ts'],
                              config.randomized, config.rand_bkgd, config.white_bkgd, alpha, train=True,
                              noise=noise, loss_prep=prep, zero_fill=zbuf)
    B, N, K = ctx['B'], ctx['N'], ctx['K']
    lay = variables.layout
    rows = B * N
    pixels = batch['pixels'][..., :3].contiguous()
    dyn = ret[0][8].reshape(-1).to(torch.int32).contiguous()
    bg = 0.0 if config.rand_bkgd else (1.0 if config.white_bkgd else 0.5)
    f32 = model.mlp_precision == 'f32'
    obj_f32 = ctx['obj_f32']                    # object branch on the exact-fp32 kernels (MipNerfModel.object_precision)
    Kb = 0 if obj_f32 else K                    # objects on the bf16 kernels
    bufs = None if f32 else ops.dw_buffers(om.W_BKGD, dev)
    dzs = [None] * L                            # per-level (dz, dz_out) of the bkgd MLP, consumed by ONE dW launch
    dd = ctx.get('dedup')                        # de-duplicated background evaluation (obbpose_model._forward)
    # Side stream (ops.overlap_mode; large batches only -- a fork / join is one more dependency in a latency-bound step): the
    # object backward / weight gradients and, when the forward did not write it, the view-direction tile (read by the
    # weight-gradient launch only).  (Until round 5 also the loss launches of the levels below the last: one launch for every
    # level on this stream measured better -- two cross-stream hops fewer.)
    side = ops.on_side(dev, not f32 and ops.overlap_backward(rows))
    main = torch.cuda.current_stream() if side.enabled else None
    obj_side = side if Kb else ops.on_side(dev, False)

    def make_view_tile():
        if dd is not None:
            return ops.expand_view(rows, N, ctx['view'], ray_idx=dd['idx'][0], count=dd['count'][0:1],
                                   tail_idx=dd['idx'][1], tail_count=dd['count'][1:2])
        return ops.expand_view(rows, N, ctx['view'])

    ray_sums = torch.empty(L, B, 4, device=dev) if dd is not None else None
    sums = torch.empty(L, ops.TERM_ROWS, device=dev)      # filled by the stats launch from the per-ray terms (ops.train_stats)
    terms = [None] * L
    radii = rays.radii.reshape(-1).contiguous()
    obj_enc_flags = ((ops.ENC_NO_INTEGRATION if model.disable_integration else 0) |          # the pose gradient's way back
                     (ops.ENC_CYLINDER if model.ray_shape == 'cylinder' else 0))              # through the object encoding
    pose_ts = variables['params']['box_centers'][ctx['ts']].contiguous()
    pose_sums = zbuf[n_par:].view(-1, 21) if n_sums else (torch.zeros(max(K, 1), 21, device=dev) if pose_opt else None)

    def level_loss(lvl):
        lv = ctx['levels'][lvl]
        out = (lv['rgb'], lv['depth'], lv['acc'], lv['weights'], lv['t_mids'], lv['t_dists']) if lv['deferred'] else None
        return ops.loss_bwd(lv['raw_b'], lv['raws'], ctx['slot'], lv['t_vals'], ctx['d_s'], pixels, lossmult,
                            gt_depth, sky, dyn, ctx['zo'], norms[lvl], float(eps),
                            level_multipliers(config, lvl, L), float(config.box_loss_mult), lvl, bg,
                            model.density_bias, config.disable_multiscale_loss, render_out=out,
                            draw_ray_sum=None if dd is None else ray_sums[lvl], defer_sums=True)

    draws = [None] * L
    view_tile, view_ready = ctx.get('view_tile'), None       # (written by the level-0 forward when it encodes its own tiles)
    if side.enabled and view_tile is None:                   # (a forward that does not: the tile is made beside the backward)
        side.fork()
        with side:
            view_tile = make_view_tile()
            view_tile.record_stream(main)
            view_ready = torch.cuda.Event()
            view_ready.record(side.side)
    elif not f32 and view_tile is None:
        view_tile = make_view_tile()
    if prep is not None and L >= 2:
        # every level's loss + composite backward as ONE launch in front of the backward kernels (stop_level_grad: each is a
        # function of the forward alone) instead of one launch per level between them -- on this stream at every batch size
        # (round 5: the lower levels' launches on the side stream cost two more cross-stream hops than they hid)
        lvs = []
        for lvl in range(L):
            lv = ctx['levels'][lvl]
            lvs.append(dict(raw_bkgd=lv['raw_b'], raw_obj=lv['raws'], t_vals=lv['t_vals'], norm=norms[lvl],
                            mults=level_multipliers(config, lvl, L), level=lvl,
                            render_out=(lv['rgb'], lv['depth'], lv['acc'], lv['weights'], lv['t_mids'], lv['t_dists'])
                            if lv['deferred'] else None,
                            draw_ray_sum=None if dd is None else ray_sums[lvl]))
        res = ops.loss_bwd_levels(lvs, ctx['slot'], ctx['d_s'], pixels, lossmult, gt_depth, sky, dyn, ctx['zo'], float(eps),
                                  float(config.box_loss_mult), bg, model.density_bias, config.disable_multiscale_loss)
        for lvl in range(L):
            draws[lvl], terms[lvl] = res[lvl]
    # the object backward of EVERY level in front of the background's (one launch at small batches: durf_obj_bwd_batch_levels)
    # whenever all the levels' d(raw) exist already and no d(enc) is wanted (the pose gradient behind bf16 objects takes the
    # per-level kernel with the d(enc) epilogue)
    obj_bwd_done = bool(Kb) and not pose_opt and not f32 and all(d is not None for d in draws)
    # ... or, at a small step on one stream, level by level as items of the background backward's persistent launch
    # (durf_mlp_bwd_obj, round 6: bit-identical to the launches of their own)
    obj_bwd_mixed = obj_bwd_done and dd is not None and not obj_side.enabled and ops.obj_mix(rows)
    if obj_bwd_done and not obj_bwd_mixed:
        obj_side.fork()
        with obj_side:
            if obj_side.enabled:
                for d in draws:
                    d.record_stream(obj_side.side)
            order = list(reversed(range(L)))
            ops.obj_bwd_batch_levels([ctx['levels'][l]['slabs'] for l in order], ctx['idx'], ctx['count'],
                                     [draws[l] for l in order], ctx['packs']['obj'][1])
    # last level first: its loss kernel also fills that level's rendered outputs (ret[-1]) when the forward deferred them
    for lvl in reversed(range(L)):
        lv = ctx['levels'][lvl]
        if prep is None:
            ops.loss_prep(lv['t_vals'], lossmult, gt_depth, sky, dyn, ctx['zo'], float(eps),
                          float(config.box_loss_mult), lvl, config.disable_multiscale_loss, norm=norms[lvl])
        if draws[lvl] is None:
            draw, terms[lvl] = level_loss(lvl)
        else:
            draw = draws[lvl]
        if obj_side.enabled:
            draw.record_stream(obj_side.side)     # (read by the object backward on the side stream)
        if f32:                               # exact-fp32 parity instrument: per-MLP fp32 backward + weight gradients
            fl = lv['f32']
            off = lay.mlp_off['MLP_0']
            dz = ops.mlp_bwd_f32(om.W_BKGD, om.IN_BKGD, rows, N, draw, variables.mlp_flat('MLP_0'), fl['act_b'],
                                 wstream=ctx['bkgd_ws'])
            ops.mlp_dw_f32(om.W_BKGD, om.IN_BKGD, rows, N, fl['act_b'], dz, grad[off:off + lay.mlp_size[om.W_BKGD]])
        if obj_f32:                           # the object branch in fp32: backward + d(enc) -> pose sums, all K at once
            sl = lv['f32']['slabs32']
            o0, sz = lay.mlp_off['BoxMLP_0'], lay.mlp_size[om.W_OBJ]
            ops.objf32_bwd_batch(sl, ctx['idx'], ctx['count'], draw, variables.flat[o0:o0 + K * sz], sz, ctx['obj_ws'],
                                 want_d_enc=pose_opt, x3=ctx.get('obj_x3', False))
            if pose_opt:
                ops.encode_obj_bwd_batch(K, ctx['idx'], ctx['count'], sl.d_enc, lv['t_vals'], ctx['o_s'], ctx['d_s'], radii,
                                         rays.origins, rays.directions, pose_ts, alpha, pose_sums, precise=True,
                                         enc_flags=obj_enc_flags)
        if f32:
            continue
        if not obj_bwd_done:
            obj_side.fork()                  # the object backward runs in the shadow of the background backward
        if obj_bwd_mixed:
            dzs[lvl] = ops.mlp_bwd_obj(rows, N, draw, ctx['packs']['MLP_0'][1], lv['mask_b'], dd['idx'][0], dd['count'][0:1],
                                       dd['idx'][1], dd['count'][1:2], ray_sums[lvl], [lv['slabs']], ctx['idx'], ctx['count'],
                                       [draw], ctx['packs']['obj'][1])
        elif dd is not None:
            dzs[lvl] = ops.mlp_bwd(om.W_BKGD, rows, N, draw, ctx['packs']['MLP_0'][1], lv['mask_b'], ray_idx=dd['idx'][0],
                                   count=dd['count'][0:1], tail_idx=dd['idx'][1], tail_count=dd['count'][1:2],
                                   draw_ray_sum=ray_sums[lvl])
        else:
            dzs[lvl] = ops.mlp_bwd(om.W_BKGD, rows, N, draw, ctx['packs']['MLP_0'][1], lv['mask_b'])
        if Kb and not obj_bwd_done:           # all K object MLPs: one call (csrc/objects.hip)
            with obj_side:
                ops.obj_bwd_batch(lv['slabs'], ctx['idx'], ctx['count'], draw, ctx['packs']['obj'][1], want_d_enc=pose_opt)
                if pose_opt:                            # d(loss)/d(box pose) through the object encoding, all K at once
                    ops.encode_obj_bwd_batch(K, ctx['idx'], ctx['count'], lv['slabs'].d_enc, lv['t_vals'],
                                             ctx['o_s'], ctx['d_s'], radii, rays.origins, rays.directions, pose_ts, alpha,
                                             pose_sums, enc_flags=obj_enc_flags)
    levels = ctx['levels']

    # the ray classes' counts (multi-hit rays, the boxes they hit): from the de-duplicated forward, or -- with the A/B
    # switch DURF_DEDUP_HIT_RAYS=0, where the forward does not classify -- from a classification of their own, so that
    # the switch stays result-neutral for batches with rays that hit two boxes
    cls_count = dd['count'] if dd is not None else (
        ops.compact_classes(ctx['hit'], N)[1] if (lay.K > 1 and K > 0 and not f32) else None)

    poison_args = None
    if cls_count is not None and lay.K > 1:
        poison_args = (cls_count, lay.box[1] - lay.box[0], lay.K, lay.mlp_size[om.W_BKGD], lay.mlp_size[om.W_OBJ])

    def poison(upto=None):
        # rays that hit two boxes: the reference's gradient is NaN -> 0 for everything they touch (ops.poison_multi_hit);
        # on the local gradient, before whoever all-reduces it
        if poison_args is not None and not defer_poison:
            ops.poison_multi_hit(grad, cls_count, lay.box[1] - lay.box[0], lay.K, lay.mlp_size[om.W_BKGD],
                                 lay.mlp_size[om.W_OBJ], upto=upto)

    flat = variables.flat
    first_only = None                   # bucketed exchange: the objects' slice is FINAL (and in flight) once handed over

    def hand_over_objects(o0, n):
        # the slice leaves for its all-reduce: everything that belongs in it goes in first -- its weight-decay term (each
        # rank adds its own once; clip + Adam divide the sum by the world size) and the multi-hit outcome -- and nothing
        # after this line may write it (the collective reduces it in place)
        nonlocal first_only
        if config.weight_decay_mult != 0:
            ops.weight_decay(flat, grad, config.weight_decay_mult, o0, o0 + n, want_l2=False)
        poison()
        first_only = o0
        objects_ready(grad[o0:o0 + n])

    if obj_f32:                               # weight gradients of the K object MLPs over every level: one launch pair
        o0, sz = lay.mlp_off['BoxMLP_0'], lay.mlp_size[om.W_OBJ]
        ops.objf32_dw_batch([lv['f32']['slabs32'] for lv in levels], ctx['count'], grad[o0:o0 + K * sz], sz)
        if objects_ready is not None:
            hand_over_objects(o0, K * sz)
    if not f32:
        off = lay.mlp_off['MLP_0']
        g_b, p_b = grad[off:off + lay.mlp_size[om.W_BKGD]], variables.mlp_flat('MLP_0')
        enc_l, stash_l = [lv['enc_b'] for lv in levels], [lv['stash_b'] for lv in levels]
        if dd is not None:               # every level: one segment of `nrows` valid rows (1 row per "ray")
            geo = ([rows] * L, [1] * L, [dd['nrows']] * L)
        else:
            geo = ([rows] * L, [N] * L, [None] * L)
        o0, sz = (lay.mlp_off['BoxMLP_0'], lay.mlp_size[om.W_OBJ]) if Kb else (0, 0)
        merged = Kb and not ops.overlap_dw(rows) and objects_ready is None
        if Kb and objects_ready is not None:           # bucketed all-reduce: the objects' gradients first, finalized on their own
            obj_side.join()                            # (the object backward may still be writing dz on the side stream)
            ops.obj_dw_batch([lv['slabs'] for lv in levels], ctx['view_tiles_obj'], ctx['count'],
                             grad[o0:o0 + K * sz], sz, variables.flat[o0:o0 + K * sz])
            hand_over_objects(o0, K * sz)
        if merged:
            # The objects' split-K launch goes FIRST: the finalize launch then finds the background MLP's partials
            # (134 MB, the bulk) still in the 256 MB Infinity Cache -- behind the objects' 0.7 GB operand stream it
            # read them from HBM (k_dw_finalize 117 us instead of 2 x 37, rocprofv3)
            side.join()
            po, bo = ops.obj_dw_partials([lv['slabs'] for lv in levels], ctx['view_tiles_obj'], ctx['count'])
        if view_ready is not None:
            main.wait_event(view_ready)          # (recorded at the start of the backward: long complete)
        ops.mlp_dw_levels(om.W_BKGD, *geo, enc_l, [view_tile] * L, stash_l, [d[0] for d in dzs], [d[1] for d in dzs], *bufs)
        if merged:                       # every MLP of the model is finalized by ONE pair of launches
            ops.dw_finalize_all(*geo, *bufs, g_b, p_b,
                                obj=(K, B, N, ctx['count'], L, po, bo, grad[o0:o0 + K * sz], sz, variables.flat[o0:o0 + K * sz]))
        else:
            ops.dw_finalize_all(*geo, *bufs, g_b, p_b)
            if Kb and objects_ready is None:
                with side:
                    ops.obj_dw_batch([lv['slabs'] for lv in levels], ctx['view_tiles_obj'], ctx['count'],
                                     grad[o0:o0 + K * sz], sz, variables.flat[o0:o0 + K * sz])
    side.join()
    if real is not variables:
        from . import noview
        grad, flat = noview.gather_grad(grad, real), real.flat
    weight_l2 = None
    if config.weight_decay_mult != 0:                                          # :73-75
        # (one launch pair, durf_weight_decay: the gradient term on whatever has not left for its all-reduce yet + the scalar)
        weight_l2 = ops.weight_decay(flat, grad, config.weight_decay_mult, 0, flat.numel() if first_only is None else first_only)
    if K > 0 and pose_opt:                      # no_pose_opt and no_yaw_opt: box_centers get no gradient (:100-104)
        # (k_pose_finish ADDS: straight into this timestep's rows of the zero-filled gradient when they are a view)
        g_ts = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6)
        direct = not torch.is_tensor(ctx['ts'])          # an integer timestep: the rows are a view
        g6 = g_ts[ctx['ts']] if direct else torch.zeros(K, 6, device=dev)
        ops.pose_finish(pose_ts, pose_sums, not model.no_pose_opt, not model.no_yaw_opt, g6)
        if not model.no_pose_opt and config.tv_loss_mult != 0:               # :136,:219
            # (the multiplier enters as the fp32 value the C entry point's `float tv_loss_mult` field carries, widened for
            # the product and rounded once: csrc/train.hip computes the same constant, bit for bit, for any multiplier)
            tv_f32 = struct.unpack('f', struct.pack('f', config.tv_loss_mult))[0]
            g6[:, :3] += (tv_f32 * (1.0 + 0.1 * (L - 1)) * 2.0) * (pose_ts[:, :3] - prev[0, :, :3])
        if not direct:
            g_ts[ctx['ts']] += g6
    pose = ret[0][7][0]
    poison(upto=first_only)
    if dd is not None:
        multi = dd['multi_hit']
    else:
        multi = (dyn > 1).sum() if lay.K > 1 else ops.const_tensor(dev, (), torch.int64)
    raw = dict(norms=norms, sums=sums, terms=terms, weight_l2=weight_l2, ret=ret, ctx=ctx, pose6=pose_ts if lay.K > 0 else None,
               multi_hit=multi, poison=poison_args if defer_poison else None)
    return grad, raw, pose


def _stat_mults(config):
    c = config
    return [c.coarse_loss_mult, c.sky_loss_mult, c.depth_loss_mult, c.near_loss_mult, c.empty_loss_mult, c.tv_loss_mult]


def _assemble_stats(config, batch, raw, prev, mode):
    """Scalars of utils.Stats from the per-level sums (train_boxpose.py:123-249): one fused launch."""
    pose6 = raw['pose6']
    K = 0 if pose6 is None else pose6.shape[0]
    t_levels = [r[4] for r in raw['ret']]
    return ops.train_stats(raw['norms'], raw['sums'], raw['weight_l2'], pose6,
                           prev[0].contiguous() if K else None, batch['target'].contiguous() if K else None,
                           t_levels, _stat_mults(config), mode, terms=raw['terms'])


def train_step(model, config, rng, state, batch, lr, eps, alpha, prev, noise=None, reduce_stats=True):
    """One optimization step (train_boxpose.py:49-321).

    batch: dict(rays=BoxRays, pixels[B,3], depth[B,1], sky[B,1], init[T,K,6], ext[K,3], ts, target[K,6])
    of device tensors for THIS rank's shard.  Returns (new_state, stats, rng, pose).

    Data-parallel exchange: ONE all-reduce of the flat gradient buffer (lax.pmean(grad), :253), issued
    asynchronously so that the stats assembly overlaps it.  `reduce_stats=False` skips the second, ~40-float
    all-reduce of the logged scalars (lax.pmean(stats), :255): the reference only reads them every
    `print_every` steps (:440), so a driver passes `step % print_every == 0`; the scalars returned on the other
    steps are this rank's shard-local values."""
    variables = state.variables
    dist = _dist()
    pending = []
    lay = variables.layout
    comm = _instream_comm(dist) if dist is not None else None
    bucket = dist is not None and comm is None and _bucketed() and lay.K > 0
    ready = (lambda g_obj: pending.append(dist.all_reduce(g_obj, async_op=True))) if bucket else None
    # the step's tail as two launches -- {logged scalars + multi-hit outcome + scrub} and Adam -- whenever the scalars are
    # not all-reduced in between; with a process group the multi-hit NaNs have to exist before the all-reduce (own launch)
    merged_tail = dist is None or not reduce_stats
    grad, raw, pose = loss_and_grad(model, config, rng, variables, batch, eps, alpha, prev, noise=noise, objects_ready=ready,
                                    defer_poison=dist is None)
    world = 1
    if dist is not None:                                    # lax.pmean(grad) (:253)
        world = dist.get_world_size()
        if comm is not None:
            comm.all_reduce_sum(grad)                       # in the compute stream: what follows is ordered behind it
        else:
            rest = grad[:lay.mlp_off['BoxMLP_0']] if bucket else grad    # [box_centers | MLP_0] when the objects went ahead
            pending.append(dist.all_reduce(rest, async_op=True))
    L = model.num_levels
    if merged_tail:
        out = None
    else:                                                   # lax.pmean(stats) (:255), then the PSNRs (:291-292)
        out = _assemble_stats(config, batch, raw, prev, ops.STATS_ASSEMBLE)
        if comm is not None:
            comm.all_reduce_sum(out)
        else:
            dist.all_reduce(out)
        out *= 1.0 / world                                  # (the C step's k_scale multiplies by the same float)
        ops.train_stats(raw['norms'], raw['sums'], None, None, None, None, [r[4] for r in raw['ret']],
                        _stat_mults(config), ops.STATS_PSNR, out=out)
    for work in pending:
        work.wait()                                         # orders the current stream behind the collective(s)
    if merged_tail:
        pose6 = raw['pose6']
        Kp = 0 if pose6 is None else pose6.shape[0]
        out, scratch = ops.stats_scrub(raw['norms'], raw['sums'], raw['weight_l2'], pose6,
                                       prev[0].contiguous() if Kp else None, batch['target'].contiguous() if Kp else None,
                                       [r[4] for r in raw['ret']], _stat_mults(config), ops.STATS_ASSEMBLE | ops.STATS_PSNR,
                                       raw['terms'], grad, 1.0 / world, float(config.grad_max_val), poison=raw['poison'])
        gs = ops.adam_apply(variables.flat, state.m, state.v, grad, float(config.grad_max_norm), float(lr), state.step, scratch)
    else:
        gs = ops.clip_adam(variables.flat, state.m, state.v, grad, 1.0 / world, float(config.grad_max_val),
                           float(config.grad_max_norm), float(lr), state.step)
    st = ops.stats_views(out, L)
    model.prefetch_const_trunk(variables)                   # (fp32 hit-ray path only) the next step's parameter-only work
    new_state = TrainState(variables, state.m, state.v, state.step + 1)
    ret = raw['ret']
    stats = utils.Stats(
        loss=st['loss'], obj_losses=st['obj_losses'], losses=st['losses'], d_losses=st['d_losses'],
        n_losses=st['n_losses'], e_losses=st['e_losses'], s_losses=st['s_losses'],
        distr_losses=st['distr_losses'], tv_losses=st['tv_losses'], sampling_stats=st['sampling_stats'],
        offsets=st['offsets'], offset_x=st['offset_x'], offset_y=st['offset_y'], offset_z=st['offset_z'],
        offset_yaw=st['offset_yaw'], pose=pose, weights=[r[3] for r in ret], samples=[r[4] for r in ret],
        weight_l2=st['weight_l2'], psnr=st['psnrs'][-1], psnrs=st['psnrs'], obj_psnr=st['obj_psnrs'][-1],
        grad_norm=gs[0], grad_abs_max=gs[1], grad_norm_clipped=gs[3], multi_hit_rays=raw['multi_hit'])
    new_rng = (int(rng) + 1) if isinstance(rng, int) else rng
    return new_state, stats, new_rng, pose              # (the prologue's snapshot: does not alias the updated parameters)


def one_call_refusal(model, variables, update=True):
    """why durf_train_step does not cover this step (None: it does) -- the conditions train_step_one_call raises on"""
    lay = variables.layout
    K, L = lay.K, model.num_levels
    dist = _dist()
    if model.mlp_precision != 'bf16' or (K and not model.dynamics) or L < 2 or not lay.use_viewdirs:
        return 'durf_train_step covers the step with a bf16 background MLP (12-Dense tree), dynamic boxes and >= 2 levels'
    if dist is not None and (_instream_comm(dist) is None or not update):
        return ('durf_train_step is data-parallel only through the library\'s own in-stream all-reduce '
                '(nccl backend, self-check passed, DURF_INSTREAM_ALLREDUCE != 0; see csrc/train.hip, csrc/comm.hip)')
    if bool(K) and not (model.no_pose_opt and model.no_yaw_opt) and model.object_precision() != 'f32':
        # (train_step runs this combination -- obj_precision='bf16' forced under pose optimisation -- on the bf16 object
        # kernels; the C entry point only has the fp32 hit-ray branch behind the pose gradient)
        return "durf_train_step optimises box poses with the hit rays in fp32 only (MipNerfModel.obj_precision = 'auto' or 'f32')"
    return None


def best_step_fn(model, variables):
    """the faster host path of a training step for this model: the one C call (durf_train_step -- bit-identical to train_step,
    within +-1.5 % of it at every measured shape, +0.3-0.8 % at the 4096-ray headline: no interpreter between the launches, two
    cross-stream hops fewer; profiles/r05_ab_host_path.txt) wherever it covers the step, else train_step.  bench.py and train_loop use it; the parity tests name the path they mean."""
    return train_step_one_call if one_call_refusal(model, variables) is None else train_step


def train_step_one_call(model, config, rng, state, batch, lr, eps, alpha, prev, noise=None, update=True, reduce_stats=True):
    """`train_step` through ONE library call (durf_train_step, csrc/train.hip): the orchestration of loss_and_grad /
    train_step done in C for hosts that are not Python; same arguments, same (new_state, stats, rng, pose), bit-identical
    results (tests/test_gpu_train_call.py).  Scope of the C entry point: every BASELINE.json training configuration on one
    device -- bf16 background MLP, the object branch in bf16 (frozen poses) or in fp32 with box-pose optimisation behind it
    (cfg4), >= 2 levels; weight decay, density noise and `rand_bkgd` included.  update=False: gradient and scalars
    only (durf_loss_backward) -> (grad, stats buffer views, per-level outputs).  reduce_stats is accepted (and has nothing
    to do on one device) so that this function can be handed to train_loop as its step_fn."""
    model._check()
    variables = state.variables
    lay = variables.layout
    K, L, N = lay.K, model.num_levels, model.num_samples
    dist = _dist()
    comm = _instream_comm(dist) if dist is not None else None
    why = one_call_refusal(model, variables, update)
    if why is not None:
        raise NotImplementedError(why)
    pose_opt = bool(K) and not (model.no_pose_opt and model.no_yaw_opt)
    obj_fp32 = bool(K) and model.object_precision() == 'f32'
    rays = batch['rays']
    B = rays.origins.shape[0]
    dev = variables.flat.device
    seed = None
    if config.randomized and noise is None:
        if isinstance(rng, torch.Generator):
            u = torch.rand(2, B, N + 1, device=dev, generator=rng)
            noise = dict(t_rand=u[0], u_rand=u[1])
            if model.density_noise > 0:             # (level by level, as the model draws them)
                noise['density'] = [torch.randn(B, N, device=dev, generator=rng) for _ in range(L)]
        else:                                       # the library draws (durf_forward_args.draw_noise), as train_step does
            seed = int(rng) if rng is not None else 0
            noise = dict(t_rand=None, u_rand=None)
    dn = model.density_noise if (config.randomized and model.density_noise > 0) else 0.0
    if dn and seed is None and 'density' not in noise:      # injected sampling draws only: the generator the model falls back to
        gd = om._make_generator(rng, dev)
        noise = dict(noise, density=[torch.randn(B, N, device=dev, generator=gd) for _ in range(L)])
    ts = int(batch['ts'])
    pose = variables['params']['box_centers'][ts]          # a view of the parameters: the pose gradient goes to the same rows
    assert pose.is_contiguous()
    flags = ((ops.ENC_CONTRACT if model.contraction else 0) | (ops.ENC_NO_INTEGRATION if model.disable_integration else 0) |
             (ops.ENC_CYLINDER if model.ray_shape == 'cylinder' else 0))
    # The constant trunk of the fp32 hit-ray branch (a function of the parameters alone) across steps: the call refills the
    # buffer behind its update, on its side stream, and the next call uses it if nothing wrote the parameters in between --
    # through torch (version counter) or through this library (ops.param_generation), exactly MipNerfModel.prefetch_const_trunk's
    # guard for the Python-issued launches
    trunk_buf = trunk_ok = None
    if obj_fp32:
        c = getattr(variables, '_c_trunk', None)
        if c is None or c['buf'].device != dev:
            c = variables._c_trunk = dict(buf=torch.empty(264, device=dev), key=None)
        trunk_buf, trunk_ok = c['buf'], c['key'] == (variables.flat._version, ops.param_generation(variables.flat))
    # (pose_used: what the step renders with -- the update is in place; snapshot by the call's first launch, like cls, the
    # ray classes' counts: no launch of their own)
    outs, dyn, zo, grad, out, gs, pose_used, cls = ops.train_call(
        rays, pose, batch['ext'].reshape(-1, 3).contiguous() if K else None, variables.flat, state.m, state.v,
        lay.box[1] - lay.box[0], lay.mlp_size[om.W_BKGD], lay.mlp_size[om.W_OBJ], N, L, alpha, flags,
        rays.lossmult, batch['pixels'][..., :3], batch['depth'], batch['sky'], batch['target'] if K else None,
        prev[0] if K else None, eps, config.box_loss_mult, 0.0 if config.rand_bkgd else (1.0 if config.white_bkgd else 0.5),
        config.disable_multiscale_loss,
        [level_multipliers(config, lvl, L) for lvl in range(L)], _stat_mults(config), lr, config.grad_max_val,
        config.grad_max_norm, state.step, lindisp=model.lindisp,
        bkgd_mode=ops.BKGD_RAND if config.rand_bkgd else (ops.BKGD_WHITE if config.white_bkgd else ops.BKGD_GREY),
        density_bias=model.density_bias, resample_padding=model.resample_padding,
        t_rand=noise['t_rand'] if config.randomized else None, u_rand=noise['u_rand'] if config.randomized else None,
        update=update, obj_fp32=obj_fp32, obj_x3=model.object_x3(), want_pos=pose_opt and not model.no_pose_opt, want_rot=pose_opt and not model.no_yaw_opt,
        tv_loss_mult=config.tv_loss_mult if pose_opt else 0.0, seed=seed, comm=comm,
        world=dist.get_world_size() if dist is not None else 1, reduce_stats=dist is not None and reduce_stats,
        density_noise=dn, density_rand=noise.get('density') if dn else None, weight_decay_mult=config.weight_decay_mult,
        const_trunk=trunk_buf, const_trunk_valid=trunk_ok)
    if obj_fp32:        # (update=True: the buffer now holds the trunk of the updated parameters; update=False: of the unchanged ones)
        variables._c_trunk['key'] = (variables.flat._version, ops.param_generation(variables.flat))
    if pose_used is None:
        pose_used = pose                                       # K = 0: [0, 6]
    box_rot0 = pose_used[0, 3:] if K > 0 else ops.const_tensor(dev, (3,))
    ret = [tuple(o) + ([pose_used[:, :3], box_rot0], dyn, zo) for o in outs]
    st = ops.stats_views(out, L)
    if not update:
        return grad, st, ret
    pose_out = ret[0][7][0]
    stats = utils.Stats(
        loss=st['loss'], obj_losses=st['obj_losses'], losses=st['losses'], d_losses=st['d_losses'],
        n_losses=st['n_losses'], e_losses=st['e_losses'], s_losses=st['s_losses'],
        distr_losses=st['distr_losses'], tv_losses=st['tv_losses'], sampling_stats=st['sampling_stats'],
        offsets=st['offsets'], offset_x=st['offset_x'], offset_y=st['offset_y'], offset_z=st['offset_z'],
        offset_yaw=st['offset_yaw'], pose=pose_out, weights=[r[3] for r in ret], samples=[r[4] for r in ret],
        weight_l2=st['weight_l2'], psnr=st['psnrs'][-1], psnrs=st['psnrs'], obj_psnr=st['obj_psnrs'][-1],
        grad_norm=gs[0], grad_abs_max=gs[1], grad_norm_clipped=gs[3],
        multi_hit_rays=cls[3] if K > 1 else ops.const_tensor(dev, (), torch.int64))
    new_rng = (int(rng) + 1) if isinstance(rng, int) else rng
    return TrainState(variables, state.m, state.v, state.step + 1), stats, new_rng, pose_out


# ---------------------------------------------------------------------------
# data-parallel plumbing (one process per GPU; jax.pmap's role, train_boxpose.py:370-374)
# ---------------------------------------------------------------------------
def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RCCL on GPU, gloo on CPU).
    Returns (rank, world, local_rank)."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:               # before anything touches the GPU: this rank's host thread on the cores near its device
        pin_host_thread(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)     # several ranks may share a GPU in tests (gloo)
    if (world > 1 or _force_dist()) and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('DURF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        rdzv = os.environ.get('DURF_RDZV_FILE')             # bench.py's own spawner: a file store, no port to race for
        if rdzv:
            dist.init_process_group(backend=backend, init_method='file://' + rdzv, rank=rank, world_size=world)
        elif world == 1 and 'MASTER_ADDR' not in os.environ:      # DURF_FORCE_DIST on a single process
            import tempfile
            path = os.path.join(tempfile.gettempdir(), 'durf_rdzv_%d' % os.getpid())
            if os.path.exists(path):
                os.remove(path)
            dist.init_process_group(backend=backend, init_method='file://' + path, rank=0, world_size=1)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(batch, rank, world):
    """utils.shard's intent (internal/utils.py:193-196): split the per-ray leaves along axis 0;
    init / ext / ts / target are replicated (all rays of a step share one timestep)."""
    if world == 1:
        return batch
    B = batch['pixels'].shape[0]
    if B % world:
        raise ValueError('Batch size must be divisible by the number of devices.')   # :332-333
    n = B // world
    sl = slice(rank * n, (rank + 1) * n)
    out = dict(batch)
    out['rays'] = utils.namedtuple_map(lambda r: r[sl].contiguous(), batch['rays'])
    for k in ('pixels', 'depth', 'sky'):
        out[k] = batch[k][sl].contiguous()
    return out


# ---------------------------------------------------------------------------
# thin training driver (train_boxpose.py:324-437,529-580): schedules -> train_step -> pose feedback ->
# logging / checkpoints / test-set render.  Everything device-side stays on the device between steps.
# ---------------------------------------------------------------------------
def make_schedules(config):
    """learning_rate_fn, eps_rate_fn, alpha_rate_fn (train_boxpose.py:347-368)."""
    import functools
    lr_fn = functools.partial(dmath.learning_rate_decay, lr_init=config.lr_init, lr_final=config.lr_final,
                              max_steps=config.max_steps, lr_delay_steps=config.lr_delay_steps,
                              lr_delay_mult=config.lr_delay_mult)
    eps_fn = functools.partial(dmath.learning_rate_decay, lr_init=config.eps_init, lr_final=config.eps_final,
                               max_steps=config.eps_max_steps, lr_delay_steps=config.eps_delay_steps,
                               lr_delay_mult=config.lr_delay_mult)
    alpha_fn = functools.partial(dmath.freq_alpha_rate, alpha_init=config.alpha_init, alpha_final=config.alpha_final,
                                 alpha_delay_steps=config.alpha_delay_steps, alpha_max_steps=config.alpha_max_steps)
    return lr_fn, eps_fn, alpha_fn


def make_render_fn(model, config, variables, one_call=None):
    """render_eval_fn (train_boxpose.py:377-390): test-mode model.apply; the all-gather lives in render_image.
    one_call: each chunk through the single C entry point durf_forward (MipNerfModel.apply_one_call; bit-identical, a few
    launches' worth of host work less per chunk); None = wherever that entry point covers the model."""
    if one_call is None:
        one_call = model.supports_one_call(variables)
    apply = model.apply_one_call if one_call else model.apply

    def render_fn(rng, batch):
        return apply(variables, rng, batch['rays'], batch['init'], batch['ext'], batch['ts'],
                     randomized=False, rand_bkgd=False, white_bkgd=config.white_bkgd, alpha=batch['alpha'])
    return render_fn


def evaluate(model, config, variables, test_case, alpha, chunk=8192, rng=0):
    """One test image (train_boxpose.py:535-563): render, PSNR, SSIM.  -> dict(psnr, ssim, rgb, distance, acc, rays)"""
    from . import metrics
    if _dist() is None and model.supports_one_call(variables):
        # one device: the whole image as ONE C call (durf_render_image: the chunk loop over the resident ray buffer)
        rgb, dist_, acc = model.render_image_one_call(variables, test_case['rays'], test_case['init'], test_case['ext'],
                                                      test_case['ts'], config.white_bkgd, alpha, chunk=chunk)
    else:
        rgb, dist_, acc = om.render_image(make_render_fn(model, config, variables), test_case['rays'], test_case['init'],
                                          test_case['ext'], test_case['ts'], rng, alpha, chunk=chunk)
    gt = test_case['pixels'][..., :3]
    psnr = dmath.mse_to_psnr(((rgb - gt) ** 2).mean())                                   # :562
    ssim = metrics.compute_ssim(rgb, gt, 1.0) if rgb.is_cuda else None                   # :563
    return dict(psnr=psnr, ssim=ssim, rgb=rgb, distance=dist_, acc=acc, rays=rgb.shape[0] * rgb.shape[1])


def train_loop(model, config, state, dataset, test_dataset=None, train_dir=None, render_every=0, chunk=8192,
               rng=20200823, step_fn=None, log=print, world=1, rank=0, keep=100):
    """The body of the reference's main() (train_boxpose.py:416-580) around `train_step`.

    dataset: iterator of this rank's batches (dicts as train_step takes them, 'ts' a host int) with .peek();
    test_dataset: iterator of test cases (full-image rays [H,W,.], pixels [H,W,3], init, ext, ts) or None.
    Box poses estimated at one timestep seed the TV prior of its neighbours (`prevs`, :414,426-437): the table
    stays on the device and is updated in place from the step's returned pose -- no host round trip.
    Returns (state, history) with history = list of (step, dict of logged scalars)."""
    import time
    from . import checkpoints
    step_fn = step_fn or best_step_fn(model, state.variables)       # (the one C call wherever it covers the step)
    if isinstance(rng, int) and world > 1:
        # every rank draws its OWN stratified-sampling / resampling / density noise for its own rays, as the reference's keys
        # differ per device (train_boxpose.py:415 `keys = random.split(rng, jax.local_device_count())`, one per pmap replica):
        # the rank is the high word of the in-kernel Philox key (ops.split_seed), the step counter stays in the low word
        rng = (int(rng) & 0xFFFFFFFF) | (int(rank) << 32)
    lr_fn, eps_fn, alpha_fn = make_schedules(config)
    if train_dir is not None:
        state = checkpoints.restore_checkpoint(train_dir, state)                         # :404
    init_step = state.step + 1                                                           # :406
    prevs = dataset.peek()['init'].clone()                                               # :414  [T,K,6] on the device
    T = prevs.shape[0]
    history, trace = [], []
    t_loop, reset_timer = time.time(), True
    alpha = alpha_fn(init_step)
    for step, batch in zip(range(init_step, config.max_steps + 1), dataset):            # :420
        if reset_timer:
            t_loop, reset_timer = time.time(), False
        lr, eps, alpha = lr_fn(step), eps_fn(step), alpha_fn(step)                       # :425-427
        ts = int(batch['ts'])
        nb = (ts + 1 if ts == 0 else ts - 1) % max(T, 1)                                 # :429-432
        prev = prevs[nb:nb + 1]
        logging = step % config.print_every == 0
        state, stats, rng, pose = step_fn(model, config, rng, state, batch, lr, eps, alpha, prev, reduce_stats=logging)
        if prevs.shape[1] > 0:
            prevs[ts, :, :3] = pose                                                      # :437 (device to device)
        trace.append((stats.loss, stats.psnr, stats.grad_norm))
        if logging:                                                                      # :448-527 (rank 0 writes)
            losses = torch.stack([torch.as_tensor(t[0]).float().reshape(()) for t in trace])
            psnrs = torch.stack([torch.as_tensor(t[1]).float().reshape(()) for t in trace])
            gn = torch.stack([torch.as_tensor(t[2]).float().reshape(()) for t in trace])
            d_ = _dist()
            if d_ is not None and d_.get_world_size() > 1:
                # the reference's stats_trace holds pmean'd stats of EVERY step (train_boxpose.py:255,440); here only
                # the logging step's were all-reduced, so average the window's loss / psnr over the ranks now -- one
                # small collective per print_every steps (the gradient norm is global already: it is taken after pmean)
                # (the logging step's own entry is already a mean over the ranks -- averaging equal values again is exact)
                lp = torch.stack([losses, psnrs])
                d_.all_reduce(lp)
                losses, psnrs = lp[0] / d_.get_world_size(), lp[1] / d_.get_world_size()
            steps_per_sec = len(trace) / max(time.time() - t_loop, 1e-9)
            rec = dict(loss=float(stats.loss), avg_loss=float(losses.mean()), avg_psnr=float(psnrs.mean()),
                       max_grad_norm=float(gn.max()), lr=lr, eps=eps, alpha=alpha,
                       rays_per_sec=config.batch_size * steps_per_sec)
            pose_opt = not (model.no_pose_opt and model.no_yaw_opt)
            if pose_opt and int(getattr(stats, 'multi_hit_rays', 0)) > 0 and rank == 0:
                log('warning: %d rays of this batch hit two or more boxes; their cross-object pose gradient is not '
                    'propagated (see utils.Stats.multi_hit_rays)' % int(stats.multi_hit_rays))
            history.append((step, rec))
            trace, reset_timer = [], True
            if rank == 0:
                log('%*d/%d: i_loss=%0.4f, avg_loss=%0.4f, avg_psnr=%0.2f, lr=%0.2e, %0.0f rays/sec' % (
                    len(str(config.max_steps)) + 1, step, config.max_steps, rec['loss'], rec['avg_loss'],
                    rec['avg_psnr'], lr, rec['rays_per_sec']))
        if train_dir is not None and rank == 0 and step % config.save_every == 0:        # :528-532
            checkpoints.save_checkpoint(train_dir, state, int(step), keep=keep)
        if test_dataset is not None and render_every > 0 and step % render_every == 0:   # :535-575 (all ranks render)
            t0 = time.time()
            ev = evaluate(model, config, state.variables, next(test_dataset), alpha, chunk=chunk, rng=rng)
            dt = time.time() - t0
            history.append((step, dict(test_psnr=float(ev['psnr']), test_ssim=None if ev['ssim'] is None else float(ev['ssim']),
                                       test_rays_per_sec=ev['rays'] / dt)))
            if rank == 0:
                log('Eval %d: %0.3fs., %0.0f rays/sec, psnr %0.3f' % (step, dt, ev['rays'] / dt, float(ev['psnr'])))
    if train_dir is not None and rank == 0 and config.max_steps % config.save_every != 0 and state.step > 0:   # :577-580
        checkpoints.save_checkpoint(train_dir, state, int(config.max_steps), keep=keep)
    return state, history


class SyntheticTimestepDataset:
    """Stand-in for obbpose_dataset.Waymo with batching == 'timestep' (obbpose_dataset.py:1551-1587): T timesteps of
    procedural images resident in HBM (raygen.TimestepData), each step samples one timestep and `batch_size` pixel
    indices of it and generates the rays on the device.  No dataset exists in this environment; this drives main()."""

    def __init__(self, config, K=3, T=5, hw=(64, 96), n_cams=2, seed=0, device='cuda', rank=0, world=1, split='train'):
        import numpy as np
        from . import raygen
        self.config, self.device, self.rank, self.world, self.split = config, torch.device(device), rank, world, split
        self.rs = np.random.default_rng(seed + (0 if split == 'train' else 1))
        H, W = hw
        # K boxes fanned out in azimuth at distance 6 in front of the rig, so that no ray can hit two of them (the
        # reference's model is undefined there: obbpose_model.py:120-122), even after the random_box noise
        dist_, half = 6.0, np.array([0.2, 0.17, 0.45])
        az = np.deg2rad(np.linspace(-35.0, 35.0, K)) if K > 1 else np.zeros(K)
        centers = np.stack([dist_ * np.sin(az), np.zeros(K), -dist_ * np.cos(az)], -1)
        rots = np.zeros((K, 3))
        rots[:, 1] = self.rs.uniform(-0.5, 0.5, K)
        base = np.concatenate([centers, rots], -1)
        init = np.tile(base[None], (T, 1, 1))
        init[:, :, :3] += np.random.default_rng(seed).normal(0, 0.03, (T, K, 3))       # the boxes drift between timesteps
        self.target_all = torch.tensor(init, dtype=torch.float32, device=self.device)
        self.init = self.target_all.clone()
        self.ext = torch.tensor(np.tile(half, (K, 1)), dtype=torch.float32, device=self.device)
        if config.random_box and split == 'train' and K > 0:                          # configs/waymo.gin:6,8
            gap = np.deg2rad(70.0 / (K - 1)) if K > 1 else np.pi
            cap = max(0.0, dist_ * (gap / 2 - 0.12) / 2)                              # keeps the noisy boxes apart in azimuth
            amp = min(config.box_noise, cap)
            noise = self.rs.uniform(-amp, amp, (T, K, 3))
            self.init[..., :3] += torch.tensor(noise, dtype=torch.float32, device=self.device)
        self.T, self.H, self.W, self.n_cams = T, H, W, n_cams
        self.ts_data = []
        uu, vv = np.meshgrid(np.linspace(0, 1, W), np.linspace(0, 1, H))
        for t in range(T):
            c2w, imgs, deps, skys = [], [], [], []
            for c in range(n_cams):
                yaw = 0.2 * (c - 0.5 * (n_cams - 1)) + 0.02 * t
                R_ = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
                c2w.append(np.concatenate([R_, np.array([[0.05 * t], [0.0], [0.0]])], 1))
                img = np.stack([0.5 + 0.4 * np.sin(6 * uu + c + 0.3 * t), 0.5 + 0.4 * np.cos(5 * vv + 0.2 * t),
                                0.5 + 0.3 * np.sin(4 * (uu + vv))], -1)
                imgs.append(img.astype(np.float32))
                deps.append(np.where(self.rs.uniform(0, 1, (H, W)) < 0.3, 2.0 + 4.0 * vv, 0.0).astype(np.float32))
                skys.append(np.where(vv < 0.1, 0.975, 0.0).astype(np.float32))
            foc = [0.8 * W] * n_cams
            pp = [(W / 2.0, H / 2.0)] * n_cams
            self.ts_data.append(raygen.TimestepData(c2w, foc, pp, [H] * n_cams, [W] * n_cams, imgs, deps, skys,
                                                    device=self.device))
        self._peek = None

    def _train_batch(self):
        from . import raygen
        ts = int(self.rs.integers(0, self.T))
        td = self.ts_data[ts]
        per = self.config.batch_size // self.world
        idx_all = self.rs.integers(0, td.n_rays, self.config.batch_size)                  # same draw on every rank
        if getattr(self, '_upload', None) is None:
            self._upload = raygen.IndexUploader(self.device)                              # (no host stall per step)
        idx = self._upload(idx_all[self.rank * per:(self.rank + 1) * per])
        rays, px, dp, sk = raygen.generate_batch(td, idx, self.config.near, self.config.far)
        return dict(rays=rays, pixels=px, depth=dp, sky=sk, init=self.init, ext=self.ext, ts=ts, target=self.target_all[ts])

    def _test_case(self):
        from . import raygen
        ts = int(self.rs.integers(0, self.T))
        td = self.ts_data[ts]
        rays, px, dp, sk = raygen.generate_batch(td, None, self.config.near, self.config.far)
        n = self.H * self.W                                                               # first camera of the timestep
        img = lambda x: x[:n].reshape(self.H, self.W, -1)
        return dict(rays=utils.namedtuple_map(img, rays), pixels=img(px), depth=img(dp), sky=img(sk), init=self.init,
                    ext=self.ext, ts=ts)

    def peek(self):
        if self._peek is None:
            self._peek = self._train_batch() if self.split == 'train' else self._test_case()
        return self._peek

    def __iter__(self):
        return self

    def __next__(self):
        if self._peek is not None:
            b, self._peek = self._peek, None
            return b
        return self._train_batch() if self.split == 'train' else self._test_case()


def main(argv=None):
    """python -m durf_amd.train_boxpose --gin_file configs/waymo.gin --train_dir /tmp/run [--gin_param ...]
    One process per GPU (launch N with torchrun / the bench's own spawner): the reference's main() :324-580."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--gin_file', action='append', default=[])
    ap.add_argument('--gin_param', action='append', default=[])
    ap.add_argument('--train_dir', default=None)
    ap.add_argument('--data_dir', default=None, help='dataset directory (loaders: durf_amd.datasets); synthetic when omitted')
    ap.add_argument('--render_every', type=int, default=0)
    ap.add_argument('--chunk', type=int, default=8192)
    ap.add_argument('--objects', type=int, default=3, help='dynamic boxes of the synthetic scene')
    args = ap.parse_args(argv)
    rank, world, local = init_distributed()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    utils.clear_gin()
    config = utils.load_config(args.gin_file, args.gin_param)
    if config.batch_size % world != 0:
        raise ValueError('Batch size must be divisible by the number of devices.')      # :332-333
    if args.data_dir:
        from . import datasets
        dataset = datasets.get_dataset('train', args.data_dir, config, device=dev, rank=rank, world=world)
        test_dataset = datasets.get_dataset('test', args.data_dir, config, device=dev, rank=rank, world=world)
    else:
        dataset = SyntheticTimestepDataset(config, K=args.objects, device=dev, rank=rank, world=world, split='train')
        test_dataset = SyntheticTimestepDataset(config, K=args.objects, device=dev, rank=rank, world=world, split='test')
    model, variables = om.construct_mipnerf(20200823, dataset.peek(), device=dev)       # :325,339
    if rank == 0:
        print('Number of parameters being optimized: %d' % variables.flat.numel())       # :340-342
    state = create_train_state(variables)
    state, history = train_loop(model, config, state, dataset, test_dataset, args.train_dir, args.render_every,
                                args.chunk, world=world, rank=rank)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        shutdown_instream()                 # the library's communicator goes before the process group it was built through
        dist.destroy_process_group()
    return history


if __name__ == '__main__':
    main()
