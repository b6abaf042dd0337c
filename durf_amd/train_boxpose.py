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


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
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


def loss_and_grad(model, config, rng, variables, batch, eps, alpha, prev, noise=None):
    """value_and_grad(loss_fn) (train_boxpose.py:67-252) for this rank's shard.
    Returns (grad_flat, raw stats dict of device tensors, pose)."""
    pose_opt = not (model.no_pose_opt and model.no_yaw_opt)
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
    ret, ctx = model._forward(variables, rng, rays, batch['init'], batch['ext'], batch['ts'],
                              config.randomized, config.rand_bkgd, config.white_bkgd, alpha, train=True,
                              noise=noise, loss_prep=prep)
    B, N, K = ctx['B'], ctx['N'], ctx['K']
    lay = variables.layout
    rows = B * N
    pixels = batch['pixels'][..., :3].contiguous()
    dyn = ret[0][8].reshape(-1).to(torch.int32).contiguous()
    bg = 0.0 if config.rand_bkgd else (1.0 if config.white_bkgd else 0.5)
    grad = torch.zeros_like(variables.flat)
    bufs = ops.dw_buffers(om.W_BKGD, dev)
    dzs = [None] * L                            # per-level (dz, dz_out) of the bkgd MLP, consumed by ONE dW launch
    view_tile = ops.expand_view(rows, N, ctx['view'])
    sums = torch.empty(L, ops.TERM_ROWS, device=dev)
    radii = rays.radii.reshape(-1).contiguous()
    pose_ts = variables['params']['box_centers'][ctx['ts']].contiguous()
    pose_sums = torch.zeros(max(K, 1), 21, device=dev) if pose_opt else None
    # last level first: its loss kernel also fills that level's rendered outputs (ret[-1]) when the forward deferred them
    for lvl in reversed(range(L)):
        lv = ctx['levels'][lvl]
        if prep is None:
            ops.loss_prep(lv['t_vals'], lossmult, gt_depth, sky, dyn, ctx['zo'], float(eps),
                          float(config.box_loss_mult), lvl, config.disable_multiscale_loss, norm=norms[lvl])
        norm = norms[lvl]
        out = (lv['rgb'], lv['depth'], lv['acc'], lv['weights'], lv['t_mids'], lv['t_dists']) if lv['deferred'] else None
        draw, _ = ops.loss_bwd(lv['raw_b'], lv['raws'], ctx['slot'], lv['t_vals'], ctx['d_s'], pixels, lossmult,
                               gt_depth, sky, dyn, ctx['zo'], norm, float(eps),
                               level_multipliers(config, lvl, L), float(config.box_loss_mult), lvl, bg,
                               model.density_bias, config.disable_multiscale_loss, sums=sums[lvl], render_out=out)
        dzs[lvl] = ops.mlp_bwd(om.W_BKGD, rows, N, draw, ctx['packs']['MLP_0'][1], lv['mask_b'])
        if K:                                 # all K object MLPs: one call (csrc/objects.hip)
            ops.obj_bwd_batch(lv['slabs'], ctx['idx'], ctx['count'], draw, ctx['packs']['obj'][1], want_d_enc=pose_opt)
            for k in range(K if pose_opt else 0):   # d(loss)/d(box pose) through the object encoding
                ops.encode_obj_bwd(k, ctx['idx'][k], ctx['count'][k:k + 1], lv['slabs'].d_enc[k], lv['t_vals'],
                                   ctx['o_s'], ctx['d_s'], radii, rays.origins, rays.directions, pose_ts, alpha,
                                   pose_sums)
    levels = ctx['levels']
    ops.mlp_dw(om.W_BKGD, rows, N, [lv['enc_b'] for lv in levels], [view_tile] * L, [lv['stash_b'] for lv in levels],
               [d[0] for d in dzs], [d[1] for d in dzs], *bufs)
    off = lay.mlp_off['MLP_0']
    ops.mlp_dw_finalize(om.W_BKGD, om.IN_BKGD, rows, N, L, *bufs, grad[off:off + lay.mlp_size[om.W_BKGD]])
    if K:
        o0, sz = lay.mlp_off['BoxMLP_0'], lay.mlp_size[om.W_OBJ]
        ops.obj_dw_batch([lv['slabs'] for lv in levels], ctx['view_tiles_obj'], ctx['count'], grad[o0:o0 + K * sz], sz)
    flat = variables.flat
    weight_l2 = None
    if config.weight_decay_mult != 0:                                          # :73-75
        weight_l2 = config.weight_decay_mult * (flat * flat).sum() / flat.numel()
        grad += (2.0 * config.weight_decay_mult / flat.numel()) * flat
    if K > 0 and pose_opt:                      # no_pose_opt and no_yaw_opt: box_centers get no gradient (:100-104)
        g6 = torch.zeros(K, 6, device=dev)
        if pose_opt:
            ops.pose_finish(pose_ts, pose_sums, not model.no_pose_opt, not model.no_yaw_opt, g6)
        if not model.no_pose_opt and config.tv_loss_mult != 0:               # :136,:219
            g6[:, :3] += (config.tv_loss_mult * (1.0 + 0.1 * (L - 1)) * 2.0) * (pose_ts[:, :3] - prev[0, :, :3])
        grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6)[ctx['ts']] += g6
    pose = ret[0][7][0]
    raw = dict(norms=norms, sums=sums, weight_l2=weight_l2, ret=ret, ctx=ctx, pose6=pose_ts if lay.K > 0 else None)
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
                           t_levels, _stat_mults(config), mode)


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
    grad, raw, pose = loss_and_grad(model, config, rng, variables, batch, eps, alpha, prev, noise=noise)
    dist = _dist()
    world = 1
    pending = None
    if dist is not None:                                    # lax.pmean(grad) (:253)
        world = dist.get_world_size()
        pending = dist.all_reduce(grad, async_op=True)
    L = model.num_levels
    if dist is None or not reduce_stats:
        out = _assemble_stats(config, batch, raw, prev, ops.STATS_ASSEMBLE | ops.STATS_PSNR)
    else:                                                   # lax.pmean(stats) (:255), then the PSNRs (:291-292)
        out = _assemble_stats(config, batch, raw, prev, ops.STATS_ASSEMBLE)
        dist.all_reduce(out)
        out /= world
        ops.train_stats(raw['norms'], raw['sums'], None, None, None, None, [r[4] for r in raw['ret']],
                        _stat_mults(config), ops.STATS_PSNR, out=out)
    if pending is not None:
        pending.wait()                                      # orders the current stream behind the collective
    st = ops.stats_views(out, L)
    gs = ops.clip_adam(variables.flat, state.m, state.v, grad, 1.0 / world, float(config.grad_max_val),
                       float(config.grad_max_norm), float(lr), state.step)
    new_state = TrainState(variables, state.m, state.v, state.step + 1)
    ret = raw['ret']
    stats = utils.Stats(
        loss=st['loss'], obj_losses=st['obj_losses'], losses=st['losses'], d_losses=st['d_losses'],
        n_losses=st['n_losses'], e_losses=st['e_losses'], s_losses=st['s_losses'],
        distr_losses=st['distr_losses'], tv_losses=st['tv_losses'], sampling_stats=st['sampling_stats'],
        offsets=st['offsets'], offset_x=st['offset_x'], offset_y=st['offset_y'], offset_z=st['offset_z'],
        offset_yaw=st['offset_yaw'], pose=pose, weights=[r[3] for r in ret], samples=[r[4] for r in ret],
        weight_l2=st['weight_l2'], psnr=st['psnrs'][-1], psnrs=st['psnrs'], obj_psnr=st['obj_psnrs'][-1],
        grad_norm=gs[0], grad_abs_max=gs[1], grad_norm_clipped=gs[3])
    new_rng = (int(rng) + 1) if isinstance(rng, int) else rng
    return new_state, stats, new_rng, pose.clone()


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
    if torch.cuda.is_available():
        local = local % max(torch.cuda.device_count(), 1)     # several ranks may share a GPU in tests (gloo)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('DURF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
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
