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
    ret, ctx = model._forward(variables, rng, rays, batch['init'], batch['ext'], batch['ts'],
                              config.randomized, config.rand_bkgd, config.white_bkgd, alpha, train=True,
                              noise=noise)
    B, N, K = ctx['B'], ctx['N'], ctx['K']
    L = model.num_levels
    lay = variables.layout
    dev = variables.flat.device
    rows = B * N
    lossmult = rays.lossmult.reshape(-1).contiguous()
    gt_depth = batch['depth'].reshape(-1).contiguous()
    sky = batch['sky'].reshape(-1).contiguous()
    pixels = batch['pixels'][..., :3].contiguous()
    dyn = ret[0][8].reshape(-1).to(torch.int32).contiguous()
    bg = 0.0 if config.rand_bkgd else (1.0 if config.white_bkgd else 0.5)
    grad = torch.zeros_like(variables.flat)
    names = lay.mlp_names()
    bufs = {n: ops.dw_buffers(lay.mlp_dims(n)[0], L, dev) for n in names}
    view_tile = ops.expand_view(rows, N, ctx['view'])
    view_tiles_obj = [ops.expand_view(rows, N, ctx['view'], ray_idx=ctx['idx'][k], count=ctx['count'][k:k + 1])
                      for k in range(K)]
    norms, sums = [], []
    radii = rays.radii.reshape(-1).contiguous()
    pose_ts = variables['params']['box_centers'][ctx['ts']].contiguous()
    pose_sums = torch.zeros(max(K, 1), 21, device=dev) if pose_opt else None
    for lvl in range(L):
        lv = ctx['levels'][lvl]
        norm = ops.loss_prep(lv['t_vals'], lossmult, gt_depth, sky, dyn, ctx['zo'], float(eps),
                             float(config.box_loss_mult), lvl, config.disable_multiscale_loss)
        draw, s = ops.loss_bwd(lv['raw_b'], lv['raws'], ctx['slot'], lv['t_vals'], ctx['d_s'], pixels, lossmult,
                               gt_depth, sky, dyn, ctx['zo'], norm, float(eps),
                               level_multipliers(config, lvl, L), float(config.box_loss_mult), lvl, bg,
                               model.density_bias, config.disable_multiscale_loss)
        norms.append(norm)
        sums.append(s)
        dz, dz_out = ops.mlp_bwd(om.W_BKGD, rows, N, draw, ctx['packs']['MLP_0'][1], lv['mask_b'])
        ops.mlp_dw(om.W_BKGD, rows, N, lv['enc_b'], view_tile, lv['stash_b'], dz, dz_out, lvl, L, *bufs['MLP_0'])
        for k in range(K):
            nm = 'BoxMLP_%d' % k
            cnt = ctx['count'][k:k + 1]
            res = ops.mlp_bwd(om.W_OBJ, rows, N, draw, ctx['packs'][nm][1], lv['masks'][k],
                              ray_idx=ctx['idx'][k], count=cnt, want_d_enc=pose_opt)
            dzk, dzk_out = res[0], res[1]
            if pose_opt:                      # d(loss)/d(box pose) through the object encoding
                ops.encode_obj_bwd(k, ctx['idx'][k], cnt, res[2], lv['t_vals'], ctx['o_s'], ctx['d_s'], radii,
                                   rays.origins, rays.directions, pose_ts, alpha, pose_sums)
            ops.mlp_dw(om.W_OBJ, rows, N, lv['encs'][k], view_tiles_obj[k], lv['stashes'][k], dzk, dzk_out,
                       lvl, L, *bufs[nm], count=cnt)
    for n in names:
        width, in_dim = lay.mlp_dims(n)
        off = lay.mlp_off[n]
        ops.mlp_dw_finalize(width, in_dim, L, *bufs[n], grad[off:off + lay.mlp_size[width]])
    flat = variables.flat
    weight_l2 = torch.zeros((), device=dev)
    if config.weight_decay_mult != 0:                                          # :73-75
        weight_l2 = config.weight_decay_mult * (flat * flat).sum() / flat.numel()
        grad += (2.0 * config.weight_decay_mult / flat.numel()) * flat
    if K > 0:
        g6 = torch.zeros(K, 6, device=dev)
        if pose_opt:
            ops.pose_finish(pose_ts, pose_sums, not model.no_pose_opt, not model.no_yaw_opt, g6)
        if not model.no_pose_opt and config.tv_loss_mult != 0:               # :136,:219
            g6[:, :3] += (config.tv_loss_mult * (1.0 + 0.1 * (L - 1)) * 2.0) * (pose_ts[:, :3] - prev[0, :, :3])
        grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6)[ctx['ts']] += g6
    pose = ret[0][7][0]
    raw = dict(norms=torch.stack(norms), sums=torch.stack(sums), weight_l2=weight_l2, ret=ret, ctx=ctx)
    return grad, raw, pose


def _assemble_stats(model, config, batch, raw, prev, pose, yaw0):
    """Scalars of utils.Stats from the per-level sums (train_boxpose.py:123-249)."""
    norms, sums = raw['norms'], raw['sums']            # [L,5], [L,7]
    one = torch.ones((), device=norms.device)
    D = torch.maximum(norms[:, 1], one)
    S = torch.maximum(norms[:, 2], one)
    losses = sums[:, 0] / norms[:, 0]
    obj_losses = sums[:, 1] / norms[:, 4]
    d_losses, n_losses, e_losses = sums[:, 2] / D, sums[:, 3] / D, sums[:, 4] / D
    s_losses = sums[:, 5] / S
    distr = sums[:, 6]
    L = losses.shape[0]
    target = batch['target']
    tv = ((pose - prev[:, :, :3]) ** 2).sum().expand(L)
    c = config
    loss = c.coarse_loss_mult * losses[:-1].sum() + losses[-1] + raw['weight_l2']
    loss = loss + c.sky_loss_mult * s_losses[:-1].sum() + 10.0 * c.sky_loss_mult * s_losses[-1]
    loss = loss + c.depth_loss_mult * d_losses[-1] + 0.1 * c.depth_loss_mult * d_losses[:-1].sum()
    loss = loss + c.near_loss_mult * n_losses[-1] + 0.1 * c.near_loss_mult * n_losses[:-1].sum()
    loss = loss + c.empty_loss_mult * e_losses[-1] + 0.1 * c.empty_loss_mult * e_losses[:-1].sum()
    loss = loss + c.tv_loss_mult * tv[-1] + 0.1 * c.tv_loss_mult * tv[:-1].sum()
    loss = loss + 0.000001 * distr[-1] + 0.000001 * distr[:-1].sum()
    ret = raw['ret']
    K = pose.shape[0]
    if K > 0:
        offsets = ((pose - target[:, :3]) ** 2).sum().expand(L)
        ox = ((pose[:, 0] - target[:, 0]) ** 2).sum().expand(L)
        oy = ((pose[:, 1] - target[:, 1]) ** 2).sum().expand(L)
        oz = ((pose[:, 2] - target[:, 2]) ** 2).sum().expand(L)
        oyaw = ((yaw0 - target[:, 3:]) ** 2).sum().expand(L)
    else:
        offsets = ox = oy = oz = oyaw = torch.zeros(L, device=norms.device)
    sampling = torch.stack([x for r in ret for x in (r[4][0, 0], r[4][0, -1])])
    return dict(loss=loss, obj_losses=obj_losses, losses=losses, d_losses=d_losses, n_losses=n_losses,
                e_losses=e_losses, s_losses=s_losses, distr_losses=distr, tv_losses=tv, offsets=offsets,
                offset_x=ox, offset_y=oy, offset_z=oz, offset_yaw=oyaw, sampling_stats=sampling,
                weight_l2=raw['weight_l2'])


def train_step(model, config, rng, state, batch, lr, eps, alpha, prev, noise=None):
    """One optimization step (train_boxpose.py:49-321).

    batch: dict(rays=BoxRays, pixels[B,3], depth[B,1], sky[B,1], init[T,K,6], ext[K,3], ts, target[K,6])
    of device tensors for THIS rank's shard.  Returns (new_state, stats, rng, pose)."""
    variables = state.variables
    grad, raw, pose = loss_and_grad(model, config, rng, variables, batch, eps, alpha, prev, noise=noise)
    dist = _dist()
    world = 1
    if dist is not None:                                    # lax.pmean(grad) (:253)
        world = dist.get_world_size()
        dist.all_reduce(grad)
    yaw0 = raw['ret'][0][7][1]
    st = _assemble_stats(model, config, batch, raw, prev, pose, yaw0)
    if dist is not None:                                    # lax.pmean(stats) (:255), scalars only
        keys = sorted(k for k in st if torch.is_tensor(st[k]))
        flat = torch.cat([st[k].reshape(-1).float() for k in keys])
        dist.all_reduce(flat)
        flat /= world
        off = 0
        for k in keys:
            n = st[k].numel()
            st[k] = flat[off:off + n].reshape(st[k].shape)
            off += n
    gs = ops.clip_adam(variables.flat, state.m, state.v, grad, 1.0 / world, float(config.grad_max_val),
                       float(config.grad_max_norm), float(lr), state.step)
    new_state = TrainState(variables, state.m, state.v, state.step + 1)
    psnrs = dmath.mse_to_psnr(st['losses'])
    ret = raw['ret']
    stats = utils.Stats(
        loss=st['loss'], obj_losses=st['obj_losses'], losses=st['losses'], d_losses=st['d_losses'],
        n_losses=st['n_losses'], e_losses=st['e_losses'], s_losses=st['s_losses'],
        distr_losses=st['distr_losses'], tv_losses=st['tv_losses'], sampling_stats=st['sampling_stats'],
        offsets=st['offsets'], offset_x=st['offset_x'], offset_y=st['offset_y'], offset_z=st['offset_z'],
        offset_yaw=st['offset_yaw'], pose=pose, weights=[r[3] for r in ret], samples=[r[4] for r in ret],
        weight_l2=st['weight_l2'], psnr=psnrs[-1], psnrs=psnrs, obj_psnr=dmath.mse_to_psnr(st['obj_losses'])[-1],
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
