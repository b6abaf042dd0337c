"""MI355X-native MipNerfModel -- host-side mirror of internal/obbpose_model.py.

Same call surface as the reference (`MipNerfModel`, `construct_mipnerf`, `render_image`,
gin knobs with the reference's names); the body of `MipNerfModel.__call__`
(obbpose_model.py:68-261) runs as hand-written gfx950 kernels behind the C ABI of
libdurf_hip.so.  torch tensors stand in for jnp arrays; `variables` keeps flax's tree
names (`params/box_centers`, `params/MLP_0/Dense_i/{kernel,bias}`, `params/BoxMLP_k/...`)
as views into ONE flat fp32 buffer, so the data-parallel gradient exchange is a single
all-reduce (train_boxpose.py:253).
"""
import dataclasses
import math
from typing import Any

import torch

from . import ops
from . import utils

IN_BKGD, IN_OBJ, W_BKGD, W_OBJ = 60, 63, 256, 128


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
def _layer_shapes(width, in_dim, use_viewdirs=True):
    trunk = [(in_dim, width)] + [(width, width)] * 4 + [(width + in_dim, width)] + [(width, width)] * 2
    if not use_viewdirs:        # obbpose_model.py:336-352 without a condition: density head, then the rgb head straight off the trunk
        return trunk + [(width, 1), (width, 3)]
    return trunk + [(width, 1), (width, width), (width + 27, 128), (128, 3)]


class ParamLayout:
    """Offsets of every leaf in the flat fp32 parameter buffer (include/durf_hip.h)."""

    def __init__(self, T, K, use_viewdirs=True):
        # use_viewdirs=False: the 10-Dense tree of MipNerfModel.use_viewdirs = False (static model only: durf_amd/noview.py)
        self.T, self.K, self.use_viewdirs = T, K, bool(use_viewdirs)
        self.box = (0, T * K * 6)
        off = T * K * 6
        self.mlp_size = {W_BKGD: sum(a * b + b for a, b in _layer_shapes(W_BKGD, IN_BKGD, use_viewdirs)),
                         W_OBJ: sum(a * b + b for a, b in _layer_shapes(W_OBJ, IN_OBJ))}
        self.mlp_off = {'MLP_0': off}
        off += self.mlp_size[W_BKGD]
        for k in range(K):
            self.mlp_off['BoxMLP_%d' % k] = off
            off += self.mlp_size[W_OBJ]
        self.total = off

    def mlp_names(self):
        return ['MLP_0'] + ['BoxMLP_%d' % k for k in range(self.K)]

    @staticmethod
    def mlp_dims(name):
        return (W_BKGD, IN_BKGD) if name == 'MLP_0' else (W_OBJ, IN_OBJ)

    def layer_shapes(self, name):
        """(fan_in, fan_out) of Dense_0.. of MLP `name`, in parameter order"""
        width, in_dim = self.mlp_dims(name)
        return _layer_shapes(width, in_dim, self.use_viewdirs or name != 'MLP_0')


class Variables(dict):
    """flax-style {'params': {...}} tree whose leaves are views of `flat`."""

    def __init__(self, flat, layout):
        super().__init__()
        self.flat, self.layout = flat, layout
        p = {'box_centers': flat[layout.box[0]:layout.box[1]].view(layout.T, layout.K, 6)}
        for name in layout.mlp_names():
            width, in_dim = layout.mlp_dims(name)
            off = layout.mlp_off[name]
            d = {}
            for i, (fi, fo) in enumerate(layout.layer_shapes(name)):
                d['Dense_%d' % i] = {'kernel': flat[off:off + fi * fo].view(fi, fo),
                                     'bias': flat[off + fi * fo:off + fi * fo + fo]}
                off += fi * fo + fo
            p[name] = d
        self['params'] = p

    def mlp_flat(self, name):
        width, _ = self.layout.mlp_dims(name)
        off = self.layout.mlp_off[name]
        return self.flat[off:off + self.layout.mlp_size[width]]

    def like(self, flat):
        return Variables(flat, self.layout)


def init_boxes(rng, box_centers):
    """obbpose_model.py:35-39."""
    if box_centers.dim() < 3:
        return box_centers[:, None, :]
    return box_centers


# ----------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------
@dataclasses.dataclass
class MLP:
    """obbpose_model.py:293-303 (knobs only; evaluated by the fused MFMA kernel)."""
    net_depth: int = 8
    net_width: int = 256
    net_depth_condition: int = 1
    net_width_condition: int = 128
    net_activation: Any = 'relu'
    skip_layer: int = 4
    num_rgb_channels: int = 3
    num_density_channels: int = 1


@dataclasses.dataclass
class BoxMLP(MLP):
    """obbpose_model.py:357-367."""
    net_width: int = 128


def _check_mlp(m, width):
    ok = (m.net_depth == 8 and m.net_width == width and m.net_depth_condition == 1 and
          m.net_width_condition == 128 and m.skip_layer == 4 and m.num_rgb_channels == 3 and
          m.num_density_channels == 1 and m.net_activation == 'relu')
    if not ok:
        raise NotImplementedError('fused MLP kernels are built for the shipped gin topology '
                                  '(8x%d trunk, skip 4, 1x128 view layer, relu); got %r' % (width, m))


def _make_generator(rng, device):
    if isinstance(rng, torch.Generator):
        return rng
    g = torch.Generator(device=device)
    g.manual_seed(int(rng) if rng is not None else 0)
    return g


@dataclasses.dataclass
class MipNerfModel:
    """Nerf NN Model with both coarse and fine MLPs (obbpose_model.py:42-66 knobs)."""
    num_samples: int = 128
    num_levels: int = 2
    resample_padding: float = 0.01
    stop_level_grad: bool = True
    use_viewdirs: bool = True
    lindisp: bool = False
    ray_shape: str = 'cone'
    min_deg_point: int = 0
    max_deg_point: int = 10
    deg_view: int = 4
    num_objects: int = 2
    density_activation: Any = 'softplus'
    density_noise: float = 0.1
    density_bias: float = -1.
    rgb_activation: Any = 'sigmoid'
    rgb_padding: float = 0.001
    disable_integration: bool = False
    contraction: bool = True
    dynamics: bool = True
    timesteps: int = 5
    no_pose_opt: bool = False
    no_yaw_opt: bool = False
    # not a reference knob: 'bf16' = the fused MFMA kernels (production); 'f32' = the exact-fp32 parity instrument
    # (csrc/mlp_f32.hip: fp32 encodings, v_mfma_f32_32x32x2_f32 Dense layers, ~16x slower), which evaluates the
    # model in the reference's own arithmetic type (obbpose_model.py:326-327, internal/math.py:22-24)
    mlp_precision: str = 'bf16'
    # not a reference knob: arithmetic of the OBJECT branch (BoxMLP forward / backward / d(enc), hit rays only) when
    # mlp_precision == 'bf16'.  'auto' = 'f32' when box-pose optimisation is on (no_pose_opt / no_yaw_opt False, cfg4),
    # else 'bf16': d(loss)/d(box pose) is a sum over the hit rays that cancels to ~1 % of its summed magnitudes, so the
    # bf16 rounding of the object branch shows up as tens of per cent on it -- the background MLP, whose
    # weights only see MLP gradients, stays on the bf16 MFMA kernels either way.
    obj_precision: str = 'auto'

    def object_precision(self):
        """'bf16' or 'f32': which kernels the object branch runs on ('bf16x3' is the fp32 branch's data flow on split bf16
        operands: object_x3())"""
        if self.mlp_precision == 'f32':
            return 'f32'
        if self.obj_precision == 'auto':
            return 'bf16' if (self.no_pose_opt and self.no_yaw_opt) else 'f32'
        return 'f32' if self.obj_precision == 'bf16x3' else self.obj_precision

    def object_x3(self):
        """obj_precision = 'bf16x3' (round 6): the fp32 object branch -- records, weight gradients, pose gradient as under 'f32'
        -- with its forward / backward GEMMs on the bf16 matrix pipe as three products of (hi, lo) split operands
        (csrc/mlp_f32.hip chunk_mma_x3: ~2^-16 relative error per product instead of exact fp32)"""
        return self.mlp_precision != 'f32' and self.obj_precision == 'bf16x3'

    def _check(self):
        bad = []
        if self.mlp_precision not in ('bf16', 'f32'): bad.append('mlp_precision')
        if self.obj_precision not in ('auto', 'bf16', 'f32', 'bf16x3'): bad.append('obj_precision')
        if self.ray_shape not in ('cone', 'cylinder'): bad.append('ray_shape')
        if (self.min_deg_point, self.max_deg_point, self.deg_view) != (0, 10, 4): bad.append('degrees')
        if not self.dynamics and not (self.no_pose_opt and self.no_yaw_opt):
            # (the reference's pose gradient then runs through the BACKGROUND encoding of the box-hit rays, whose origins are
            # in box coordinates, obbpose_model.py:121-122: a backward through contraction + IPE that is not built)
            bad.append('dynamics=False with box-pose optimisation')
        if not self.stop_level_grad: bad.append('stop_level_grad=False')
        if self.num_samples % 32 or not (32 <= self.num_samples <= 256): bad.append('num_samples')
        if bad:
            raise NotImplementedError('knob values outside the shipped gin configs are not built: %s' % bad)
        _check_mlp(utils.configured(MLP), W_BKGD)
        _check_mlp(utils.configured(BoxMLP), W_OBJ)

    def _kernel_variables(self, variables):
        """the 12-Dense parameter tree the kernels evaluate: `variables` itself, or -- use_viewdirs=False, whose tree has no
        bottleneck and no view layer -- its embedding (durf_amd/noview.py)"""
        if variables.layout.use_viewdirs != bool(self.use_viewdirs) and not hasattr(variables, '_noview_of'):
            raise ValueError('MipNerfModel.use_viewdirs = %r but the parameter tree was built for %r'
                             % (self.use_viewdirs, variables.layout.use_viewdirs))
        if variables.layout.use_viewdirs:
            return variables
        if variables.layout.K and self.dynamics:
            raise NotImplementedError('use_viewdirs=False is the static model\'s knob (no boxes, or dynamics=False): with dynamic '
                                      'boxes the reference itself fails on it (obbpose_model.py:192-199: viewdirs_enc is undefined)')
        from . import noview
        return noview.embed(variables)

    # -- forward -------------------------------------------------------------
    def _forward(self, variables, rng, rays, init, ext, ts, randomized, rand_bkgd, white_bkgd, alpha,
                 train=False, noise=None, loss_prep=None, zero_fill=None):
        """loss_prep (training, num_levels >= 2): dict(lossmult, gt_depth, sky, eps, box_loss_mult, disable_multiscale,
        norms [L,5]) -- the inputs of durf_loss_prep; the fused per-ray launches then fill `norms` for every level.
        zero_fill: a flat fp32 tensor (the step's gradient buffer) the prologue launch zero fills on its way."""
        self._check()
        variables = self._kernel_variables(variables)
        lay = variables.layout
        K, N = lay.K, self.num_samples
        Kd = K if self.dynamics else 0         # dynamics=False: boxes only select rays (obbpose_model.py:167,232,257-260)
        B = rays.origins.shape[0]
        dev = rays.origins.device
        ts = int(ts)
        pc = variables['params']['box_centers']
        pose = pc[ts].contiguous()
        ext = ext.reshape(-1, 3).contiguous() if K > 0 else torch.zeros(0, 3, device=dev)
        radii = rays.radii.reshape(-1).contiguous()
        near, far = rays.near.reshape(-1).contiguous(), rays.far.reshape(-1).contiguous()
        f32 = self.mlp_precision == 'f32'
        obj_f32 = bool(Kd) and not f32 and self.object_precision() == 'f32'     # mixed: bf16 background, fp32 objects
        Kb = 0 if obj_f32 else Kd                                                # objects on the bf16 kernels
        # randomized without injected draws: the step's first launch draws them itself (durf_ray_prologue's Philox stream keyed
        # by the host's PRNG key `rng`; mip.py:364, math.py:257-260 draw inside the program too) -- no generator launch.  A
        # torch.Generator as `rng` keeps the torch.rand path (its state cannot key a counter-based stream).
        g = _make_generator(rng, dev) if randomized else None
        seed = None
        if randomized and noise is None:
            if isinstance(rng, torch.Generator):
                u = torch.rand(2, B, N + 1, device=dev, generator=g)      # one launch for both levels' noise
                noise = dict(t_rand=u[0], u_rand=u[1])
            else:
                seed = int(rng) if rng is not None else 0
        # (the fp32 object branch needs the box-hit rays' background evaluation as its own rows: always de-duplicated)
        use_dd = bool(Kd) and not f32 and (ops.DEDUP_HIT_RAYS or obj_f32)
        tail_side = trunk = None
        if obj_f32:
            # The background MLP's ONE evaluation of every box-hit ray is redone in fp32: those rays' rendered values --
            # and through them the head gradients that drive d(loss)/d(box pose) -- then carry no bf16 rounding at all
            # (DESIGN.md 2).  Its trunk sees the same input for every such ray and depends on the parameters only: one
            # workgroup, started here on a side stream so that it runs beside the small per-ray launches below (beside
            # the persistent MLP kernels it would wait for a CU and slow them: measured).
            # A training loop starts it even earlier: right behind the optimizer update of the previous step
            # (prefetch_const_trunk), where nothing else is running.
            tail_side = ops.side_stream(dev)
            cached = getattr(variables, '_trunk_cache', None)
            variables._trunk_cache = None
            if cached is not None and cached[1] == (variables.flat._version, ops.param_generation(variables.flat)):
                trunk = cached[0]
            else:
                tail_side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(tail_side):
                    trunk = ops.bkgd_const_trunk_f32(variables.mlp_flat('MLP_0'))
        # ray setup + view encoding + level-0 sample positions: one launch; both compactions: one launch
        # (the launch also snapshots the poses: the outputs must not alias the parameters the optimizer updates in place)
        pose_used = torch.empty_like(pose)
        # (the launch also packs every bf16 weight stream of the step: the K object MLPs sit back to back in the flat buffer)
        pack_arg = None
        if not f32:
            o0 = lay.mlp_off['BoxMLP_0'] if Kb else 0
            pack_arg = (variables.mlp_flat('MLP_0'), Kb, variables.flat[o0:o0 + Kb * lay.mlp_size[W_OBJ]] if Kb else None,
                        lay.mlp_size[W_OBJ], train)
        pro = ops.ray_prologue(rays.origins, rays.directions, pose, ext, rays.viewdirs, near, far, N,
                               noise['t_rand'] if (randomized and seed is None) else None, self.lindisp,
                               pose_copy=pose_used, zero=zero_fill, seed=seed, pack=pack_arg)
        o_s, d_s, hit, zo, view, t_vals0 = pro[:6]
        if seed is not None:
            noise = dict(t_rand=None, u_rand=pro[6])
        if use_dd:
            (idx, count, slot), cls = ops.compact_all(hit, N)     # cls also counts the boxes each ray hits
        else:
            (idx, count, slot), cls = ops.compact_hits(hit), None
        view27 = ops.view_enc(rays.viewdirs, want_f32=True)[1] if (f32 or obj_f32) else None
        packs = {}
        if not f32:
            pk_b, pk_o = pro[-1]
            packs = {'MLP_0': pk_b}
            if Kb:
                packs['obj'] = pk_o
        bk = ops.BKGD_RAND if rand_bkgd else (ops.BKGD_WHITE if white_bkgd else ops.BKGD_GREY)
        rows = B * N
        cyl = self.ray_shape == 'cylinder'
        view_tiles_obj = ops.obj_view_tiles(Kb, B, N, dev) if (train and Kb and not f32) else None
        ctx = dict(o_s=o_s, d_s=d_s, hit=hit, zo=zo, idx=idx, count=count, slot=slot, view=view,
                   packs=packs, levels=[], B=B, N=N, K=Kd, ts=ts, bkgd_mode=bk, view_tiles_obj=view_tiles_obj,
                   obj_f32=obj_f32 or (f32 and bool(Kd)), obj_x3=obj_f32 and self.object_x3())
        obj_flat = None
        if ctx['obj_f32']:                           # BoxMLP_0 .. BoxMLP_{K-1} sit back to back in the flat buffer
            o0 = lay.mlp_off['BoxMLP_0']
            obj_flat = variables.flat[o0:o0 + Kd * lay.mlp_size[W_OBJ]]
            ctx['obj_ws'] = ops.mlp_f32_pack(W_OBJ, IN_OBJ, obj_flat, K=Kd, param_stride=lay.mlp_size[W_OBJ], x3=ctx['obj_x3'])
        if f32:
            ctx['bkgd_ws'] = ops.mlp_f32_pack(W_BKGD, IN_BKGD, variables.mlp_flat('MLP_0'))
        raw_tail = None
        ret = []
        t_vals = weights = None
        box_rot0 = pose_used[0, 3:] if K > 0 else ops.const_tensor(dev, (3,))
        if cls is not None:
            dyn_mask = cls[3].reshape(B, 1)
        elif K > 1:
            dyn_mask = hit.sum(dim=-1, keepdim=True, dtype=torch.int32)
        else:
            dyn_mask = hit if K == 1 else ops.const_tensor(dev, (B, 1), torch.int32)
        if loss_prep is not None:
            if self.num_levels < 2:
                loss_prep = None
            else:
                loss_prep = dict(loss_prep, dyn=dyn_mask.reshape(-1).to(torch.int32).contiguous(), zo=zo)
        # Training fuses the per-ray stages (composite / resample are launch-bound at 4096 rays): a level
        # that is followed by another runs composite + resample + the loss normalisers of both levels as ONE launch
        # (durf_composite_resample); the last level launches no composite at all -- durf_loss_bwd recomputes it
        # anyway and fills this level's rgb / depth / acc / weights (`deferred`).  Inference keeps the plain calls.
        fused = train and loss_prep is not None
        # De-duplicated background evaluation (include/durf_hip.h, durf_expand_raw): a ray that hits exactly ONE box
        # feeds the background MLP the same trunk input at all its samples, so it is evaluated once (a "tail row" of
        # the same launch) and the other rays sample by sample on a compacted list.  Same results, ~hit-fraction less
        # background MLP work.  Rays that hit no box or several (the reference's garbage-in case) take the full path.
        dd = None
        if cls is not None:
            dd = dict(idx=cls[0], count=cls[1], slot=cls[2], nrows=cls[1][2:3], multi_hit=cls[1][3])
            ctx['dedup'] = dd
        t_next = None
        for lvl in range(self.num_levels):
            last = lvl == self.num_levels - 1
            if lvl == 0:
                t_vals = t_vals0
            elif t_next is not None:
                t_vals = t_next
            else:
                t_vals = ops.resample(t_vals, weights, self.resample_padding,
                                      _u_plane(noise, lvl - 1) if randomized else None)
            if f32:
                lvd = self._level_f32(variables, obj_flat, ctx, train, t_vals, o_s, d_s, radii, hit, Kd, cyl, view27, idx, count,
                                      alpha, B, N)
                raw_b, slabs = lvd['raw_b'], None
                enc_b = stash_b = mask_b = None
                raws = lvd['raws']
            else:
                lvd = None
                stash_b = torch.empty(ops.mlp_stash_bytes(W_BKGD, rows), dtype=torch.uint8, device=dev) if train else None
                mask_b = torch.empty(ops.mlp_mask_bytes(rows), dtype=torch.uint8, device=dev) if train else None
                side = ops.on_side(dev, bool(Kb) and ops.overlap_forward(rows))          # the object MLPs run in the shadow of the background MLP
                enc_kw = dict(contraction=self.contraction, disable_integration=self.disable_integration, cylinder=cyl)
                slabs = None
                # side-stream forward: the object launches are issued BEFORE the persistent background forward, which takes
                # every CU: they then run at its start instead of in its tail (round 4: 956-958 -> 961-963 k rays/s at cfg3)
                obj_first = bool(Kb) and side.enabled

                def launch_objects():                    # all K object MLPs of this level: one call (csrc/objects.hip)
                    with side:
                        ops.obj_fwd_batch(slabs, idx, count, t_vals, o_s, d_s, radii, alpha, view, packs['obj'][0],
                                          view_tile=view_tiles_obj if lvl == 0 else None,
                                          disable_integration=self.disable_integration, cylinder=cyl)
                if Kb:
                    slabs = ops.ObjSlabs(Kd, B, N, dev, train)      # allocated on the main stream, filled on the side one
                if obj_first:
                    # issued BEFORE the persistent background forward takes every CU: the object launches (one latency-
                    # bound round of ~100 workgroups) then run at its start instead of in its tail
                    side.fork()
                    launch_objects()
                vt = None
                if train and lvl == 0 and ops.FUSED_ENCODE:   # ... and writes the view-direction tile of the weight-gradient launch
                    vt = ctx['view_tile'] = torch.empty(ops.tile_rows(rows), ops.VIEW_DIM, dtype=torch.bfloat16, device=dev)
                # a small training step (one stream): the K object MLPs' forward rides in the background MLP's persistent
                # launch as (object, tile pair) items (durf_mlp_fwd_enc_obj, round 6) -- bit-identical to the two launches
                mixed = (train and bool(Kb) and dd is not None and ops.FUSED_ENCODE and ops.FWD_SCATTER_RAW and
                         not side.enabled and ops.obj_mix(rows))
                if mixed:
                    scatter = True
                    raw_c, enc_b = ops.mlp_fwd_enc_obj(rows, N, t_vals, o_s, d_s, radii, hit, view, packs['MLP_0'][0], slabs, idx,
                                                       count, alpha, packs['obj'][0], dd['idx'][0], dd['count'][0:1], dd['idx'][1],
                                                       dd['count'][1:2], stash_b, mask_b, view_tile=vt,
                                                       obj_view_tile=view_tiles_obj if lvl == 0 else None, **enc_kw)
                    obj_first = True                          # (the objects are done: nothing left to launch below)
                elif dd is not None and ops.FUSED_ENCODE:       # the forward encodes its own tiles (durf_mlp_fwd_enc)
                    side.fork()
                    # (raw straight in the full layout unless the box-hit rays' rows come from the fp32 evaluation, raw_tail)
                    scatter = not obj_f32 and ops.FWD_SCATTER_RAW
                    raw_c, enc_b = ops.mlp_fwd_enc(rows, N, t_vals, o_s, d_s, radii, hit, view, packs['MLP_0'][0],
                                                   ray_idx=dd['idx'][0], count=dd['count'][0:1], stash=stash_b,
                                                   relu_mask=mask_b, tail_idx=dd['idx'][1], tail_count=dd['count'][1:2],
                                                   view_tile=vt, raw_full=scatter, **enc_kw)
                elif dd is not None:
                    scatter = False
                    enc_b, _ = ops.encode_bkgd(t_vals, o_s, d_s, radii, hit, self.contraction,
                                               disable_integration=self.disable_integration, cylinder=cyl,
                                               idx=dd['idx'][0], count=dd['count'][0:1])
                    side.fork()
                    raw_c = ops.mlp_fwd(W_BKGD, rows, N, enc_b, view, packs['MLP_0'][0], ray_idx=dd['idx'][0],
                                        count=dd['count'][0:1], stash=stash_b, relu_mask=mask_b,
                                        tail_idx=dd['idx'][1], tail_count=dd['count'][1:2])
                if dd is not None:
                    if tail_side is not None:                # the fp32 hit-ray evaluation (side stream) must have landed
                        torch.cuda.current_stream().wait_stream(tail_side)
                        tail_side = None
                    if obj_f32 and lvl == 0:
                        # the view layer + rgb head per box-hit ray (same input at both levels: once per step), HERE, on
                        # the main stream behind the level-0 forward: beside that persistent launch it found no CU until its
                        # tail and the main stream waited for it (160-180 us instead of 40; ~10 us now that it has the chip)
                        if trunk is not None:
                            trunk.record_stream(torch.cuda.current_stream())      # (produced on the side stream's pool)
                        raw_tail = ops.bkgd_hit_rays_f32(B, view27, variables.mlp_flat('MLP_0'), dd['idx'][1],
                                                         dd['count'][1:2], trunk=trunk)
                    raw_b = raw_c if scatter else ops.expand_raw(B, N, raw_c, dd['slot'], dd['count'], raw_tail=raw_tail)
                elif ops.FUSED_ENCODE:
                    side.fork()
                    raw_b, enc_b = ops.mlp_fwd_enc(rows, N, t_vals, o_s, d_s, radii, hit if Kd else None, view,
                                                   packs['MLP_0'][0], stash=stash_b, relu_mask=mask_b, view_tile=vt, **enc_kw)
                else:
                    enc_b, _ = ops.encode_bkgd(t_vals, o_s, d_s, radii, hit if Kd else None, self.contraction,
                                               disable_integration=self.disable_integration, cylinder=cyl)
                    side.fork()
                    raw_b = ops.mlp_fwd(W_BKGD, rows, N, enc_b, view, packs['MLP_0'][0], stash=stash_b, relu_mask=mask_b)
                if obj_f32:                              # object branch in exact fp32 (object_precision)
                    lvd = self._objects_f32(obj_flat, lay.mlp_size[W_OBJ], ctx['obj_ws'], train, t_vals, o_s, d_s, radii, Kd, cyl,
                                            view27, idx, count, alpha, B, N)
                if Kb:
                    if not obj_first:
                        launch_objects()
                    side.join()
                raws = slabs.raws() if Kb else (lvd['raws'] if obj_f32 else [])
            if randomized and self.density_noise > 0:    # :236-240: one launch (the draws: injected, the library's own
                # under the host's key, or a torch.Generator's)
                if 'density' in noise:
                    ops.density_noise(raw_b, self.density_noise, normal=noise['density'][lvl])
                elif seed is not None:
                    ops.density_noise(raw_b, self.density_noise, seed=seed, level=lvl)
                else:
                    ops.density_noise(raw_b, self.density_noise, normal=torch.randn(B, N, device=dev, generator=g))
            deferred = False
            if fused and not last:
                rgb, depth, acc, weights, t_mids, t_dists, t_next = ops.composite_resample(
                    raw_b, raws, slot, t_vals, d_s, self.density_bias, bk, self.resample_padding,
                    _u_plane(noise, lvl) if randomized else None, prep=dict(loss_prep, level=lvl))
            elif fused:
                t_next = None
                rgb, depth, acc = torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
                weights, t_mids, t_dists = (torch.empty(B, N, device=dev) for _ in range(3))
                deferred = True
            else:
                t_next = None
                rgb, depth, acc, weights, t_mids, t_dists = ops.composite_fwd(
                    raw_b, raws, slot, t_vals, d_s, self.density_bias, bk)
            ret.append((rgb, depth, acc, weights, t_vals, t_mids, t_dists, [pose_used[:, :3], box_rot0],
                        dyn_mask, zo))
            if train:
                ctx['levels'].append(dict(t_vals=t_vals, enc_b=enc_b, raw_b=raw_b, stash_b=stash_b,
                                          raws=raws, slabs=slabs, mask_b=mask_b, rgb=rgb, depth=depth,
                                          acc=acc, weights=weights, t_mids=t_mids, t_dists=t_dists, deferred=deferred,
                                          f32=lvd))
        return ret, ctx

    def _level_f32(self, variables, obj_flat, ctx, train, t_vals, o_s, d_s, radii, hit, Kd, cyl, view27, idx, count, alpha, B, N):
        """encodings + MLPs of one level in exact fp32 (mlp_precision='f32'): accurate-libm encodings emitted as
        fp32, Dense layers on v_mfma_f32_32x32x2_f32 straight from the fp32 parameters."""
        rows = B * N
        _, enc = ops.encode_bkgd(t_vals, o_s, d_s, radii, hit if Kd else None, self.contraction, tile=False, f32=True,
                                 disable_integration=self.disable_integration, cylinder=cyl)
        out = ops.mlp_fwd_f32(W_BKGD, IN_BKGD, rows, N, enc, view27, variables.mlp_flat('MLP_0'), want_act=train,
                              wstream=ctx['bkgd_ws'])
        d = dict(raw_b=out[0] if train else out, act_b=out[1] if train else None, raws=[], slabs32=None)
        if Kd:
            d.update(self._objects_f32(obj_flat, variables.layout.mlp_size[W_OBJ], ctx['obj_ws'], train, t_vals, o_s, d_s, radii,
                                       Kd, cyl, view27, idx, count, alpha, B, N))
        return d

    def _objects_f32(self, obj_flat, stride, ws, train, t_vals, o_s, d_s, radii, Kd, cyl, view27, idx, count, alpha, B, N):
        """the K object MLPs of one level in exact fp32, hit rays only (obbpose_model.py:167-201): accurate-libm fp32
        encodings + the fp32 MFMA forward of all K objects, one launch each (csrc/mlp_f32.hip, durf_objf32_*)"""
        slabs = ops.ObjSlabsF32(Kd, B, N, t_vals.device, train)
        ops.objf32_fwd_batch(slabs, idx, count, t_vals, o_s, d_s, radii, alpha, view27, obj_flat, stride, ws,
                             disable_integration=self.disable_integration, cylinder=cyl, x3=self.object_x3())
        return dict(raws=slabs.raws(), slabs32=slabs)

    def prefetch_const_trunk(self, variables):
        """Training loops call this right after the optimizer update: when the next step will evaluate the box-hit rays
        in fp32 (object_precision() == 'f32'), the background trunk on their constant input -- a function of the
        parameters alone, one workgroup reading 2.4 MB of cold weights -- is started NOW on the side stream, where it
        overlaps the step's tail and the next step's small per-ray launches instead of the persistent MLP kernels (beside
        those it waits for a CU and delays the workgroup that finally shares it: measured +40 us on the forward).
        Used once, and only if the parameters have not been touched since: through torch (tensor version counter) or
        through this library's in-place updates, which torch does not see (ops.param_generation)."""
        lay = variables.layout
        if self.mlp_precision != 'bf16' or not self.dynamics or lay.K == 0 or self.object_precision() != 'f32':
            return
        if variables.flat.device.type != 'cuda':
            return
        side = ops.side_stream(variables.flat.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            trunk = ops.bkgd_const_trunk_f32(variables.mlp_flat('MLP_0'))
        variables._trunk_cache = (trunk, (variables.flat._version, ops.param_generation(variables.flat)))

    def supports_one_call(self, variables, randomized=False):
        """whether durf_forward (apply_one_call) covers this model's inference path"""
        K = variables.layout.K
        return (self.mlp_precision == 'bf16' and not (K and (not self.dynamics or self.object_precision() != 'bf16')) and
                variables.flat.device.type == 'cuda')

    def apply_one_call(self, variables, rng, rays, init, ext, ts, randomized, rand_bkgd, white_bkgd, alpha, noise=None):
        """`apply` through ONE library call (durf_forward, csrc/forward.hip): the orchestration of `_forward(train=False)`
        done in C for hosts that are not Python; same arguments, same list of 10-tuples, bit-identical results
        (tests/test_gpu_forward_call.py).  bf16 MLPs, objects on the bf16 kernels.  noise: dict(t_rand, u_rand[, density:
        the standard-normal draws [B,N] per level]) injected draws; otherwise `rng` keys the library's own (an int) or is the
        torch.Generator they come from."""
        self._check()
        variables = self._kernel_variables(variables)
        lay = variables.layout
        K = lay.K
        if not self.supports_one_call(variables, randomized):
            raise NotImplementedError('durf_forward covers the bf16 inference path (dynamics=True, bf16 object MLPs)')
        B, N = rays.origins.shape[0], self.num_samples
        dev = rays.origins.device
        seed = None
        if randomized and noise is None:
            if isinstance(rng, torch.Generator):
                u = torch.rand(2, B, N + 1, device=dev, generator=rng)
                noise = dict(t_rand=u[0], u_rand=u[1])
                if self.density_noise > 0:          # (level by level, as apply() draws them)
                    noise['density'] = [torch.randn(B, N, device=dev, generator=rng) for _ in range(self.num_levels)]
            else:                                   # the library draws (durf_forward_args.draw_noise), as in apply()
                seed = int(rng) if rng is not None else 0
                noise = dict(t_rand=None, u_rand=None)
        dn = self.density_noise if (randomized and self.density_noise > 0) else 0.0
        if dn and seed is None and 'density' not in noise:      # injected sampling draws only: the generator apply() falls back to
            gd = _make_generator(rng, dev)
            noise = dict(noise, density=[torch.randn(B, N, device=dev, generator=gd) for _ in range(self.num_levels)])
        pose = variables['params']['box_centers'][int(ts)].contiguous()
        flags = ((ops.ENC_CONTRACT if self.contraction else 0) | (ops.ENC_NO_INTEGRATION if self.disable_integration else 0) |
                 (ops.ENC_CYLINDER if self.ray_shape == 'cylinder' else 0))
        o0 = lay.mlp_off['BoxMLP_0'] if K else 0
        bk = ops.BKGD_RAND if rand_bkgd else (ops.BKGD_WHITE if white_bkgd else ops.BKGD_GREY)
        outs, dyn, zo = ops.forward_call(
            rays, pose, ext.reshape(-1, 3).contiguous() if K else None, variables.mlp_flat('MLP_0'),
            variables.flat[o0:o0 + K * lay.mlp_size[W_OBJ]] if K else None, lay.mlp_size[W_OBJ], N, self.num_levels, alpha, flags,
            lindisp=self.lindisp, bkgd_mode=bk, density_bias=self.density_bias, resample_padding=self.resample_padding,
            t_rand=noise['t_rand'] if randomized else None, u_rand=noise['u_rand'] if randomized else None, seed=seed,
            density_noise=dn, density_rand=noise.get('density') if dn else None)
        box_rot0 = pose[0, 3:] if K > 0 else ops.const_tensor(dev, (3,))
        return [tuple(o) + ([pose[:, :3], box_rot0], dyn, zo) for o in outs]

    def render_image_one_call(self, variables, rays, init, ext, ts, white_bkgd, alpha, chunk=8192):
        """render_image (obbpose_model.py:421-479) for one device as ONE library call (durf_render_image, csrc/forward.hip): the
        chunk loop runs in C over the image's rays where they are -- no per-chunk slicing, output allocation or argument
        marshalling in the interpreter.  rays: [H, W, .] fields -> (rgb [H,W,3], distance [H,W], acc [H,W]), bit-identical to
        render_image over apply_one_call chunks.  Same scope as apply_one_call (supports_one_call)."""
        self._check()
        variables = self._kernel_variables(variables)
        lay = variables.layout
        K = lay.K
        if not self.supports_one_call(variables):
            raise NotImplementedError('durf_render_image covers the bf16 inference path (dynamics=True, bf16 object MLPs)')
        height, width = rays[0].shape[:2]
        flat = utils.namedtuple_map(lambda r: r.reshape(height * width, -1), rays)
        pose = variables['params']['box_centers'][int(ts)].contiguous()
        flags = ((ops.ENC_CONTRACT if self.contraction else 0) | (ops.ENC_NO_INTEGRATION if self.disable_integration else 0) |
                 (ops.ENC_CYLINDER if self.ray_shape == 'cylinder' else 0))
        o0 = lay.mlp_off['BoxMLP_0'] if K else 0
        rgb, dist_, acc = ops.render_image_call(
            flat, pose, ext.reshape(-1, 3).contiguous() if K else None, variables.mlp_flat('MLP_0'),
            variables.flat[o0:o0 + K * lay.mlp_size[W_OBJ]] if K else None, lay.mlp_size[W_OBJ], self.num_samples, self.num_levels,
            alpha, flags, chunk, lindisp=self.lindisp, bkgd_mode=ops.BKGD_WHITE if white_bkgd else ops.BKGD_GREY,
            density_bias=self.density_bias, resample_padding=self.resample_padding)
        return rgb.reshape(height, width, 3), dist_.reshape(height, width), acc.reshape(height, width)

    def apply(self, variables, rng, rays, init, ext, ts, randomized, rand_bkgd, white_bkgd, alpha,
              noise=None):
        """model.apply(variables, key, rays, init, ext, ts, randomized=, rand_bkgd=, white_bkgd=,
        alpha=) -> list[num_levels] of (rgb, depth, acc, weights, t_vals, t_mids, t_dists,
        [box_pose, box_rot0], dyn_mask, zo)   (train_boxpose.py:82-92, obbpose_model.py:258)."""
        return self._forward(variables, rng, rays, init, ext, ts, randomized, rand_bkgd, white_bkgd,
                             alpha, train=False, noise=noise)[0]

    __call__ = apply


def _u_plane(noise, level):
    """the resampling draws of the resample behind `level`: plane `level` of the library's own draws ([3,B,N+1]: Philox words
    1, 2, 3), or the one [B,N+1] array of injected draws (tests: the same array at every level)"""
    u = noise['u_rand']
    return u[level] if (u is not None and u.dim() == 3) else u


def glorot_uniform_(t, fan_in, fan_out, gen):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    t.copy_((torch.rand(t.shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(lim).to(t.dtype))


def construct_mipnerf(rng, example_batch, device='cuda'):
    """Construct a Neural Radiance Field (obbpose_model.py:264-291).

    box_centers is initialised to the batch's `init` verbatim (:35-39,88); Dense kernels are
    glorot-uniform, biases zero (flax defaults).  `rng`: int seed or CPU torch.Generator."""
    model = utils.configured(MipNerfModel)
    init = torch.as_tensor(example_batch['init']).squeeze()
    if init.dim() == 2:
        init = init[:, None, :]
    if init.dim() == 1:    # K == 0
        init = init.reshape(init.shape[0] if init.numel() else model.timesteps, 0, 6)
    T, K = init.shape[0], init.shape[1]
    layout = ParamLayout(T, K, model.use_viewdirs)
    gen = rng if isinstance(rng, torch.Generator) else torch.Generator().manual_seed(int(rng))
    flat = torch.zeros(layout.total, dtype=torch.float32)
    v = Variables(flat, layout)
    v['params']['box_centers'].copy_(init_boxes(None, init.float().cpu()))
    for name in layout.mlp_names():
        for i in range(len(layout.layer_shapes(name))):
            k = v['params'][name]['Dense_%d' % i]['kernel']
            glorot_uniform_(k, k.shape[0], k.shape[1], gen)
    return model, Variables(flat.to(device), layout)


def render_image(render_fn, rays, init, ext, ts, rng, alpha, chunk=8192):
    """Render all the pixels of an image in test mode (obbpose_model.py:421-479).

    render_fn(rng, batch) -> list of per-level tuples (as the pmapped render_eval_fn,
    train_boxpose.py:377-390); batch keys: rays, init, ext, ts, alpha.

    Multi-rank (torch.distributed initialised, world > 1): like the reference, every chunk is edge-padded to a
    multiple of the device count (:454-458), each rank renders its contiguous slice of the chunk, and the slices
    are all-gathered (the `jax.lax.all_gather` of train_boxpose.py:379) and un-padded, so every rank returns the
    full image."""
    import torch.distributed as dist
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(), dist.get_rank()
    height, width = rays[0].shape[:2]
    num_rays = height * width
    rays = utils.namedtuple_map(lambda r: r.reshape(num_rays, -1), rays)
    results = []
    for i in range(0, num_rays, chunk):
        chunk_rays = utils.namedtuple_map(lambda r: r[i:i + chunk], rays)
        n = chunk_rays[0].shape[0]
        pad = (-n) % world
        if pad:                                      # jnp.pad(..., mode='edge'): repeat the last ray
            chunk_rays = utils.namedtuple_map(lambda r: torch.cat([r, r[-1:].expand(pad, -1)], 0), chunk_rays)
        per = (n + pad) // world
        mine = utils.namedtuple_map(lambda r: r[rank * per:(rank + 1) * per].contiguous(), chunk_rays)
        batch = dict(rays=mine, init=init, ext=ext, ts=ts, alpha=alpha)
        last = render_fn(rng, batch)[-1][:3]
        if world > 1:
            gathered = []
            for x in last:
                x = x.contiguous()
                parts = [torch.empty_like(x) for _ in range(world)]
                dist.all_gather(parts, x)
                gathered.append(torch.cat(parts, 0)[:n])         # utils.unshard(x, padding)
            last = gathered
        results.append(last)
    rgb, distance, acc = [torch.cat(r, dim=0) for r in zip(*results)]
    return (rgb.reshape(height, width, -1), distance.reshape(height, width), acc.reshape(height, width))
