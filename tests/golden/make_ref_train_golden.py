#!/usr/bin/env python3
"""Generate tests/golden/ref_train_*.npz: what the REFERENCE's own train_step computes on seeded inputs.

Build container only.  /root/reference/train_boxpose.py is imported unmodified under the numpy stand-ins of
tests/ref_standin.py and its `train_step` (:49-321) runs on the reference's MipNerfModel (float64; PRNG draws replayed).
Committed per case: the PRNG draws, a parameter checksum, every logged scalar of the reference's `loss_fn`, and the
derivative of the reference's loss by central differences of the reference's own `loss_fn` closure with
`lax.stop_gradient` replayed (tests/test_reference_train_crosscheck.py explains the method and checks the oracle against
the same runs), along three families of directions:
  * `derivatives`: seeded random directions, one per parameter group (all of MLP_0, each BoxMLP, the step's box poses);
  * `grad_dir_derivatives`: per parameter group the direction g / |g| of the oracle's own gradient g restricted to the
    group -- along it the reference's central difference IS the norm of the reference's gradient, so a gradient that is
    zero, or scaled, or rotated away by more than the tolerance cannot reproduce it (the random directions alone cannot
    tell: a dense Gaussian v over 594 308 entries makes tol * |g| * |v| hundreds of times the derivative);
  * `layer_dir_derivatives`: the same per Dense kernel of MLP_0 (12 directions), which pins the split of the gradient
    over the layers.
The directions are not stored: a machine without /root/reference regenerates them from the oracle (float64 autograd on
the committed seeds) and `grad_norms` checks that it got the same ones.
tests/test_golden_ref_train.py checks the oracle (CPU) and the HIP path (GPU) against these vectors on machines that have
no /root/reference.  A fixture is data -- seeds, draws, expected outputs -- no reference text.
    python tests/golden/make_ref_train_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from durf_amd import obbpose_model, synthetic, utils  # noqa: E402
from oracle import durf_ref as R  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import ref_standin  # noqa: E402
from tests.ref_standin import Hooks, StopGrad, Uniform  # noqa: E402

SCALARS = ('loss', 'losses', 'obj_losses', 'd_losses', 'n_losses', 'e_losses', 's_losses', 'distr_losses', 'tv_losses',
           'sampling_stats', 'offsets', 'offset_x', 'offset_y', 'offset_z', 'offset_yaw', 'weight_l2')
STEPS = (1e-7, 1e-8, 1e-9)
LAYER_STEPS = (1e-7, 1e-8)

CASES = {
    # Waymo knobs, pose optimisation with the TV prior, box-weighted rgb loss, weight decay, stratified sampling
    'K3_pose_opt_rand': dict(B=96, K=3, N=32, seed=311, alpha=4.5, eps=0.7,
                             config=dict(randomized=True, tv_loss_mult=1e-2, box_loss_mult=2, weight_decay_mult=1e-3),
                             model=dict(no_pose_opt=False, no_yaw_opt=False)),
    # frozen poses, deterministic sampling, white background, single-scale loss
    'K1_frozen_det': dict(B=64, K=1, N=32, seed=312, alpha=10.0, eps=3.0,
                          config=dict(randomized=False, white_bkgd=True, disable_multiscale_loss=True),
                          model=dict(no_pose_opt=True, no_yaw_opt=True)),
    # yaw only (positions frozen), dynamic model, cone rays: the yaw-only switch on a configuration the product runs
    'K2_yaw_only': dict(B=64, K=2, N=32, seed=318, alpha=7.0, eps=0.2,
                        config=dict(randomized=True, tv_loss_mult=1e-3),
                        model=dict(no_pose_opt=True, no_yaw_opt=False)),
    # yaw only, static model (boxes select rays only), cylinder rays: a knob combination the product rejects
    # (MipNerfModel._check) -- pins the oracle only, see HIP_CASES
    'K2_yaw_only_static': dict(B=64, K=2, N=32, seed=313, alpha=10.0, eps=0.2,
                               config=dict(randomized=True, tv_loss_mult=1e-3),
                               model=dict(no_pose_opt=True, no_yaw_opt=False, dynamics=False, ray_shape='cylinder')),
}
HIP_CASES = ('K3_pose_opt_rand', 'K1_frozen_det', 'K2_yaw_only')        # the cases the HIP train step is held to


class Optimizer:
    def __init__(self, target):
        self.target, self.applied, self.lr = target, None, None

    def apply_gradient(self, grad, learning_rate=None):
        new = Optimizer(self.target)
        new.applied, new.lr = grad, learning_rate
        return new


class State:
    def __init__(self, optimizer):
        self.optimizer = optimizer

    def replace(self, optimizer):
        return State(optimizer)


def setup(case):
    c = CASES[case]
    B, K, N, seed = c['B'], c['K'], c['N'], c['seed']
    b = synthetic.make_batch(B, K, seed=seed, noise_boxes=0.3)
    cb = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in b.items() if k != 'rays'}
    utils.clear_gin()
    _, variables = obbpose_model.construct_mipnerf(seed, cb, device='cpu')
    g = torch.Generator().manual_seed(seed)
    for nm in variables.layout.mlp_names():
        for i in range(12):
            bias = variables['params'][nm]['Dense_%d' % i]['bias']
            bias.copy_((torch.rand(bias.shape, generator=g) - 0.5) * 0.1)
    dt = torch.float64
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g, dtype=dt), u_rand=torch.rand(B, N + 1, generator=g, dtype=dt))
    ob = H.oracle_batch(b, dt)
    params = H.oracle_params_from_variables(variables, dt)
    # poses a step away from the initial ones, and a `prev` that differs from both: the TV and offset terms are non-trivial
    params['box_centers'] = params['box_centers'] + 0.05 * torch.randn(params['box_centers'].shape, generator=g, dtype=dt)
    prev = ob['init'][0:1] + 0.02 * torch.randn(ob['init'][0:1].shape, generator=g, dtype=dt)
    config = dict(R.CONFIG_DEFAULTS, **c['config'])
    model_cfg = dict(num_samples=N, density_noise=0.0, **c['model'])
    return c, b, ob, params, prev, noise, config, model_cfg


def oracle(params, ob, config, model_cfg, c, prev, noise):
    leaves = [z.detach().clone().requires_grad_(True) for z in R.params_leaves(params)]
    p = R.set_leaves(params, leaves)
    loss, S, _ = R.loss_fn(p, ob, config, model_cfg, c['eps'], c['alpha'], prev, noise=noise if config['randomized'] else None)
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    grads = [torch.zeros_like(z) if gr is None else gr for gr, z in zip(grads, leaves)]
    return S, grads


def tree_of(params, leaves):
    """flat oracle leaves -> flax-shaped numpy tree"""
    return ref_standin.flax_tree(R.set_leaves(params, list(leaves)))


def ref_batch(ref, ob, b):
    f = lambda t: t.detach().double().numpy()
    rays = ref.utils.BoxRays(*[f(getattr(ob['rays'], n)) for n in ref.utils.BoxRays._fields])
    return dict(rays=rays, init=f(ob['init']), ext=f(ob['ext']), ts=np.array([int(b['ts'])]), depth=f(ob['depth']),
                sky=f(ob['sky']), pixels=f(ob['pixels']), target=f(ob['target']))


def owners(params):
    """which parameter group each leaf of R.params_leaves belongs to"""
    out = ['box_centers']
    names = ['MLP_0'] + sorted([k for k in params if k.startswith('BoxMLP_')], key=lambda s: int(s.split('_')[1]))
    for n in names:
        out += [n, n] * len(params[n])
    return out


def directions(params, b, seed):
    """seeded random directions, one per parameter group: [(group, [v per leaf of R.params_leaves])]"""
    leaves = R.params_leaves(params)
    gen = torch.Generator().manual_seed(seed + 1)
    ts = int(b['ts'])
    out = []
    for target in ['box_centers'] + [n for n in params if n != 'box_centers']:
        vs = []
        for leaf, owner in zip(leaves, owners(params)):
            v = torch.randn(leaf.shape, generator=gen, dtype=torch.float64) if owner == target else torch.zeros_like(leaf)
            if owner == target == 'box_centers':        # only this step's timestep row takes part
                keep = torch.zeros_like(v)
                keep[ts] = 1.0
                v = v * keep
            vs.append(v)
        out.append((target, vs))
    return out


def groups(params):
    return ['box_centers'] + [n for n in params if n != 'box_centers']


def gradient_directions(params, grads):
    """per parameter group: the gradient restricted to the group, normalised -> [(group, [v per leaf]), ...], norms.
    `grads`: one tensor per leaf of R.params_leaves (the oracle's autograd gradient).  Along g / |g| the directional
    derivative is |g|: what a zeroed, scaled or mis-directed gradient cannot reproduce."""
    own = owners(params)
    out, norms = [], []
    for target in groups(params):
        nrm = float(sum(float((g * g).sum()) for g, o in zip(grads, own) if o == target)) ** 0.5
        vs = [(g / nrm) if (o == target and nrm > 0.0) else torch.zeros_like(g) for g, o in zip(grads, own)]
        out.append((target, vs))
        norms.append(nrm)
    return out, norms


def layer_directions(params, grads, group='MLP_0'):
    """the same per Dense kernel of one MLP: [('MLP_0/Dense_3', [v per leaf]), ...], norms"""
    own = owners(params)
    first = own.index(group)
    out, norms = [], []
    for i in range(len(params[group])):
        li = first + 2 * i                                 # leaves of an MLP: kernel, bias per Dense
        nrm = float(grads[li].norm())
        vs = [torch.zeros_like(g) for g in grads]
        if nrm > 0.0:
            vs[li] = grads[li] / nrm
        out.append(('%s/Dense_%d' % (group, i), vs))
        norms.append(nrm)
    return out, norms


def run_reference(ref, case, grad_tree=None, lr=5e-4):
    """the reference's train_step on the case -> (new_state, stats, pose, loss closure(tree, mode) -> float, inputs)"""
    c, b, ob, params, prev, noise, config, model_cfg = setup(case)
    rconf = ref.utils.Config(**{k: v for k, v in config.items() if k in ref.utils.Config.__dataclass_fields__})
    model = ref.obbpose_model.MipNerfModel(**model_cfg)
    tree = ref_standin.flax_tree(params)
    Hooks.grad_provider = (lambda x: grad_tree) if grad_tree is not None else None
    uniforms = [noise['t_rand'].numpy(), noise['u_rand'].numpy()] if config['randomized'] else []
    Uniform.queue = [u.copy() for u in uniforms]
    StopGrad.start(None)
    state = State(Optimizer(tree))
    new_state, stats, _, pose = ref.train_boxpose.train_step(model, rconf, 0, state, ref_batch(ref, ob, b), lr, c['eps'],
                                                             c['alpha'], prev.numpy())
    assert not Uniform.queue
    loss_fn = Hooks.loss_fn
    # (train_step re-binds its argument `eps` to 1e-6 for nan_to_num AFTER differentiating (:262); the closure shares that
    # variable, so evaluated later it would see the near-loss interval 1e-6: put the step's value back in the cell)
    loss_fn.__closure__[loss_fn.__code__.co_freevars.index('eps')].cell_contents = c['eps']

    def ref_loss(tr, mode):
        Uniform.queue = [u.copy() for u in uniforms]
        StopGrad.start(mode)
        return float(loss_fn(tr)[0])
    Hooks.grad_provider = None
    return new_state, stats, pose, ref_loss, (c, b, ob, params, prev, noise, config, model_cfg, tree)


def reference_derivatives(ref_loss, params, tree, dirs, steps=STEPS):
    """central differences of the reference's loss along each direction, at every step size of `steps`: [groups, steps]"""
    leaves = R.params_leaves(params)
    ref_loss(tree, 'record')
    out = np.zeros((len(dirs), len(steps)))
    for gi, (_, vs) in enumerate(dirs):
        if all(float(v.abs().max()) == 0.0 for v in vs):      # a group no ray reaches: derivative 0, nothing to run
            continue
        for hi, h in enumerate(steps):
            plus = tree_of(params, [z + h * v for z, v in zip(leaves, vs)])
            minus = tree_of(params, [z - h * v for z, v in zip(leaves, vs)])
            out[gi, hi] = (ref_loss(plus, 'replay') - ref_loss(minus, 'replay')) / (2 * h)
    StopGrad.start(None)
    return out


def main():
    gold = os.path.join(ROOT, 'tests', 'golden')
    ref = ref_standin.load(train=True)
    try:
        for case in CASES:
            _, stats, pose, ref_loss, (c, b, ob, params, prev, noise, config, model_cfg, tree) = run_reference(ref, case)
            dirs = directions(params, b, c['seed'])
            rec = {k: np.asarray(getattr(stats, k), dtype=np.float64) for k in SCALARS}
            rec['pose'] = np.asarray(pose, dtype=np.float64)
            rec['derivatives'] = reference_derivatives(ref_loss, params, tree, dirs)
            _, grads = oracle(params, ob, config, model_cfg, c, prev, noise)
            gdirs, gnorms = gradient_directions(params, grads)
            ldirs, lnorms = layer_directions(params, grads)
            rec['grad_norms'], rec['layer_grad_norms'] = np.array(gnorms), np.array(lnorms)
            rec['grad_dir_derivatives'] = reference_derivatives(ref_loss, params, tree, gdirs)
            rec['layer_dir_derivatives'] = reference_derivatives(ref_loss, params, tree, ldirs, LAYER_STEPS)
            flat = torch.cat([z.reshape(-1) for z in R.params_leaves(params)])
            rec['param_checksum'] = np.array([float(flat.sum()), float((flat * flat).sum())])
            rec['t_rand'], rec['u_rand'] = noise['t_rand'].numpy(), noise['u_rand'].numpy()
            path = os.path.join(gold, 'ref_train_' + case + '.npz')
            np.savez_compressed(path, **rec)
            print(case, '%.1f KB' % (os.path.getsize(path) / 1024), 'loss', float(rec['loss']), 'derivatives', rec['derivatives'][:, 1])
            print('   |g| per group: oracle', rec['grad_norms'], 'reference', rec['grad_dir_derivatives'][:, 1])
            print('   |g| per MLP_0 kernel: oracle', rec['layer_grad_norms'], 'reference', rec['layer_dir_derivatives'][:, 1])
    finally:
        ref_standin.unload()


if __name__ == '__main__':
    main()
