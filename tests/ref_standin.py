"""numpy-backed stand-ins for `jax`, `flax`, `gin` and `absl` under which the reference's OWN modules run unmodified.

Build container only (/root/reference is not on the GPU box).  The reference is JAX + flax and neither is installable
here, but what its hot path uses of them is small: array arithmetic (`jax.numpy` -> numpy, float64), `lax.stop_gradient`,
`jax.linearize` (central differences), `jax.random.uniform` (replayed from a queue the test fills), and of flax.linen the
module mechanics -- dataclass fields, `@nn.compact`, `self.param`, children named `<Class>_<n>` in order of construction,
`Dense`, `model.apply({'params': tree}, ...)`.  `load()` registers the stand-ins, imports `internal.{math,mip,mip360,
box_helpers,utils,obbpose_model}` from /root/reference and returns them (`train=True`: also /root/reference/train_boxpose.py,
whose `train_step` then runs with `jax.value_and_grad` handing back a supplied gradient tree and `lax.pmean` the identity);
`unload()` restores `sys.modules`.

A stand-in pins nothing about JAX's or XLA's arithmetic.  What it does pin is the reference's source text: formulas,
argument order, axes, masks, the order of the level loop and of the PRNG draws are executed as written, not restated.
No reference text is copied: the modules are imported from where they lie.
"""
import dataclasses
import importlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference'
_NAMES = ('jax', 'jax.numpy', 'jax.lax', 'jax.random', 'jax.scipy', 'jax.nn', 'jax.tree_util', 'flax', 'flax.linen', 'flax.nn', 'flax.struct',
          'flax.optim', 'gin', 'gin.config', 'absl', 'absl.flags')


def available():
    return os.path.isdir(os.path.join(REF, 'internal'))


class Uniform:
    """jax.random.uniform(key, shape, minval=0, maxval=1): replays queued U[0,1) arrays, in the order the reference draws"""
    queue = []

    @classmethod
    def uniform(cls, key, shape, dtype=None, minval=0.0, maxval=1.0):
        u = cls.queue.pop(0)
        assert list(u.shape) == list(shape), (u.shape, shape)
        return u * (maxval - minval) + minval


class _JArr(np.ndarray):
    """jnp arrays are immutable: `u += x` rebinds u to a NEW (broadcast) array where numpy would write in place"""
    __iadd__ = lambda self, o: self + o
    __isub__ = lambda self, o: self - o
    __imul__ = lambda self, o: self * o
    __itruediv__ = lambda self, o: self / o


class StopGrad:
    """lax.stop_gradient under finite differences: 'record' stores what every call sees at the base point, 'replay' hands
    those values back in call order at a perturbed point -- so a central difference of the reference's loss treats them as
    the constants the derivative treats them as.  None: identity."""
    mode, tape, pos = None, [], 0

    @classmethod
    def stop_gradient(cls, x):
        if cls.mode == 'record':
            cls.tape.append(np.array(x, copy=True))
        elif cls.mode == 'replay':
            x = cls.tape[cls.pos]
            cls.pos += 1
        return x

    @classmethod
    def start(cls, mode):
        cls.mode, cls.pos = mode, 0
        if mode == 'record':
            cls.tape = []


class Devices:
    count = 1


class Hooks:
    """jax.value_and_grad(loss_fn, has_aux=True)(variables): the stand-in evaluates loss_fn, keeps the closure (so a test can
    evaluate the reference's own loss at other points) and returns the gradient tree `grad_provider(variables)` supplies
    (the oracle's autograd result) -- the reference's post-processing then runs on it as written."""
    loss_fn = None
    grad_provider = None


def tree_leaves(tree):
    if isinstance(tree, dict):
        return [x for k in sorted(tree) for x in tree_leaves(tree[k])]
    return [tree]


def tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


class _Any:
    """whatever the train script imports beside the hot path (tensorboard, checkpoints, datasets, plotting)"""
    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return self


class _AnyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Any()


_DUMMIES = ('cv2', 'natsort', 'absl.app', 'flax.core', 'flax.metrics', 'flax.metrics.tensorboard', 'flax.training', 'flax.training.checkpoints',
            'matplotlib', 'matplotlib.pyplot', 'internal.obbpose_dataset', 'internal.c2f_obb_dataset', 'internal.vis')


# ---------------------------------------------------------------------------------------------------------------------
# flax.linen: the module mechanics the reference's model uses
# ---------------------------------------------------------------------------------------------------------------------
_active = []          # modules whose __call__ is executing (innermost last): the parent of whatever is constructed now


def _wrap_call(fn):
    def call(self, *a, **k):
        _active.append(self)
        self._counts = {}
        try:
            return fn(self, *a, **k)
        finally:
            _active.pop()
    call.__wrapped__ = fn
    return call


class Module:
    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        if '__call__' in cls.__dict__:
            cls.__call__ = _wrap_call(cls.__dict__['__call__'])
        dataclasses.dataclass(cls, eq=False, repr=False)          # flax turns every Module subclass into a dataclass

    def __post_init__(self):
        self._counts = {}
        self._params = None
        if _active:                                               # constructed inside a parent's compact __call__
            parent = _active[-1]
            n = parent._counts.get(type(self).__name__, 0)
            parent._counts[type(self).__name__] = n + 1
            self._params = parent._params.setdefault('%s_%d' % (type(self).__name__, n), {})

    def param(self, name, init_fn, *init_args):
        if name not in self._params:                              # model.init: the initializer's value
            self._params[name] = init_fn(None, *init_args)
        return self._params[name]

    def apply(self, variables, *args, rngs=None, **kwargs):
        self._params = variables['params']
        return self(*args, **kwargs)


class Dense(Module):
    features: int = 0
    kernel_init: object = None
    use_bias: bool = True

    def __call__(self, x):
        kernel = self.param('kernel', lambda rng: (_ for _ in ()).throw(KeyError('no kernel given')))
        y = np.matmul(x, kernel)
        return y + self.param('bias', lambda rng: np.zeros(self.features)) if self.use_bias else y


def _linen():
    nn = types.ModuleType('flax.linen')
    nn.Module, nn.Dense = Module, Dense
    nn.compact = lambda f: f
    nn.relu = lambda x: np.maximum(x, 0.0)
    nn.sigmoid = lambda x: 1.0 / (1.0 + np.exp(-x))
    nn.softplus = lambda x: np.logaddexp(x, 0.0)
    return nn


def _install():
    jnp = types.ModuleType('jax.numpy')
    for name in dir(np):
        if not name.startswith('_'):
            setattr(jnp, name, getattr(np, name))
    jnp.ndarray = np.ndarray
    _ax = lambda a: tuple(a) if isinstance(a, list) else a            # jnp reductions accept a list of axes
    jnp.mean = lambda x, axis=None, **k: np.mean(x, axis=_ax(axis), **k)
    jnp.sum = lambda x, axis=None, **k: np.sum(x, axis=_ax(axis), **k)
    jnp.arange = lambda *a, **k: np.arange(*a, **k).view(_JArr)
    jnp.matmul = lambda a, b, precision=None: np.matmul(a, b)
    jnp.linalg = np.linalg
    jnp.float32 = np.float64                       # the check runs in float64 on both sides
    jnp.array = lambda x, dtype=None: np.array(x, dtype=np.float64 if dtype in (None, np.float64) else dtype)
    lax = types.ModuleType('jax.lax')
    lax.stop_gradient = StopGrad.stop_gradient
    lax.pmean = lambda x, axis_name=None: x
    lax.Precision = types.SimpleNamespace(HIGHEST=None)
    random = types.ModuleType('jax.random')
    random.uniform = Uniform.uniform
    random.normal = lambda key, shape, dtype=None: np.zeros(shape)
    random.randint = lambda key, shape, lo, hi: np.zeros(shape)          # randint(0, 1) == 0 (mip.py:324)
    random.split = lambda key, num=2: tuple(key for _ in range(num))
    random.PRNGKey = lambda seed: seed
    jax = types.ModuleType('jax')

    def linearize(f, x):
        def jvp(t, h=1e-6):
            return (f(x + h * t) - f(x - h * t)) / (2 * h)
        return f(x), jvp
    jax.linearize = linearize
    def vmap(f, in_axes=0, out_axes=0):
        if in_axes == 0 and out_axes == 0:         # the hot path only maps functions that are vectorized already
            return f
        return lambda z: np.stack([f(np.take(z, i, axis=in_axes)) for i in range(z.shape[in_axes])], axis=out_axes)
    jax.vmap = vmap
    import functools
    tu = types.ModuleType('jax.tree_util')
    tu.tree_map = tree_map
    tu.tree_leaves = tree_leaves
    tu.tree_reduce = lambda fn, tree, initializer=0: functools.reduce(fn, tree_leaves(tree), initializer)
    jax.tree_util, jax.tree_map = tu, tree_map

    def value_and_grad(f, has_aux=False):
        def run(x):
            Hooks.loss_fn = f
            out = f(x)
            grad = Hooks.grad_provider(x) if Hooks.grad_provider else tree_map(np.zeros_like, x)
            return out, grad
        return run
    jax.value_and_grad = value_and_grad
    jax.config = _Any()
    # one host; `Devices.count` devices (tests set it to exercise render_image's padding to a multiple of the device count)
    jax.host_id = lambda: 0
    jax.host_count = lambda: 1
    jax.device_count = lambda: Devices.count
    jax.local_device_count = lambda: Devices.count
    jnn = types.ModuleType('jax.nn')
    jnn.initializers = types.SimpleNamespace(glorot_uniform=lambda: None)
    jsp = types.ModuleType('jax.scipy')
    import scipy.signal
    jsp.signal = types.SimpleNamespace(convolve2d=lambda z, f, mode='full', precision=None: scipy.signal.convolve2d(z, f, mode=mode))
    jax.numpy, jax.lax, jax.random, jax.nn, jax.scipy = jnp, lax, random, jnn, jsp
    sys.modules['jax.tree_util'] = tu

    nn = _linen()
    flax = types.ModuleType('flax')
    fnn = types.ModuleType('flax.nn')
    fnn.relu, fnn.sigmoid, fnn.softplus = nn.relu, nn.sigmoid, nn.softplus
    struct = types.ModuleType('flax.struct')
    struct.dataclass = dataclasses.dataclass
    optim = types.ModuleType('flax.optim')
    optim.Optimizer = object
    flax.linen, flax.nn, flax.struct, flax.optim = nn, fnn, struct, optim

    gin = types.ModuleType('gin')

    def configurable(*a, **k):                     # @gin.configurable, @gin.configurable(), @gin.configurable('name')
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    gin.configurable = configurable
    gin.add_config_file_search_path = lambda p: None
    gcfg = types.ModuleType('gin.config')
    gcfg.external_configurable = lambda f, module=None, name=None: f
    gin.config = gcfg

    absl = types.ModuleType('absl')
    flags = types.ModuleType('absl.flags')
    def _flag_attr(name):                          # DEFINE_* are only called from functions
        if name.startswith('DEFINE_'):
            return lambda *a, **k: None
        raise AttributeError(name)
    flags.__getattr__ = _flag_attr
    flags.FLAGS = types.SimpleNamespace()
    absl.flags = flags

    sys.modules.update({'jax': jax, 'jax.numpy': jnp, 'jax.lax': lax, 'jax.random': random, 'jax.scipy': jsp, 'jax.nn': jnn,
                        'flax': flax, 'flax.linen': nn, 'flax.nn': fnn, 'flax.struct': struct, 'flax.optim': optim,
                        'gin': gin, 'gin.config': gcfg, 'absl': absl, 'absl.flags': flags})


_saved = None


def _drop_internal():
    for k in [k for k in sys.modules if k == 'internal' or k.startswith('internal.')]:
        del sys.modules[k]


def load(modules=('math', 'mip', 'mip360', 'box_helpers', 'utils', 'obbpose_model'), train=False, dataset=False):
    """-> namespace of the reference's modules, imported unmodified under the stand-ins"""
    global _saved
    assert available(), 'reference tree not present'
    _saved = {k: sys.modules.get(k) for k in _NAMES}
    _install()
    sys.path.insert(0, REF)
    _drop_internal()
    ns = types.SimpleNamespace(**{n: importlib.import_module('internal.' + n) for n in modules})
    if dataset:    # internal/obbpose_dataset.py for real (cv2 / natsort: dummies; its file readers are not called)
        for name in ('cv2', 'natsort'):
            sys.modules[name] = _AnyModule(name)
        ns.obbpose_dataset = importlib.import_module('internal.obbpose_dataset')
    if train:      # /root/reference/train_boxpose.py: train_step (:49-321); everything it imports beside the hot path is a dummy
        for name in _DUMMIES:
            sys.modules[name] = _AnyModule(name)
        sys.modules['absl'].app = sys.modules['absl.app']
        sys.modules['flax'].core = sys.modules['flax.core']
        spec = importlib.util.spec_from_file_location('ref_train_boxpose', os.path.join(REF, 'train_boxpose.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ns.train_boxpose = mod
    return ns


def unload():
    global _saved
    if REF in sys.path:
        sys.path.remove(REF)
    _drop_internal()
    for name in _DUMMIES:
        sys.modules.pop(name, None)
    StopGrad.start(None)
    Hooks.loss_fn = Hooks.grad_provider = None
    for k, v in (_saved or {}).items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    _saved = None


# ---------------------------------------------------------------------------------------------------------------------
# running the reference's model
# ---------------------------------------------------------------------------------------------------------------------
def flax_tree(params):
    """the oracle's params dict (torch, oracle/durf_ref.py) -> the flax variable tree the reference's model.apply takes"""
    tree = {'box_centers': params['box_centers'].detach().double().numpy()}
    for name, layers in params.items():
        if name != 'box_centers':
            tree[name] = {'Dense_%d' % i: {'kernel': k.detach().double().numpy(), 'bias': b.detach().double().numpy()}
                          for i, (k, b) in enumerate(layers)}
    return {'params': tree}


def run_model(ref, model_kwargs, params, rays, ext, ts, randomized, white_bkgd, alpha, uniforms=()):
    """MipNerfModel(**model_kwargs).apply(...) of the reference (obbpose_model.py:68-261) on numpy float64 inputs.
    rays: the oracle's Rays namedtuple (torch); uniforms: the U[0,1) arrays its PRNG draws are replaced with, in order."""
    f = lambda t: t.detach().double().numpy()
    r = ref.utils.BoxRays(*[f(getattr(rays, n)) for n in ref.utils.BoxRays._fields])
    tree = flax_tree(params)
    model = ref.obbpose_model.MipNerfModel(**model_kwargs)
    Uniform.queue = [np.asarray(u, dtype=np.float64) for u in uniforms]
    out = model.apply(tree, 0, r, tree['params']['box_centers'], f(ext), np.array([int(ts)]), randomized=randomized,
                      rand_bkgd=False, white_bkgd=white_bkgd, alpha=alpha)
    assert not Uniform.queue, 'the reference drew fewer uniforms than expected'
    return out
