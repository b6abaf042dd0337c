"""Types, config knobs and a minimal gin reader -- host-side mirror of internal/utils.py.

Names, fields and defaults follow the reference (internal/utils.py:37-144) so that train
scripts and notebooks written against it keep working; gin itself is not available here, so
`parse_gin` reads the `Name.attr = literal` subset the two shipped configs use
(configs/waymo.gin, configs/carla_dyn.gin)."""
import ast
import collections
import dataclasses
from typing import Any

Rays = collections.namedtuple(
    'Rays', ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far', 'delta'))
DynRays = collections.namedtuple(
    'DynRays', ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far', 'time', 'object'))
BoxRays = collections.namedtuple(
    'BoxRays', ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far'))


@dataclasses.dataclass
class Stats:
    """internal/utils.py:42-74 (same field names; tensors/scalars)."""
    loss: Any = 0.0
    obj_losses: Any = 0.0
    losses: Any = 0.0
    d_losses: Any = 0.0
    n_losses: Any = 0.0
    e_losses: Any = 0.0
    s_losses: Any = 0.0
    distr_losses: Any = 0.0
    tv_losses: Any = 0.0
    offsets: Any = 0.0
    offset_x: Any = 0.0
    offset_y: Any = 0.0
    offset_z: Any = 0.0
    offset_yaw: Any = 0.0
    pose: Any = 0.0
    sampling_stats: Any = 0.0
    weights: Any = 0.0
    samples: Any = 0.0
    weight_l2: Any = 0.0
    psnr: Any = 0.0
    psnrs: Any = 0.0
    obj_psnr: Any = 0.0
    grad_norm: Any = 0.0
    grad_abs_max: Any = 0.0
    grad_norm_clipped: Any = 0.0
    # not a reference field: rays of the batch that hit two or more boxes.  The reference sums their object-frame
    # origins (obbpose_model.py:120-122, "assumes that objects do not occlude each other"); this build treats them
    # as out of domain (non-finite on both sides) and, with pose optimisation on, drops the cross-object part of
    # their pose gradient -- so a driver can see how many there were (train_loop warns when pose_opt is on).
    multi_hit_rays: Any = 0


@dataclasses.dataclass
class Config:
    """Configuration flags for everything (internal/utils.py:89-144)."""
    dataset_loader: str = 'multicam'
    batching: str = 'all_images'
    batch_size: int = 4096
    factor: int = 0
    spherify: bool = False
    centering: bool = False
    random_box: bool = False
    random_yaw: bool = False
    box_noise: float = 0.5
    yaw_noise: float = 5.
    render_path: bool = False
    llffhold: int = 8
    timesteps: int = 5
    lr_init: float = 5e-4
    lr_final: float = 5e-6
    lr_delay_steps: int = 2500
    eps_delay_steps: int = 0
    eps_init: float = 3
    eps_final: float = 0.2
    eps_max_steps: int = 1000000
    l2_reg: bool = False
    alpha_init: float = 0.0
    alpha_final: float = 10.0
    alpha_delay_steps: int = 0
    alpha_max_steps: int = 1000000
    psreg_init: float = 10e5
    psreg_final: float = 10e-1
    psreg_delay_steps: int = 5000
    psreg_delay_mult: float = 1.0
    tv_loss_mult: float = 0.0001
    depth_loss_mult: float = 0.0001
    near_loss_mult: float = 0.01
    empty_loss_mult: float = 1.0
    sky_loss_mult: float = 1.0
    c2f_steps: tuple = (5000, 10000, 15000)
    lr_delay_mult: float = 0.01
    grad_max_norm: float = 0.
    grad_max_val: float = 0.
    max_steps: int = 1000000
    save_every: int = 100000
    print_every: int = 100
    gc_every: int = 10000
    test_render_interval: int = 1
    disable_multiscale_loss: bool = False
    randomized: bool = True
    near: float = 2.
    far: float = 6.
    coarse_loss_mult: float = 0.1
    box_loss_mult: int = 0
    weight_decay_mult: float = 0.
    white_bkgd: bool = False
    rand_bkgd: bool = True


# class name -> {attr: value}; filled by parse_gin, consumed by `configured`
_BINDINGS = collections.defaultdict(dict)

_GIN_SYMBOLS = {'@flax.nn.relu': 'relu', '@flax.nn.sigmoid': 'sigmoid', '@flax.nn.softplus': 'softplus'}


def parse_gin(text_or_path, bindings=None):
    """Read `Class.attr = literal` lines (comments with '#').  Returns the binding dict and
    also records it for `configured()`.  Unknown syntax raises ValueError, like gin."""
    text = text_or_path
    if '\n' not in text_or_path and text_or_path.endswith('.gin'):
        with open(text_or_path) as f:
            text = f.read()
    lines = text.splitlines() + list(bindings or [])
    for ln in lines:
        ln = ln.split('#', 1)[0].strip()
        if not ln:
            continue
        if '=' not in ln or '.' not in ln.split('=', 1)[0]:
            raise ValueError('unsupported gin line: %r' % ln)
        lhs, rhs = [s.strip() for s in ln.split('=', 1)]
        cls, attr = lhs.rsplit('.', 1)
        cls = cls.split('.')[-1]
        if rhs in _GIN_SYMBOLS:
            val = _GIN_SYMBOLS[rhs]
        else:
            try:
                val = ast.literal_eval(rhs)
            except (ValueError, SyntaxError):
                raise ValueError('unsupported gin value: %r' % ln)
        _BINDINGS[cls][attr] = val
    return {k: dict(v) for k, v in _BINDINGS.items()}


def clear_gin():
    _BINDINGS.clear()


def configured(cls, **overrides):
    """Instantiate a dataclass with the gin bindings recorded for its name applied."""
    kw = dict(_BINDINGS.get(cls.__name__, {}))
    kw.update(overrides)
    names = {f.name for f in dataclasses.fields(cls)}
    bad = set(kw) - names
    if bad:
        raise ValueError('%s has no configurable attribute(s) %s' % (cls.__name__, sorted(bad)))
    return cls(**kw)


def load_config(gin_files=(), gin_params=()):
    """internal/utils.py:162-165."""
    for f in gin_files:
        parse_gin(f)
    if gin_params:
        parse_gin('\n', bindings=list(gin_params))
    return configured(Config)


def namedtuple_map(fn, tup):
    """internal/utils.py:188-190."""
    return type(tup)(*map(fn, tup))
