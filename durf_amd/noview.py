"""MipNerfModel.use_viewdirs = False (obbpose_model.py:47,221-232; MLP.__call__ :336-352 with condition=None).

Without a condition the reference's MLP has NO bottleneck and NO view layer: its parameter tree is Dense_0..7 (trunk),
Dense_8 (density head) and Dense_9 = the rgb head straight off the trunk, [256, 3].  (With boxes and dynamics=True the
reference itself cannot run this knob -- `viewdirs_enc` is undefined in the object loop, obbpose_model.py:192-199 -- so it is
the static model's knob: no boxes, or dynamics=False, and that is what is supported here; the object MLPs keep their 12
Dense layers either way.)

The fused kernels are built for the 12-Dense topology, so the 10-Dense network is evaluated THROUGH them by an embedding
that is exact in real arithmetic:
    Dense_9'  (bottleneck, linear)  columns 0..2 = the rgb head's kernel and bias, every other column zero
    Dense_10' (view layer, ReLU)    unit j <- +bottleneck_j, unit 3 + j <- -bottleneck_j   (j = 0..2; view rows zero)
    Dense_11' (rgb head)            rgb_j = unit_j - unit_{3+j}                            (relu(v) - relu(-v) = v)
and the gradient of the real parameters is read back from the same positions (every embedded parameter is a copy of a real
one; the constant +-1 entries' gradients are dropped).  What it costs, stated: the FLOPs of the view branch the network does
not have (+ 21 %), and on the bf16 kernels ONE extra rounding -- the bottleneck is an MFMA operand, so raw_rgb is the rgb
head's output rounded to bf16 (2^-9 relative; the +-1 layers behind it are exact); the exact-fp32 kernels
(mlp_precision = 'f32') carry no error at all.  Where a head output is exactly zero both units are inactive and that
sample's rgb gradient is dropped (a measure-zero set).  tests/test_noview.py (the algebra, on the CPU),
tests/test_gpu_noview.py (against the oracle and the reference's own model outputs)."""
import torch

from . import obbpose_model as om
from . import ops


def _maps(real, dev):
    """index of every real parameter in the 12-Dense flat buffer, and that buffer's constant part"""
    T, K = real.T, real.K
    full = om.ParamLayout(T, K, True)
    W = om.W_BKGD
    shapes = full.layer_shapes('MLP_0')
    off = [full.mlp_off['MLP_0']]
    for fi, fo in shapes:
        off.append(off[-1] + fi * fo + fo)
    n_box = real.box[1]
    head = sum(fi * fo + fo for fi, fo in shapes[:9])                     # Dense_0..8: the same bytes in both trees
    assert real.mlp_off['MLP_0'] == full.mlp_off['MLP_0'] == n_box
    idx = [torch.arange(n_box + head)]
    o9 = off[9]
    idx.append((o9 + torch.arange(W)[:, None] * W + torch.arange(3)[None, :]).reshape(-1))     # kernel [W,3] -> columns 0..2
    idx.append(o9 + W * W + torch.arange(3))                                                    # bias
    if K:                                                                                       # the object MLPs: moved, not changed
        idx.append(full.mlp_off['BoxMLP_0'] + torch.arange(K * real.mlp_size[om.W_OBJ]))
    idx = torch.cat(idx)
    assert idx.numel() == real.total
    const = torch.zeros(full.total)
    o10, o11 = off[10], off[11]
    for j in range(3):
        const[o10 + j * 128 + j] = 1.0                  # Dense_10' kernel [W + 27, 128]: unit j <- +bottleneck_j
        const[o10 + j * 128 + 3 + j] = -1.0             #                                  unit 3 + j <- -bottleneck_j
        const[o11 + j * 3 + j] = 1.0                    # Dense_11' kernel [128, 3]: rgb_j = unit_j - unit_{3+j}
        const[o11 + (3 + j) * 3 + j] = -1.0
    return full, idx.to(dev), const.to(dev)


def embed(variables):
    """Variables of the 10-Dense tree -> Variables of the 12-Dense tree the kernels evaluate (cached on `variables`; refreshed
    when the parameters changed: by torch, `_version`, or by a library call, ops.param_generation)"""
    lay = variables.layout
    if lay.use_viewdirs:
        return variables
    c = getattr(variables, '_noview', None)
    if c is None:
        full, idx, const = _maps(lay, variables.flat.device)
        c = variables._noview = dict(full=om.Variables(const, full), idx=idx, key=None)
        c['full']._noview_of = True          # (MipNerfModel._kernel_variables: a use_viewdirs=False model may be handed this tree)
    key = (variables.flat._version, ops.param_generation(variables.flat))
    if c['key'] != key:
        c['full'].flat.index_copy_(0, c['idx'], variables.flat)
        c['key'] = key
    return c['full']


def gather_grad(grad_full, variables):
    """d(loss)/d(real parameters) from the gradient of the embedded ones"""
    return grad_full.index_select(0, variables._noview['idx'])
