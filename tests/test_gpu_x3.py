"""MipNerfModel.obj_precision = 'bf16x3' (round 6; csrc/mlp_f32.hip chunk_mma_x3): the fp32 object branch of a pose-optimisation
step (cfg4) with its forward / backward GEMMs on the bf16 matrix pipe -- every operand a (hi, lo) pair of bf16, a product three
MFMAs (hi.hi + hi.lo + lo.hi), fp32 accumulation -- instead of v_mfma_f32_32x32x2_f32.  The reference's Dense layers are fp32
with HIGHEST-precision matmuls (obbpose_model.py:326-327, internal/math.py:22-24); what this variant has to hold are the gates
the exact branch holds against the ORACLE: the box-pose gradient (5e-2 norm-wise vs the fp32 oracle's autograd, the same
numbers as tests/test_gpu_train.py::test_box_pose_gradients) and the object MLPs' gradients (5e-3)."""
import pytest
import torch

from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from oracle import durf_ref as R
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rel(a, c):
    return float((a.double() - c.double()).norm() / c.double().norm())


def _gin(N, tv, precision):
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
                    'MipNerfModel.no_pose_opt = False\nMipNerfModel.no_yaw_opt = False\n'
                    'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = %g\nMipNerfModel.obj_precision = "%s"\n' % (N, tv, precision))
    return utils.configured(utils.Config)


@pytest.mark.parametrize('K,alpha,tv,B,N', [(2, 4.5, 0.0, 1024, 32), (1, 10.0, 0.01, 1024, 32), (3, 10.0, 0.01, 512, 128)])
def test_pose_and_object_gradients_against_the_oracle(cuda, K, alpha, tv, B, N):
    config = _gin(N, tv, 'bf16x3')
    b = synthetic.make_batch(B, K, seed=77 + K, noise_boxes=0.05)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
    assert model.object_precision() == 'f32' and model.object_x3()
    params = H.oracle_params_from_variables(variables)
    prev_c, prev_d = ob['init'][0:1] + 0.01, db['init'][0:1] + 0.01
    grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, prev_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=tv)
    mcfg = dict(num_samples=N, no_pose_opt=False, no_yaw_opt=False)
    _, _, ostats, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, mcfg, 5e-4, 3.0, alpha, prev_c)
    lay, ts = variables.layout, b['ts']
    got = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6).cpu()
    want = ograds[0]
    assert float(want[ts].abs().max()) > 0
    for k in range(K):
        rp, rr = _rel(got[ts, k, :3], want[ts, k, :3]), _rel(got[ts, k, 3:], want[ts, k, 3:])
        assert rp < 5e-2 and rr < 5e-2, 'object %d: position rel err %g, rotation rel err %g' % (k, rp, rr)
    og = torch.cat([x.reshape(-1) for x in ograds]).float()
    for k in range(K):
        so = slice(lay.mlp_off['BoxMLP_%d' % k], lay.mlp_off['BoxMLP_%d' % k] + lay.mlp_size[128])
        assert _rel(grad.cpu()[so], og[so]) < 5e-3, 'BoxMLP_%d grad rel err %g' % (k, _rel(grad.cpu()[so], og[so]))
    sl = slice(lay.mlp_off['MLP_0'], lay.mlp_off['MLP_0'] + lay.mlp_size[256])
    assert _rel(grad.cpu()[sl], og[sl]) < 5e-2


def test_split_operand_kernels_against_the_exact_ones(cuda):
    """the same step through obj_precision = 'f32' (exact fp32 MFMA) and 'bf16x3': rendered colours, the loss, the object MLPs'
    gradients and the pose gradient differ by what 16-17 significant bits per operand allow -- and they DO differ (the variant
    ran), while everything outside the object branch's GEMMs is the same code"""
    B, N, K, alpha = 768, 64, 2, 10.0
    out = {}
    for prec in ('f32', 'bf16x3'):
        config = _gin(N, 0.01, prec)
        b = synthetic.make_batch(B, K, seed=90, noise_boxes=0.05)
        db = H.device_batch(b, cuda)
        model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
        grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, db['init'][0:1] + 0.01)
        torch.cuda.synchronize()
        out[prec] = (grad.clone(), [r[0].clone() for r in raw['ret']], variables.layout, b['ts'])
    (g0, rgb0, lay, ts), (g1, rgb1, _, _) = out['f32'], out['bf16x3']
    for a, c in zip(rgb0, rgb1):
        assert float((a - c).abs().max()) < 2e-4
    so = slice(lay.mlp_off['BoxMLP_0'], lay.mlp_off['BoxMLP_0'] + K * lay.mlp_size[128])
    r = _rel(g1[so], g0[so])
    assert 0.0 < r < 1e-3, r
    pg0, pg1 = g0[lay.box[0]:lay.box[1]].view(lay.T, K, 6)[ts], g1[lay.box[0]:lay.box[1]].view(lay.T, K, 6)[ts]
    assert _rel(pg1, pg0) < 2e-2, _rel(pg1, pg0)


def test_one_call_step_is_bit_identical_to_the_python_issued_launches(cuda):
    """DURF_TRAIN_OBJ_X3 through durf_train_step == train_step with obj_precision = 'bf16x3', over three steps with the poses moving"""
    B, N, K = 1024, 32, 3
    res = []
    for fn in (train_boxpose.train_step, train_boxpose.train_step_one_call):
        config = _gin(N, 0.01, 'bf16x3')
        b = synthetic.make_batch(B, K, seed=953, noise_boxes=0.2, redraw_noisy_multi_hit=True)
        db = H.device_batch(b, cuda)
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        state = train_boxpose.create_train_state(variables)
        rng = 11
        for _ in range(3):
            state, stats, rng, pose = fn(model, config, rng, state, db, 5e-4, 0.7, 6.5, db['init'][0:1] + 0.01)
        torch.cuda.synchronize()
        res.append((state.variables.flat.clone(), state.m.clone(), float(stats.loss), pose.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][2] == res[0][2], 'the loss (finite: the batch has no ray that hits two boxes)' 
    assert torch.equal(res[0][3], res[1][3])


def test_the_split_operand_step_is_deterministic(cuda):
    """two runs of the same step give the same bits (the first build of chunk_mma_x3 did not: hipcc copied the results of its
    inline-asm LDS reads before they had arrived)"""
    outs = []
    for _ in range(3):
        config = _gin(64, 0.01, 'bf16x3')
        b = synthetic.make_batch(1024, 3, seed=31, noise_boxes=0.5, redraw_noisy_multi_hit=True)
        db = H.device_batch(b, cuda)
        model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
        grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 3.3, db['init'][0:1] + 0.01)
        torch.cuda.synchronize()
        outs.append(grad.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
