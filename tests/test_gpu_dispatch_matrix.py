"""Dispatch matrix: every kernel variant the launchers select by SIZE (or by a remaining switch) runs at least once inside a
test that compares its results with the oracle.  Round 4's review found the 512-workgroup weight-gradient plan -- the one the
headline times -- covered by a checksum only, because every oracle-sized case had dropped below the size switch.  The
launchers now log which variant they dispatched (durf_dispatch_seen, include/durf_hip.h DURF_DISPATCH_*); each row below runs an
oracle comparison that lives elsewhere in the suite at a size (or under the forcing switch) that selects the variant, and
asserts that it was the variant that ran.  Reference work item: train_boxpose.py:251-252 (value_and_grad of the whole step)."""
import os

import pytest
import torch

from durf_amd import ops
from tests import test_gpu_fullsize as FS
from tests import test_gpu_train as TR

pytestmark = pytest.mark.gpu


def _env(**kw):
    class _E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kw}
            os.environ.update({k: str(v) for k, v in kw.items()})

        def __exit__(self, *a):
            for k, v in self.old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _E()


# (what must have run, the oracle-compared scenario that selects it)
ROWS = [
    # 256 rays x 32 samples = 32 blocks: 128-sample blocks of 4 waves; objects on the M-split kernels; one-round dW plans
    # 256 rays x 32 samples: the object MLPs' forward / backward ride as items in the background MLP's persistent launches
    # (round 6: k_mlp_fwd / k_mlp_bwd <.., MIX>, 256-sample blocks of 8 waves); one-round dW plans
    ('small step vs the bf16-rounded oracle',
     {'FWD256_8W', 'BWD256_8W', 'FWD_MIX', 'BWD_MIX', 'DW256_256WG', 'DW128_128WG', 'FWD_ENC', 'FWD_TAIL', 'FWD_RAW_FULL'},
     lambda cuda: TR.test_train_step(cuda, 3, 32, 256)),
    # the same with the mix switched off: 128-sample blocks of 4 waves (32 blocks), objects on the M-split kernels of their own
    ('small step with launches of their own for the objects',
     {'FWD256_4W', 'BWD256_4W', 'FWD128_MSPLIT', 'BWD128_MSPLIT'},
     lambda cuda: _unmixed(cuda)),
    # the same oracle comparison with the LARGE batches' choices forced: sample-split object kernels (>= 2048 x 128 rows in
    # production) and the two-round / 256-per-object weight-gradient plans (>= 3072 x 256 rows)
    ('small step under the large-batch variants',
     {'FWD128_SAMPLE', 'BWD128_SAMPLE', 'DW256_512WG', 'DW128_256WG'},
     lambda cuda: _forced(cuda)),
    # 512 rays x 128 samples = 256 blocks: 256-sample blocks of 8 waves, against the plain fp32 oracle's autograd; its fp32
    # leg runs the W = 256 exact-fp32 weight gradients (2 x 2 blocks), its pose case the fp32 object branch (one tile per
    # workgroup) 
    ('reference batch at 128 samples vs the fp32 oracle',
     {'FWD256_8W', 'BWD256_8W', 'F32_DW_B2', 'F32_DW_TILE'},
     lambda cuda: FS.test_gradients_at_the_reference_batch_and_128_samples_against_the_oracle(cuda, 512, 3, True, 3.3)),
    # the metric's own row count through the two-round plan, against float64 products of the untiled operands
    ('weight gradients at 4096 rays x 128 x 2',
     {'DW256_512WG'},
     lambda cuda: FS.test_weight_gradients_of_both_split_plans_against_untiled_matmuls(cuda, 4096, 'DW256_512WG')),
    # bf16 object MLPs WITH the box-pose gradient behind them: k_mlp_bwd<128, POSE> (obj_precision = 'bf16' forced)
    ('box-pose gradient through the bf16 object backward',
     {'BWD_POSE'},
     lambda cuda: _bf16_objects_with_pose_gradient(cuda)),
]


def _bf16_objects_with_pose_gradient(cuda):
    """obj_precision = 'bf16' with pose optimisation on (round 2's production path; 'auto' now moves the hit rays to fp32):
    the object MLPs' gradients 5e-2 and the box-pose gradient against the oracle with bf16-rounded GEMM operands at round
    2's gates (position 0.1, rotation 0.3 norm-wise: a sum that cancels to ~1 % of its terms, tools/pose_grad_ablate.py)"""
    from durf_amd import obbpose_model, synthetic, train_boxpose, utils
    from oracle import durf_ref as R
    from tests import helpers as H
    B, N, K, alpha = 1024, 32, 1, 10.0
    utils.clear_gin()
    utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.obj_precision = "bf16"\n'
                    'MipNerfModel.no_pose_opt = False\nMipNerfModel.no_yaw_opt = False\n'
                    'Config.randomized = False\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                    'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % N)
    config = utils.configured(utils.Config)
    b = synthetic.make_batch(B, K, seed=78, noise_boxes=0.05)
    ob, db = H.oracle_batch(b), H.device_batch(b, cuda)
    model, variables = obbpose_model.construct_mipnerf(5, db, device=cuda)
    assert model.object_precision() == 'bf16'
    params = H.oracle_params_from_variables(variables)
    prev_c, prev_d = ob['init'][0:1] + 0.01, db['init'][0:1] + 0.01
    grad, _, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, prev_d)
    torch.cuda.synchronize()
    ocfg = dict(R.CONFIG_DEFAULTS, randomized=False, tv_loss_mult=0.0)
    mcfg = dict(num_samples=N, no_pose_opt=False, no_yaw_opt=False)
    _, _, _, ograds = R.train_step(params, R.new_opt_state(params), ob, ocfg, mcfg, 5e-4, 3.0, alpha, prev_c,
                                   mlp_hook=R.mlp_apply_bf16)
    lay, ts = variables.layout, b['ts']
    rel = lambda a, c: float((a - c).norm() / c.norm())
    og = torch.cat([x.reshape(-1) for x in ograds])
    so = slice(lay.mlp_off['BoxMLP_0'], lay.mlp_off['BoxMLP_0'] + lay.mlp_size[128])
    assert rel(grad.cpu()[so], og[so]) < 5e-2
    got, want = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6).cpu()[ts], ograds[0][ts]
    assert float(want.abs().max()) > 0
    assert rel(got[:, :3], want[:, :3]) < 0.1 and rel(got[:, 3:], want[:, 3:]) < 0.3, (got, want)


def _forced(cuda):
    with _env(DURF_OBJ_MSPLIT=0, DURF_DW_WGS=512, DURF_DW_WGS_OBJ=256):
        TR.test_train_step(cuda, 3, 32, 256)


def _unmixed(cuda):
    with _env(DURF_OBJ_MIX=0):
        TR.test_train_step(cuda, 3, 32, 256)


@pytest.mark.parametrize('name,variants,scenario', ROWS, ids=[r[0] for r in ROWS])
def test_every_size_selected_variant_runs_under_an_oracle_comparison(cuda, name, variants, scenario):
    assert variants <= set(ops.DISPATCH), variants - set(ops.DISPATCH)
    ops.dispatch_reset()
    scenario(cuda)                      # asserts its own parity
    seen = ops.dispatch_seen()
    assert variants <= seen, '%s: expected %s to run, the launchers dispatched %s' % (name, sorted(variants - seen), sorted(seen))


def test_the_matrix_covers_every_variant_the_launchers_know():
    covered = set().union(*[r[1] for r in ROWS])
    assert covered == set(ops.DISPATCH), 'no oracle-compared scenario for: %s' % sorted(set(ops.DISPATCH) - covered)
