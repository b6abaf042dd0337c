"""bf16 production path vs the exact-fp32 instrument: rel. error of the box-pose gradient over a grid of settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils

cuda = torch.device('cuda:0')
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
def run(B, K, N, alpha, noise_boxes, seed, rnd):
    out = {}
    g = torch.Generator().manual_seed(12)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda))
    for prec in ('bf16', 'f32'):
        utils.clear_gin()
        utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = False\n'
                        'MipNerfModel.no_yaw_opt = False\nMipNerfModel.mlp_precision = %r\nConfig.randomized = %s\n'
                        'Config.rand_bkgd = False\nConfig.grad_max_norm = 1.0\nConfig.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % (N, prec, rnd))
        config = utils.configured(utils.Config)
        b = synthetic.make_batch(B, K, seed=seed, far=40.0, noise_boxes=noise_boxes, redraw_noisy_multi_hit=True)
        db = synthetic.device_batch(b, cuda)
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        lay = variables.layout
        w = variables.flat[lay.box[1]:]; w.copy_(w.to(torch.bfloat16).float())
        grad, raw, _ = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, db['init'][0:1], noise=noise if rnd else None)
        out[prec] = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6)[b['ts']].clone()
    a, f = out['bf16'], out['f32']
    print('B=%4d K=%d N=%3d alpha=%4.1f noise=%.2f seed=%d rnd=%s hit=%.3f : pos %.3f rot %.3f   |f32 pos| %.2e' % (
        B, K, N, alpha, noise_boxes, seed, rnd, b['hit_fraction'], rel(a[:, :3], f[:, :3]), rel(a[:, 3:], f[:, 3:]), float(f[:, :3].norm())))
for args in [(1024, 3, 128, 3.3, 0.5, 93, True), (1024, 3, 128, 3.3, 0.5, 93, False), (1024, 3, 32, 3.3, 0.5, 93, True),
             (1024, 3, 128, 10.0, 0.5, 93, True), (1024, 3, 128, 3.3, 0.05, 93, True), (1024, 2, 32, 4.5, 0.05, 79, False),
             (1024, 3, 128, 3.3, 0.5, 94, True), (1024, 3, 128, 3.3, 0.5, 95, True), (4096, 3, 128, 3.3, 0.5, 93, True)]:
    run(*args)
