"""Where the bf16 noise of the box-pose gradient comes from: cfg4's per-rank shape with the box-hit rays in bf16 / fp32
(MipNerfModel.obj_precision: the object MLPs AND the background MLP's one evaluation per hit ray) against the exact-fp32
instrument, true fp32 weights (NOT rounded to bf16), several seeds.  Round-3 finding (profiles/r03_pose_grad_ablation.txt):
fp32 object MLPs alone leave 10-25 %, an fp32 background evaluation of the hit rays alone 6-18 %, both together 0."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils

cuda = torch.device('cuda:0')


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def run(B, K, N, alpha, noise_boxes, seed, rnd, round_weights=False):
    out, ms = {}, {}
    g = torch.Generator().manual_seed(12)
    noise = dict(t_rand=torch.rand(B, N + 1, generator=g).to(cuda), u_rand=torch.rand(B, N + 1, generator=g).to(cuda))
    for tag, prec, oprec in (('bf16', 'bf16', 'bf16'), ('mixed', 'bf16', 'f32'), ('f32', 'f32', 'auto')):
        utils.clear_gin()
        utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = False\n'
                        'MipNerfModel.no_yaw_opt = False\nMipNerfModel.mlp_precision = %r\nMipNerfModel.obj_precision = %r\n'
                        'Config.randomized = %s\nConfig.rand_bkgd = False\nConfig.grad_max_norm = 1.0\n'
                        'Config.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n' % (N, prec, oprec, rnd))
        config = utils.configured(utils.Config)
        b = synthetic.make_batch(B, K, seed=seed, far=40.0, noise_boxes=noise_boxes, redraw_noisy_multi_hit=True)
        db = synthetic.device_batch(b, cuda)
        model, variables = obbpose_model.construct_mipnerf(3, db, device=cuda)
        lay = variables.layout
        if round_weights:
            w = variables.flat[lay.box[1]:]
            w.copy_(w.to(torch.bfloat16).float())
        f = lambda: train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, alpha, db['init'][0:1],
                                                noise=noise if rnd else None)
        grad, raw, _ = f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ms[tag] = (time.perf_counter() - t0) / 3 * 1e3
        out[tag] = grad[lay.box[0]:lay.box[1]].view(lay.T, K, 6)[b['ts']].clone()
        o = lay.mlp_off['BoxMLP_0']
        out[tag + '_objw'] = grad[o:o + K * lay.mlp_size[128]].clone()
    f = out['f32']
    print('B=%4d K=%d N=%3d alpha=%4.1f noise=%.2f seed=%d rnd=%s wround=%s hit=%.3f |f32 pos| %.2e |f32 rot| %.2e' % (
        B, K, N, alpha, noise_boxes, seed, rnd, round_weights, b['hit_fraction'], float(f[:, :3].norm()), float(f[:, 3:].norm())))
    for tag in ('bf16', 'mixed'):
        a = out[tag]
        print('    %-6s pos %.4f rot %.4f  objW %.4f   loss_and_grad %.2f ms (f32: %.1f ms)' % (
            tag, rel(a[:, :3], f[:, :3]), rel(a[:, 3:], f[:, 3:]), rel(out[tag + '_objw'], out['f32_objw']), ms[tag], ms['f32']))


if __name__ == '__main__':
    for args in [(1024, 3, 128, 3.3, 0.5, 93, True), (1024, 3, 128, 3.3, 0.5, 93, True, True), (1024, 3, 128, 3.3, 0.5, 94, True),
                 (1024, 3, 128, 3.3, 0.5, 95, True), (1024, 3, 128, 10.0, 0.5, 93, True), (1024, 3, 128, 3.3, 0.05, 93, True),
                 (1024, 3, 32, 3.3, 0.5, 93, False), (4096, 3, 128, 3.3, 0.5, 93, True)]:
        run(*args)
