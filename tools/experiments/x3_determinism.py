"""round 6 debugging aid: is a cfg4-shaped step with obj_precision='bf16x3' bit-reproducible?  (python path and the one C call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from durf_amd import obbpose_model, synthetic, train_boxpose, utils

dev = torch.device('cuda', 0)
for prec in ('f32', 'bf16x3'):
    for N, B in ((64, 1024), (128, 1024), (128, 2048)):
        outs = []
        for rep in range(4):
            utils.clear_gin()
            utils.parse_gin('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\nMipNerfModel.no_pose_opt = False\n'
                            'MipNerfModel.no_yaw_opt = False\nConfig.randomized = False\nConfig.tv_loss_mult = 0.01\n'
                            'MipNerfModel.obj_precision = "%s"\n' % (N, prec))
            config = utils.configured(utils.Config)
            b = synthetic.make_batch(B, 3, seed=31, noise_boxes=0.5, redraw_noisy_multi_hit=True)
            db = synthetic.device_batch(b, dev)
            model, variables = obbpose_model.construct_mipnerf(5, db, device=dev)
            grad, raw, pose = train_boxpose.loss_and_grad(model, config, 0, variables, db, 3.0, 3.3, db['init'][0:1] + 0.01)
            torch.cuda.synchronize()
            state = train_boxpose.create_train_state(variables)
            g2, st, ret = train_boxpose.train_step_one_call(model, config, 0, state, db, 5e-4, 3.0, 3.3, db['init'][0:1] + 0.01, update=False)
            torch.cuda.synchronize()
            outs.append((grad.clone(), g2.clone(), [r[0].clone() for r in raw['ret']]))
        lay = variables.layout
        so = slice(lay.mlp_off['BoxMLP_0'], lay.mlp_off['BoxMLP_0'] + 3 * lay.mlp_size[128])
        same_py = all(torch.equal(outs[0][0], o[0]) for o in outs)
        same_c = all(torch.equal(outs[0][1], o[1]) for o in outs)
        same_rgb = all(all(torch.equal(a, c) for a, c in zip(outs[0][2], o[2])) for o in outs)
        d = max(float((outs[0][0][so] - o[0][so]).abs().max()) for o in outs)
        print(prec, 'N', N, 'B', B, 'python path reproducible:', same_py, ' C call:', same_c, ' rgb:', same_rgb, ' py == C:', torch.equal(outs[0][0], outs[0][1]),
              ' max |d grad_obj| over reps %.3g' % d, flush=True)
