"""cfg3 step time through the Python-issued launches (train_step) and through the single C entry point (durf_train_step):
    python tools/time_train_call.py [rays [K [num_samples]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 128
utils.clear_gin()
utils.parse_gin(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'waymo.gin')).read())
utils.parse_gin('MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\nConfig.rand_bkgd = False\nMipNerfModel.num_samples = %d\n' % NS)
config = utils.configured(utils.Config)
b = synthetic.make_batch(B, K, seed=1, far=40.0)
db = synthetic.device_batch(b, dev)
MODES = (('train_step (Python-issued launches)', train_boxpose.train_step), ('durf_train_step (one C call)', train_boxpose.train_step_one_call))
only = os.environ.get('ONLY')            # ONLY=python / ONLY=c: one path (for a kernel trace of it)
for name, fn in ((MODES[0],) if only == 'python' else (MODES[1],) if only == 'c' else MODES * 2):
    model, variables = obbpose_model.construct_mipnerf(0, db, device=dev)
    state = train_boxpose.create_train_state(variables)
    rng = 0
    for _ in range(20):
        state, stats, rng, pose = fn(model, config, rng, state, db, 5e-4, 3.0, 10.0, db['init'][0:1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        state, stats, rng, pose = fn(model, config, rng, state, db, 5e-4, 3.0, 10.0, db['init'][0:1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print('%-40s K=%d N=%d %d rays: %.3f ms/step, %.1f k rays/s, loss %.5f' % (name, K, NS, B, dt * 1e3, B / dt / 1e3, float(stats.loss)))
