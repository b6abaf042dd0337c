#!/usr/bin/env python3
"""Headline benchmark: train rays/sec of the durf ray pipeline on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

A "step" is one full training step (forward 2 levels, losses, backward, one gradient
all-reduce, clip + Adam) of MipNerfModel over one synthetic random-pose ray batch that is
already resident in HBM.  Workload = BASELINE.json configs[1] (SURVEY.md 8d "cfg2"):
CARLA-like dynamic scene, K=1 moving OBB, 128 samples/ray x 2 levels, 8x256 background MLP +
8x128 object MLP, bf16 MFMA GEMMs (fp32 accumulate, fp32 everywhere else), 4096 rays per GPU
(weak scaling: the global batch is 4096*N rays).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS_PER_GPU = 4096
N_LEVELS = 2
N_SAMPLES = 128
FAR = 200.0            # configs/carla_dyn.gin:13
MAC_BKGD, MAC_OBJ = 591872, 167552
PEAK_BF16 = 2.5e15     # dense MFMA bf16 peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12      # HBM3E spec peak (6.29e12 measured-achievable), MI355X_MICROARCH.md


def gin_text():
    return ('MipNerfModel.num_samples = %d\nMipNerfModel.density_noise = 0.0\n'
            'MipNerfModel.no_pose_opt = True\nMipNerfModel.no_yaw_opt = True\n'
            'Config.randomized = True\nConfig.rand_bkgd = False\nConfig.white_bkgd = False\n'
            'Config.grad_max_norm = 1.0\nConfig.grad_max_val = 0.1\nConfig.tv_loss_mult = 0.0\n'
            'Config.depth_loss_mult = 0.0001\nConfig.near_loss_mult = 0.01\nConfig.empty_loss_mult = 1.0\n'
            'Config.sky_loss_mult = 1.0\nConfig.box_loss_mult = 0\nConfig.far = %g\n' % (N_SAMPLES, FAR))


def cpu_baseline(batch_np, seconds_budget=15.0, K_OBJ=1):
    """The oracle's train_step (fp32 torch-CPU restatement of the reference step) timed on the
    host cores on a bounded sample of the same workload.  A reported baseline, not a target."""
    import numpy as np
    from oracle import durf_ref as R
    from tests import helpers as H
    Bc = 256
    sub = dict(batch_np)
    sub['rays'] = {k: v[:Bc] for k, v in batch_np['rays'].items()}
    for k in ('pixels', 'depth', 'sky'):
        sub[k] = batch_np[k][:Bc]
    ob = H.oracle_batch(sub)
    # torch-CPU scales poorly past a few tens of threads on these small GEMMs (256 threads
    # were 50x slower than 16 on the GPU box): use at most 16 and report what was used.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    params = R.init_params(0, ob['init'], K_OBJ)
    cfg = dict(R.CONFIG_DEFAULTS, randomized=False)
    st = R.new_opt_state(params)
    prev = ob['init'][0:1]
    mcfg = dict(num_samples=N_SAMPLES)
    R.train_step(params, st, ob, cfg, mcfg, 5e-4, 3.0, 10.0, prev)      # warm-up
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < seconds_budget and n < 20):
        params, st, _, _ = R.train_step(params, st, ob, cfg, mcfg, 5e-4, 3.0, 10.0, prev)
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=Bc / dt, unit='rays/s', cores=cores, kind='port',
                sample='%d steps of the oracle train_step (fp32 torch-CPU restatement, not JAX) on %d rays '
                       'of the same workload (N=%d, K=%d), %.2f s/step' % (n, Bc, N_SAMPLES, K_OBJ, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rays', type=int, default=RAYS_PER_GPU, help='rays per GPU')
    ap.add_argument('--objects', type=int, default=1, help='dynamic boxes K (default: cfg2; 3 = cfg3, 8 = cfg5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-ops', action='store_true', help='print the per-op time table to stderr')
    args = ap.parse_args()

    from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
    from tests import helpers as H
    import torch.distributed as dist

    rank, world, local = train_boxpose.init_distributed()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    utils.clear_gin()
    utils.parse_gin(gin_text())
    config = utils.configured(utils.Config)
    B = args.rays
    K_OBJ = args.objects
    # Weak scaling: every rank trains on the same synthetic B-ray batch (same boxes / poses / timestep, as one
    # 'timestep' batch of the reference has) with rank-dependent stratified-sampling noise, so the per-GPU work --
    # including the fraction of rays that hit a box -- does not change with the number of GPUs.  The gradient
    # all-reduce and the stats all-reduce run exactly as with distinct shards.
    batch_np = synthetic.make_batch(B, K_OBJ, far=FAR, seed=synthetic.SEED)
    full = H.device_batch(batch_np, dev)
    batch = full
    model, variables = obbpose_model.construct_mipnerf(0, full, device=dev)
    state = train_boxpose.create_train_state(variables)
    prev = full['init'][0:1]
    lr, eps, alpha = 5e-4, 3.0, 10.0

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rng = 1000 * rank                                  # stratified-sampling noise differs per rank
    for _ in range(args.warmup):
        state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, lr, eps, alpha, prev)
    sync()
    # live HIP-event timers over the timed region: only the three kernels the roofline reports, unless the full
    # per-op table is asked for (every timed op costs two event records)
    ops.TIMED_NAMES = None if args.profile_ops else {'mlp_fwd_256_train', 'mlp_bwd_256', 'mlp_dw_256'}
    ops.TIMERS = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state, stats, rng, _ = train_boxpose.train_step(model, config, rng, state, batch, lr, eps, alpha, prev)
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    totals = ops.timer_totals()
    ops.TIMERS = None

    if rank == 0:
        rows = B * N_SAMPLES
        hit = float(batch_np['hit_fraction'])
        # Roofline of the three background-MLP kernels (96 % of the step).  Algorithmic work per
        # launch (one level): FLOPs = 2 * 591 872 MAC * samples; bytes = what the
        # data flow of DESIGN.md section 3 has to move: bf16 activation stash 9*W+128 features
        # (4864 B/sample), ReLU bit-mask 288 B/sample, encoding 128 B, raw/draw 16 B, dz_out 32 B;
        # the weight-gradient GEMMs read dz + stash + encodings = 332 KB per 32-sample tile.
        stash_b, mask_b = 4864.0, 288.0
        work = {
            'mlp_fwd_256_train': (2.0 * MAC_BKGD * rows, rows * (128 + stash_b + mask_b + 16)),
            'mlp_bwd_256': (2.0 * MAC_BKGD * rows, rows * (16 + mask_b + stash_b + 32)),
            # ONE launch covers the samples of both levels
            'mlp_dw_256': (N_LEVELS * 2.0 * MAC_BKGD * rows, N_LEVELS * rows / 32.0 * 332 * 1024),
        }
        per = {k: totals[k][1] / totals[k][0] for k in work if k in totals}
        info = {}
        for k, t in per.items():
            fl, by = work[k]
            t_mfma, t_hbm = fl / PEAK_BF16, by / PEAK_HBM
            bound = 'hbm' if t_hbm > t_mfma else 'mfma'
            info[k] = dict(ms=t * 1e3, tflops=fl / t / 1e12, tbps=by / t / 1e12, bound=bound,
                           frac=(t_hbm if bound == 'hbm' else t_mfma) / t)
        dom = max(per, key=lambda k: per[k]) if per else None
        roof = None
        if dom:
            d = info[dom]
            if d['bound'] == 'hbm':
                roof = dict(bound='hbm', kernel=dom, achieved=d['tbps'] * 1e3, peak=PEAK_HBM / 1e9, unit='GB/s',
                            frac=d['frac'], traffic=None, launch_ms=d['ms'], all=info)
            else:
                roof = dict(bound='mfma', kernel=dom, achieved=d['tflops'], peak=PEAK_BF16 / 1e12, unit='TFLOP/s',
                            frac=d['frac'], traffic=None, launch_ms=d['ms'], all=info)
            # HBM traffic of the dominant kernel from the committed PMC passes (not measurable live)
            try:
                with open(os.path.join(ROOT, 'profiles', 'r01d_pmc_traffic.json')) as f:
                    tr = json.load(f)
                if B == RAYS_PER_GPU and dom in tr:
                    roof['traffic'] = tr[dom]['total_bytes']
                    roof['traffic_source'] = tr['source']
            except (OSError, ValueError):
                pass
            # end-to-end MFMA rate of the whole step (fwd + bwd-data + dW = 3 x fwd FLOPs, both levels)
            roof['step_mlp_tflops'] = 3 * 2 * 2.0 * (MAC_BKGD + hit * MAC_OBJ) * rows / (dt / args.steps) / 1e12
        if args.profile_ops:
            for k, (n, s) in sorted(totals.items(), key=lambda kv: -kv[1][1]):
                print('%-22s calls %4d  total %8.2f ms  per step %7.3f ms' % (k, n, s * 1e3, s * 1e3 / args.steps),
                      file=sys.stderr)
        cb = None if args.no_cpu_baseline else cpu_baseline(batch_np, K_OBJ=K_OBJ)
        out = dict(metric='train_rays_per_sec', value=B * world * args.steps / dt, unit='rays/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='bf16', data='synthetic',
                   config=dict(workload=('cfg2: CARLA-like dynamic scene, K=1 OBB' if K_OBJ == 1 else 'K=%d OBBs' % K_OBJ) + ', 128 samples/ray x 2 levels, '
                                        '8x256 bkgd MLP + 8x128 object MLP, full train step',
                               rays_per_gpu=B, global_batch=B * world, num_samples=N_SAMPLES, num_levels=2,
                               objects=K_OBJ, far=FAR, hit_fraction=hit, randomized=True,
                               parallelism='dp%d' % world),
                   loss=float(stats.loss), psnr=float(stats.psnr), roofline=roof, cpu_baseline=cb)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
