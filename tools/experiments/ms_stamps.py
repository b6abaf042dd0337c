"""round 6 probe: where does an (object, tile pair) item of the M-split forward spend its time?

Needs the `stamps` variant (tools/build_variant.sh stamps "-DDURF_MS_STAMPS"; DURF_LIB_PATH=durf_amd/variants/libdurf_stamps.so):
role-0's lane 0 of every item writes the 100 MHz s_memrealtime at the marks of ms_fwd_pair (csrc/mlp_fwd.hip).  One training
step of the named workload is recorded, mixed (default) and with DURF_OBJ_MIX=0 (stand-alone k_mlp_fwd_ms launches).

  python tools/experiments/ms_stamps.py [--config cfg3] [--rays 512]
"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from durf_amd import _lib, train_boxpose

ap = argparse.ArgumentParser()
ap.add_argument('--config', default='cfg3')
ap.add_argument('--rays', type=int, default=512)
ap.add_argument('--bwd', action='store_true', help='the backward item (msb_bwd_pair) instead of the forward one')
a = ap.parse_args()
dev = torch.device('cuda', 0)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.durf_debug_ms_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.durf_debug_msb_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
read_stamps = lib.durf_debug_msb_stamps if a.bwd else lib.durf_debug_ms_stamps
w = bench.setup_workload(a.config, dev, rays=a.rays)
model, config, state, batch, prev = (w[k] for k in ('model', 'config', 'state', 'batch', 'prev'))
rng = 0
for i in range(4):
    state, stats, rng, _ = train_boxpose.train_step_one_call(model, config, rng, state, batch, 5e-4, 3.0, w['alpha'], prev, reduce_stats=False)
torch.cuda.synchronize()
read_stamps(None, 1)
state, stats, rng, _ = train_boxpose.train_step_one_call(model, config, rng, state, batch, 5e-4, 3.0, w['alpha'], prev, reduce_stats=False)
torch.cuda.synchronize()
buf = np.zeros((4096, 32), dtype=np.uint64)
n = read_stamps(buf.ctypes.data, 1)
print('items recorded in one step: %d (both levels)' % n)
s = buf[:min(n, 4096)].astype(np.int64)
t0 = s[:, 0].min()
if a.bwd:
    rows = []
    for r in s:
        ent = [r[1] - r[0], r[2] - r[1], r[3] - r[2]]
        post = r[3]
        for i in range(10):
            ent += [r[16 + i] - post, r[4 + i] - r[16 + i]]
            post = r[4 + i]
        ent.append(r[15] - post)
        rows.append(ent)
    rows = np.array(rows, dtype=np.float64) * 0.01
    tot = (s[:, 15] - s[:, 0]) * 0.01
    print('backward item: mean %.2f us  min %.2f  max %.2f' % (tot.mean(), tot.min(), tot.max()))
    lab = ['wait prev', 'head gradients', 'barrier'] + sum((['stage %d work' % i, 'stage %d barrier' % i] for i in range(10)), []) + ['last stage']
    for l, m, mx in zip(lab, rows.mean(0), rows.max(0)):
        print('  %-18s mean %6.2f us   max %6.2f' % (l, m, mx))
    start = (s[:, 0] - t0) * 0.01; end = (s[:, 15] - t0) * 0.01
    print('first item starts 0.0, last item ends %.1f us after it; %d distinct workgroups' % (end.max(), len(set((s[:, 30] >> 32).tolist()))))
    hist, edges = np.histogram(start, bins=12)
    print('  start histogram:', ' '.join('%d@%.0f' % (h, e) for h, e in zip(hist, edges[:-1])))
    sys.exit(0)
NAMES = ['wait prev', 'inputs', 'wait inputs'] + ['st%d work' % i for i in range(10)] + ['st10']
rows = []
for r in s:
    ent = [r[1] - r[0], r[2] - r[1], r[3] - r[2]]
    post = r[3]
    for i in range(10):
        ent.append(r[16 + i] - post)          # the stage's own work of role 0, up to its arrival at the barrier
        ent.append(r[4 + i] - r[16 + i])      # the wait at that barrier
        post = r[4 + i]
    ent.append(r[14] - post)
    rows.append(ent)
rows = np.array(rows, dtype=np.float64) * 0.01      # ticks of 10 ns -> us
tot = (s[:, 14] - s[:, 0]) * 0.01
print('item: mean %.2f us  min %.2f  max %.2f' % (tot.mean(), tot.min(), tot.max()))
lab = ['wait prev', 'inputs (enc)', 'barrier']
for i in range(10):
    lab += ['stage %d work' % i, 'stage %d barrier' % i]
lab += ['stage 10 + raw']
for l, m, mx in zip(lab, rows.mean(0), rows.max(0)):
    print('  %-18s mean %6.2f us   max %6.2f' % (l, m, mx))
if os.environ.get('MS_STAMP_STAGE'):       # variant built with -DMS_STAMP_STAGE=2: slots 26-29 are the inside of stage 2
    st = np.stack([s[:, 26] - s[:, 5], s[:, 27] - s[:, 26], s[:, 28] - s[:, 27], s[:, 29] - s[:, 28], s[:, 18] - s[:, 29], s[:, 6] - s[:, 18]], 1) * 0.01
    for l, m in zip(['mask flush, weight requests, ring reads issued', 'this stage\'s weights have arrived', '16 MFMAs, results readable',
                     'pack + LDS writes + stash stores issued', 'lgkmcnt(0) before the barrier', 'barrier'], st.mean(0)):
        print('    stage 2: %-48s mean %6.2f us' % (l, m))
    sys.exit(0)
enc = np.stack([s[:, 26] - s[:, 1], s[:, 27] - s[:, 26], s[:, 28] - s[:, 27], s[:, 29] - s[:, 28], s[:, 2] - s[:, 29]], 1) * 0.01
for l, m in zip(['ray_idx load', 'ray data loads', 'frustum gaussian', 'features + stores', 'view / load_w issue'], enc.mean(0)):
    print('    inputs: %-20s mean %6.2f us' % (l, m))
# the launch-level picture: item start / end offsets from the first stamp, per workgroup
start = (s[:, 0] - t0) * 0.01
end = (s[:, 14] - t0) * 0.01
wg = s[:, 30] >> 32
order = np.argsort(start)
print('first item starts 0.0, last item ends %.1f us after it; distinct workgroups %d' % (end.max(), len(set(wg.tolist()))))
half = len(order) // 2
for name, sel in (('level 0', order[:half]), ('level 1', order[half:])):
    st, en = start[sel], end[sel]
    print('  %s: %d items, starts %.1f .. %.1f, ends %.1f .. %.1f us' % (name, len(sel), st.min(), st.max(), en.min(), en.max()))
    hist, edges = np.histogram(st - st.min(), bins=8)
    print('    start histogram (us from the level\'s first item):', ' '.join('%d@%.0f' % (h, e) for h, e in zip(hist, edges[:-1])))
