import sys, torch
sys.path.insert(0, '.')
from durf_amd import obbpose_model, ops, synthetic, train_boxpose, utils
from tests import helpers as H
from tests.test_gpu_fused_encode import _setup
cuda = torch.device('cuda:0')
B, K, N = 512, 3, 64
config, b, db, model, variables, noise = _setup(cuda, B, K, N, 31 + K)
for train in (False, True):
    out = {}
    for on in (True, False):
        ops.FWD_SCATTER_RAW = on
        ret, ctx = model._forward(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], True, False, False, 10.0,
                                  train=True, noise=noise, loss_prep=None) if train else (None, None)
        if not train:
            # inference: reach raw through the ops directly
            ret, ctx = model._forward(variables, 0, db['rays'], db['init'], db['ext'], b['ts'], True, False, False, 10.0,
                                      train=False, noise=noise, loss_prep=None)
        torch.cuda.synchronize()
        out[on] = (ret, ctx)
    a, c = out[True][0][0][0], out[False][0][0][0]
    bad = ((a != c) & ~(torch.isnan(a) & torch.isnan(c))).any(dim=-1)
    dd = out[True][1].get('dedup') if out[True][1] else None
    print('train', train, 'rays differing', int(bad.sum()), 'of', B)
    if dd is not None:
        slot = dd['slot'].view(B, 2)
        print('  class0', int(dd['count'][0]), 'class1', int(dd['count'][1]), 'multi', int(dd['multi_hit']))
        print('  differing rays in class 1:', int((bad & (slot[:, 1] >= 0)).sum()), ' in class 0:', int((bad & (slot[:, 0] >= 0)).sum()))
        idx = bad.nonzero().flatten()[:8]
        print('  first differing rays', idx.tolist(), 'slot', slot[idx].tolist())
        print(a[idx[:3]], c[idx[:3]])
    if train:
        ra, rc = out[True][1]['levels'][0]['raw_b'].view(B, N, 4), out[False][1]['levels'][0]['raw_b'].view(B, N, 4)
        badr = ((ra != rc) & ~(torch.isnan(ra) & torch.isnan(rc))).any(dim=-1)
        print('  raw rows differing per ray (first 8 bad rays):', badr.sum(dim=1)[badr.any(dim=1)][:8].tolist())
        r0 = int(badr.any(dim=1).nonzero().flatten()[0]) if badr.any() else -1
        if r0 >= 0:
            print('  ray', r0, 'bad samples', badr[r0].nonzero().flatten().tolist()[:70])
            print(ra[r0, :3], rc[r0, :3])
