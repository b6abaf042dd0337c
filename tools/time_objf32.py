"""The fp32 hit-ray path of a pose-optimisation step (cfg4) kernel by kernel: HIP-event times of the fp32 object encode /
forward / backward / weight gradients, the background MLP's fp32 evaluation of the box-hit rays, and the bf16 object calls
they replace, on the bench workload's own batch (python tools/time_objf32.py [cfg4] [rays])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from durf_amd import obbpose_model as om, ops

dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
w = bench.setup_workload(name, dev, rays=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
model, variables, batch = w['model'], w['state'].variables, w['batch']
B, K, N, alpha = w['B'], w['K'], model.num_samples, w['alpha']
rays = batch['rays']
lay = variables.layout
pose = variables['params']['box_centers'][int(batch['ts'])].contiguous()
radii = rays.radii.reshape(-1).contiguous()
o_s, d_s, hit, zo, view, t_vals = ops.ray_prologue(rays.origins, rays.directions, pose, batch['ext'].reshape(-1, 3).contiguous(),
                                                   rays.viewdirs, rays.near.reshape(-1).contiguous(),
                                                   rays.far.reshape(-1).contiguous(), N)
(idx, count, slot), cls = ops.compact_all(hit, N)
view27 = ops.view_enc(rays.viewdirs, want_f32=True)[1]
sz = lay.mlp_size[128]
o0 = lay.mlp_off['BoxMLP_0']
flat = variables.flat[o0:o0 + K * sz]
draw = torch.randn(B * N, 4, device=dev) * 0.01
print('%s: B=%d K=%d N=%d hits per object %s, class-1 rays %d' % (name, B, K, N, count.tolist(), int(cls[1][1])))


def timeit(label, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('  %-44s %8.1f us' % (label, e0.elapsed_time(e1) * 1e3 / n))


slabs = ops.ObjSlabsF32(K, B, N, dev, True)
ws = ops.mlp_f32_pack(128, 63, flat, K=K, param_stride=sz)
timeit('pack object weight streams', lambda: ops.mlp_f32_pack(128, 63, flat, K=K, param_stride=sz))
timeit('fp32 forward, encoding fused (one level)', lambda: ops.objf32_fwd_batch(slabs, idx, count, t_vals, o_s, d_s, radii, alpha, view27, flat, sz, ws))
timeit('fp32 encode + forward, two launches', lambda: ops.objf32_fwd_batch(slabs, idx, count, t_vals, o_s, d_s, radii, alpha, view27, flat, sz, ws, fused_encode=False))
import ctypes as C
from durf_amd import _lib
L = _lib.lib()
S = torch.cuda.current_stream().cuda_stream
timeit('  forward alone (reads the encoding)', lambda: L.durf_objf32_fwd_batch(
    S, K, B, N, idx.data_ptr(), count.data_ptr(), slabs.enc.data_ptr(), view27.data_ptr(), flat.data_ptr(), sz, ws.data_ptr(),
    slabs.raw.data_ptr(), slabs.act.data_ptr(), None, None, None, None, None, 0))
timeit('fp32 backward + d(enc) (one level)', lambda: ops.objf32_bwd_batch(slabs, idx, count, draw, flat, sz, ws, want_d_enc=True))
timeit('fp32 backward, no d(enc)', lambda: ops.objf32_bwd_batch(slabs, idx, count, draw, flat, sz, ws, want_d_enc=False))
grad = torch.zeros(K * sz, device=dev)
for ns in (2, 4, 8, 16):
    timeit('fp32 weight gradients (two levels), nsplit %d' % ns, lambda: ops.objf32_dw_batch([slabs, slabs], count, grad, sz, nsplit=ns))
timeit('background MLP, box-hit rays, fp32', lambda: ops.bkgd_hit_rays_f32(B, view27, variables.mlp_flat('MLP_0'), cls[0][1], cls[1][1:2])
       if hasattr(ops, 'bkgd_hit_rays_f32') else ops.mlp_fwd_f32(256, 60, B, 1, None, view27, variables.mlp_flat('MLP_0'),
                                                                 ray_idx=cls[0][1], count=cls[1][1:2]))
sums = torch.zeros(K, 21, device=dev)
slabs.d_enc = torch.randn(K, B * N, 64, device=dev)
for pr in (False, True):
    timeit('pose sums (one level), precise=%s' % pr, lambda: ops.encode_obj_bwd_batch(K, idx, count, slabs.d_enc, t_vals, o_s, d_s, radii, rays.origins,
                                                                                        rays.directions, pose, alpha, sums, precise=pr))
# the bf16 calls they replace
(pk_b, pk_o) = ops.pack_weights_all(variables.mlp_flat('MLP_0'), K, flat, sz, want_bwd=True)
sb = ops.ObjSlabs(K, B, N, dev, True)
vt = ops.obj_view_tiles(K, B, N, dev)
timeit('bf16 encode + forward (one level)', lambda: ops.obj_fwd_batch(sb, idx, count, t_vals, o_s, d_s, radii, alpha, view, pk_o[0], view_tile=vt))
timeit('bf16 backward + d(enc) (one level)', lambda: ops.obj_bwd_batch(sb, idx, count, draw, pk_o[1], want_d_enc=True))
timeit('bf16 weight gradients (two levels)', lambda: ops.obj_dw_batch([sb, sb], vt, count, grad, sz, flat))
