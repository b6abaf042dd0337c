"""bench.py's host logic on CPU: `--gpus N` launches itself (no torch.distributed.run), the workloads bind the
reference's gin knobs, and the roofline pricing follows SURVEY.md 8(d)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('n', [1, 2])
def test_bench_spawns_its_own_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--selftest-launch'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                  # rank 0 only
    out = json.loads(lines[0])
    assert out['ok'] and out['n_gpus'] == n and out['selftest'] == 'launch'


def test_bench_failed_rank_stops_the_others():
    """a rank that dies must not leave the launcher waiting on the survivors' collective"""
    env = dict(os.environ, DURF_SELFTEST_FAIL_RANK='1')
    env.pop('WORLD_SIZE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--selftest-launch'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode != 0


def test_workloads_bind_reference_gin():
    sys.path.insert(0, ROOT)
    import bench
    w = bench.setup_workload('cfg3', 'cpu', rays=64)
    c, m = w['config'], w['model']
    assert (w['K'], w['far'], w['B']) == (3, 40.0, 64) and c.far == 40.0       # configs/waymo.gin:12-13
    assert (c.depth_loss_mult, c.near_loss_mult, c.empty_loss_mult, c.sky_loss_mult) == (1e-4, 1e-2, 1.0, 1.0)
    assert m.num_samples == 128 and m.no_pose_opt and m.no_yaw_opt and m.contraction
    assert w['batch']['init'].shape[1] == 3 and w['state'].variables.layout.K == 3
    w = bench.setup_workload('cfg4', 'cpu', rays=64)
    assert not w['model'].no_pose_opt and not w['model'].no_yaw_opt and w['alpha'] == 3.3
    w = bench.setup_workload('cfg2', 'cpu', rays=64)
    assert (w['K'], w['config'].far) == (1, 200.0)
    assert bench.setup_workload('cfg5', 'cpu', rays=32)['K'] == 8
    # sharding: rank r of 2 gets rays [r*B, (r+1)*B) of the same global batch
    a = bench.setup_workload('cfg3', 'cpu', rank=1, world=2, rays=32)
    assert a['batch']['pixels'].shape[0] == 32
    assert (a['batch']['pixels'].numpy() == a['batch_np']['pixels'][32:64]).all()


def test_pmc_traffic_is_version_gated(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    os.makedirs(tmp_path / 'profiles')
    json.dump(dict(lib_version=7, workload='cfg3', rays_per_gpu=4096, source='x',
                   mlp_dw_256=dict(total_bytes=5.0, mfma_busy_cycles=64.0)),
              open(tmp_path / 'profiles' / 'r09_pmc_traffic.json', 'w'))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    e = bench.pmc_entry(7, 'cfg3', 4096, 'mlp_dw_256')
    assert e['total_bytes'] == 5.0 and e['mfma_busy_cycles'] == 64.0 and e['source'].startswith('r09_pmc_traffic.json')
    assert bench.pmc_entry(8, 'cfg3', 4096, 'mlp_dw_256') is None      # other build: stale numbers are not quoted
    assert bench.pmc_entry(7, 'cfg2', 4096, 'mlp_dw_256') is None
    assert bench.pmc_entry(7, 'cfg3', 1024, 'mlp_dw_256') is None


def test_extra_workloads_cover_every_baseline_config():
    """the short passes behind the headline (`workloads` in the JSON line): every BASELINE.json configuration, the
    reference's own 512-ray batch (configs/waymo.gin:17) and its own arithmetic (internal/math.py:22-24)"""
    sys.path.insert(0, ROOT)
    import bench
    names = [w[0] for w in bench.EXTRA_WORKLOADS]
    assert names == ['cfg1', 'cfg2', 'cfg3_512rays', 'cfg4', 'cfg4_bf16x3', 'cfg5', 'cfg3_f32']
    for name, cfg, rays, prec, steps, warm in bench.EXTRA_WORKLOADS:
        assert cfg in bench.WORKLOADS and prec in ('bf16', 'f32', 'bf16x3') and steps >= 5 and warm >= 2
    o = dict(value=1.0, ms_per_step=2.0, steps=3, dtype='bf16', loss=0.5,
             config=dict(rays_per_gpu=512, num_samples=64, objects=0, pose_opt=False),
             roofline=dict(kernel='mlp_dw_256', bound='hbm', frac=0.3, mfma_frac=0.3, hbm_dataflow_frac=0.7, launch_us=9.0,
                           step_mlp_frac=0.2, non_mlp_ms_per_step=0.1))
    s = bench.summarize_workload(o)
    # (round 6: `frac` is SURVEY 8(d)'s -- algorithmic FLOPs against the MFMA peak -- whatever `bound` says; the data-flow
    # bytes against HBM ride beside it under their own name)
    assert s['dominant'] == 'mlp_dw_256' and s['bound'] == 'hbm' and s['frac'] == 0.3 and s['mfma_frac'] == 0.3
    assert s['hbm_dataflow_frac'] == 0.7
    assert s['rays_per_s'] == 1.0 and s['ms_per_step'] == 2.0
