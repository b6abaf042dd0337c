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
    json.dump(dict(lib_version=7, workload='cfg3', rays_per_gpu=4096, source='x', mlp_dw_256=dict(total_bytes=5.0)),
              open(tmp_path / 'profiles' / 'r09_pmc_traffic.json', 'w'))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.pmc_traffic(7, 'cfg3', 4096, 'mlp_dw_256')[0] == 5.0
    assert bench.pmc_traffic(8, 'cfg3', 4096, 'mlp_dw_256')[0] is None      # other build: stale numbers are not quoted
    assert bench.pmc_traffic(7, 'cfg2', 4096, 'mlp_dw_256')[0] is None
    assert bench.pmc_traffic(7, 'cfg3', 1024, 'mlp_dw_256')[0] is None
