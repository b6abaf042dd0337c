"""Host-side logic that needs no GPU: gin reader, config/type mirrors, parameter layout, the
C-ABI library (loads, exports every symbol include/durf_hip.h declares), synthetic batches,
layout helpers."""
import dataclasses
import os
import re

import numpy as np
import pytest
import torch

from durf_amd import _lib, obbpose_model, ops, synthetic, train_boxpose, utils
from oracle import durf_ref as R
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gin_reader_on_shipped_configs():
    utils.clear_gin()
    utils.parse_gin(os.path.join(ROOT, 'configs', 'waymo.gin'))
    c = utils.configured(utils.Config)
    assert (c.batch_size, c.far, c.grad_max_val, c.grad_max_norm) == (512, 40.0, 0.1, 1.0)
    assert (c.depth_loss_mult, c.near_loss_mult, c.empty_loss_mult, c.sky_loss_mult) == (1e-4, 1e-2, 1.0, 1.0)
    m = utils.configured(obbpose_model.MipNerfModel)
    assert (m.num_samples, m.density_noise, m.no_pose_opt, m.no_yaw_opt, m.contraction) == (128, 0.0, True, True, True)
    assert utils.configured(obbpose_model.MLP).net_width == 256
    utils.clear_gin()
    utils.parse_gin(os.path.join(ROOT, 'configs', 'carla_dyn.gin'))
    assert utils.configured(utils.Config).far == 200.0
    utils.clear_gin()
    # defaults of the dataclasses are the reference's (internal/utils.py:93-144, obbpose_model.py:45-66)
    d = utils.Config()
    assert (d.batch_size, d.lr_init, d.lr_final, d.lr_delay_steps, d.coarse_loss_mult, d.rand_bkgd) == \
        (4096, 5e-4, 5e-6, 2500, 0.1, True)
    m = obbpose_model.MipNerfModel()
    assert (m.num_samples, m.num_levels, m.resample_padding, m.density_bias, m.density_noise) == (128, 2, 0.01, -1., 0.1)


def test_gin_reader_errors_and_syntax():
    utils.clear_gin()
    utils.parse_gin("Config.batch_size = 64  # comment\nMLP.net_activation = @flax.nn.relu\n"
                    "Config.c2f_steps = (1, 2, 3)\ninternal.utils.Config.far = 7.5\n")
    c = utils.configured(utils.Config)
    assert c.batch_size == 64 and c.c2f_steps == (1, 2, 3) and c.far == 7.5
    assert utils.configured(obbpose_model.MLP).net_activation == 'relu'
    with pytest.raises(ValueError):
        utils.parse_gin('this is not gin')
    utils.clear_gin()
    utils.parse_gin('Config.no_such_knob = 1')
    with pytest.raises(ValueError):
        utils.configured(utils.Config)
    utils.clear_gin()


def test_unsupported_knobs_fail_loudly():
    utils.clear_gin()
    m = obbpose_model.MipNerfModel(stop_level_grad=False, density_noise=0.0)
    with pytest.raises(NotImplementedError):
        m._check()
    obbpose_model.MipNerfModel(use_viewdirs=False)._check()        # (the static model's knob: durf_amd/noview.py, tests/test_noview.py)
    m = obbpose_model.MipNerfModel(num_samples=100)
    with pytest.raises(NotImplementedError):
        m._check()
    obbpose_model.MipNerfModel(num_samples=64)._check()


def test_param_layout_matches_reference_counts():
    lay = obbpose_model.ParamLayout(5, 3)
    assert lay.mlp_size[256] == 594308 and lay.mlp_size[128] == 168836
    assert lay.total == 5 * 3 * 6 + 594308 + 3 * 168836
    flat = torch.arange(lay.total, dtype=torch.float32)
    v = obbpose_model.Variables(flat, lay)
    p = v['params']
    assert p['box_centers'].shape == (5, 3, 6)
    assert p['MLP_0']['Dense_5']['kernel'].shape == (316, 256)
    assert p['MLP_0']['Dense_10']['kernel'].shape == (283, 128)
    assert p['BoxMLP_2']['Dense_5']['kernel'].shape == (191, 128)
    assert p['BoxMLP_2']['Dense_11']['bias'].shape == (3,)
    # views alias the flat buffer, and the last leaf ends exactly at the end
    assert p['BoxMLP_2']['Dense_11']['bias'].data_ptr() + 3 * 4 == flat.data_ptr() + lay.total * 4
    # same order/shapes as the oracle's flax restatement
    shapes = R.mlp_layer_shapes(60, 27, R.MLP_BKGD)
    assert [tuple(p['MLP_0']['Dense_%d' % i]['kernel'].shape) for i in range(12)] == shapes


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'durf_hip.h')).read()
    declared = set(re.findall(r'\b(durf_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(L, name), 'libdurf_hip.so does not export %s' % name
    assert declared == set(_lib.symbols()), declared ^ set(_lib.symbols())
    # ... and nothing else: the dynamic symbol table of the shared object == the header (an exported entry point the header
    # does not declare is outside the stub, the binding table and this drift test)
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln and ln.split()[-1].startswith('durf_')}
    assert exported == declared, 'exported but not declared: %s; declared but not exported: %s' % (
        sorted(exported - declared), sorted(declared - exported))
    assert L.durf_version() >= 1
    assert L.durf_mlp_param_count(256, 60) == 594308
    assert L.durf_mlp_param_count(128, 63) == 168836
    for layer in range(12):
        fi, fo = R.mlp_layer_shapes(60, 27, R.MLP_BKGD)[layer]
        assert L.durf_mlp_layer_offset(256, 60, layer, 1) - L.durf_mlp_layer_offset(256, 60, layer, 0) == fi * fo


def test_dispatch_log_constants_match_the_header_and_the_log_works_without_a_gpu():
    """ops.DISPATCH mirrors include/durf_hip.h's DURF_DISPATCH_* (the launchers' variant log the GPU suite's dispatch
    matrix reads); the log itself is host state: readable and resettable with no device"""
    from durf_amd import ops
    hdr = open(os.path.join(ROOT, 'include', 'durf_hip.h')).read()
    declared = {m.group(1): int(m.group(2), 16) for m in re.finditer(r'#define DURF_DISPATCH_(\w+) (0x[0-9a-fA-F]+)', hdr)}
    assert declared == ops.DISPATCH
    assert len(set(declared.values())) == len(declared) and all(v & (v - 1) == 0 for v in declared.values())
    ops.dispatch_reset()
    assert ops.dispatch_seen() == set()


def test_integration_stub_is_generated_from_the_header():
    """INTEGRATION.md's ctypes stub and include/durf_ctypes_stub.py are generated from include/durf_hip.h, and the
    product's own binding table (durf_amd/_lib.py) declares the same argument types for every symbol."""
    import ctypes as C
    import subprocess
    import sys
    gen = os.path.join(ROOT, 'tools', 'gen_integration_stub.py')
    p = subprocess.run([sys.executable, gen, '--check'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_integration_stub as G
    env = dict(C=C, vp=C.c_void_p, i32=C.c_int, f32=C.c_float, u64=C.c_size_t)
    fns = G.parse_header()
    assert sorted(n for n, _, _ in fns) == _lib.symbols()
    for name, ret, params in fns:
        want_ret, want_args = _lib._SIGS[name]
        assert eval(ret, env) is want_ret, name
        got = [eval(t, env) for t, _ in params]
        assert got == list(want_args), (name, [n for _, n in params])
    # the stub module itself binds against the built library
    sys.path.insert(0, os.path.join(ROOT, 'include'))
    import durf_ctypes_stub
    L = durf_ctypes_stub.bind(_lib.LIB_PATH)
    assert L.durf_version() == _lib.lib().durf_version()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libdurf_hip.so')
    with pytest.raises(RuntimeError, match='no fallback'):
        _lib.lib()


def test_synthetic_batch_schema():
    b = synthetic.make_batch(512, 3, seed=9)
    r = b['rays']
    assert r['origins'].shape == (512, 3) and r['radii'].shape == (512, 1) and r['far'].shape == (512, 1)
    assert b['init'].shape == (5, 3, 6) and b['ext'].shape == (3, 3) and b['target'].shape == (3, 6)
    assert 0 <= b['ts'] < 5 and all(v.dtype == np.float32 for v in r.values())
    assert 0.03 <= b['hit_fraction'] <= 0.2
    np.testing.assert_allclose(np.linalg.norm(r['viewdirs'], axis=-1), 1.0, rtol=1e-5)
    assert abs(float(np.median(r['radii'])) - 1.1e-3) < 3e-4
    # no ray hits two boxes (default): the oracle's dyn_mask is 0/1
    ob = H.oracle_batch(b)
    params = R.init_params(0, ob['init'], 3)
    with torch.no_grad():
        ret = R.model_apply(params, ob['rays'], b['ts'], ob['ext'], False, False, False, 10.0, cfg=dict(num_samples=4))
    assert int(ret[0][8].max()) == 1
    assert abs(float(ret[0][8].float().mean()) - b['hit_fraction']) < 1e-6


def test_tile_layout_helpers_roundtrip():
    x = torch.randn(96, 64).to(torch.bfloat16).float()
    t = H.tile(x, 4)
    assert torch.equal(H.untile(t, 96, 4), x)
    # documented formula (include/durf_hip.h): elem(row,f) at (((row/32)*nks + f/16)*64 + ((f/8)&1)*32 + row%32)*8 + f%8
    flat = t.reshape(-1).float()
    for row, f in ((0, 0), (33, 17), (95, 63), (40, 8)):
        off = (((row // 32) * 4 + f // 16) * 64 + ((f // 8) & 1) * 32 + row % 32) * 8 + f % 8
        assert float(flat[off]) == float(x[row, f])
    perm = H.cperm_cols(16)
    assert sorted(perm.tolist()) == list(range(256))


def test_barf_weights_host_side():
    for alpha in (0.0, 2.5, 4.5, 10.0):
        w = ops.barf_weights(alpha)
        ref = R.barf_weights(alpha, 10, torch.float32).numpy()
        np.testing.assert_allclose(w, ref, rtol=0, atol=2e-7)


def test_level_multipliers_follow_total_loss():
    c = utils.Config(coarse_loss_mult=0.1, sky_loss_mult=2.0, depth_loss_mult=3.0, near_loss_mult=4.0, empty_loss_mult=5.0)
    assert train_boxpose.level_multipliers(c, 1, 2) == [1.0, 20.0, 3.0, 4.0, 5.0, 1e-6]
    np.testing.assert_allclose(train_boxpose.level_multipliers(c, 0, 2), [0.1, 2.0, 0.3, 0.4, 0.5, 1e-6])
    # cross-check with the oracle's total_loss on random per-level terms
    S = {k: torch.rand(2, dtype=torch.float64) for k in ('losses', 's_losses', 'd_losses', 'n_losses', 'e_losses', 'distr_losses')}
    S['tv_losses'] = torch.zeros(2, dtype=torch.float64)
    want = R.total_loss(S, dict(dataclasses.asdict(c)))
    got = 0.0
    for lvl in range(2):
        m = train_boxpose.level_multipliers(c, lvl, 2)
        got = got + m[0] * S['losses'][lvl] + m[1] * S['s_losses'][lvl] + m[2] * S['d_losses'][lvl] + \
            m[3] * S['n_losses'][lvl] + m[4] * S['e_losses'][lvl] + m[5] * S['distr_losses'][lvl]
    assert abs(float(got) - float(want)) < 1e-12


def test_shard_batch_errors_like_the_reference():
    b = synthetic.make_batch(30, 1, seed=1)
    ob = {k: (torch.tensor(v) if isinstance(v, np.ndarray) else v) for k, v in b.items() if k != 'rays'}
    ob['rays'] = utils.BoxRays(**{k: torch.tensor(v) for k, v in b['rays'].items()})
    with pytest.raises(ValueError, match='divisible'):
        train_boxpose.shard_batch(ob, 0, 4)
    s = train_boxpose.shard_batch(ob, 1, 2)
    assert s['pixels'].shape[0] == 15 and s['rays'].origins.shape[0] == 15 and s['init'].shape == ob['init'].shape
    assert torch.equal(s['pixels'], ob['pixels'][15:])


def test_c2f_schedule():
    from durf_amd import raygen
    steps = (1000, 2000, 3000)
    got = [raygen.c2f_factor(i, steps) for i in (0, 1000, 1001, 2000, 2001, 3000, 3001, 10 ** 6)]
    assert got == [16, 16, 12, 12, 8, 8, 4, 4]          # c2f_obb_dataset.py:306-313 (inclusive upper bounds)


def test_argument_structs_match_the_header(tmp_path):
    """durf_forward_args / durf_train_args: the ctypes Structures of durf_amd/ops.py against the C definitions in
    include/durf_hip.h, compiled here with gcc -- size and the offset of every field."""
    import ctypes
    import subprocess
    from durf_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include <stdint.h>', '#include "%s/include/durf_hip.h"' % root,
             'int main(void) {']
    for cname, cls in (('durf_forward_args', ops.ForwardArgs), ('durf_train_args', ops.TrainArgs),
                       ('durf_loss_level', ops.LossLevel)):
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for name, _ in cls._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, name, cname, name))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', str(src), '-o', str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in (('durf_forward_args', ops.ForwardArgs), ('durf_train_args', ops.TrainArgs),
                       ('durf_loss_level', ops.LossLevel)):
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for name, _ in cls._fields_:
            assert int(got['%s.%s' % (cname, name)]) == getattr(cls, name).offset, (cname, name)


def test_host_thread_pinning_picks_the_cores_near_the_gpu(tmp_path):
    """train_boxpose._cpus_near_gpu: each rank of a multi-rank run is pinned (os.sched_setaffinity, before its first GPU
    call) to the cores sysfs lists as local to its GPU, shared out among the ranks that list the same cores; without sysfs
    information an even split of the cores the process may use"""
    from durf_amd import train_boxpose as tb
    avail = set(range(16))
    # no sysfs: even split
    assert tb._cpus_near_gpu(0, 2, avail, sysfs=str(tmp_path / 'none')) == list(range(0, 8))
    assert tb._cpus_near_gpu(1, 2, avail, sysfs=str(tmp_path / 'none')) == list(range(8, 16))
    assert tb._cpus_near_gpu(3, 4, set(range(8)), sysfs=str(tmp_path / 'none')) == [6, 7]
    # four AMD render devices, two per NUMA node, and one device of another vendor that must not shift the numbering
    sysfs = tmp_path / 'drm'
    for i, (pci, vendor, cpus) in enumerate([('0000:05:00.0', '0x1002', '0-7'), ('0000:15:00.0', '0x1002', '0-7'),
                                             ('0000:10:00.0', '0x10de', '0-15'), ('0000:85:00.0', '0x1002', '8-15'),
                                             ('0000:95:00.0', '0x1002', '8-11,12-15')]):
        real = tmp_path / 'pci' / pci
        real.mkdir(parents=True)
        (real / 'vendor').write_text(vendor + '\n')
        (real / 'local_cpulist').write_text(cpus + '\n')
        d = sysfs / ('renderD%d' % (128 + i))
        d.mkdir(parents=True)
        os.symlink(str(real), str(d / 'device'))
    got = [tb._cpus_near_gpu(r, 4, avail, sysfs=str(sysfs)) for r in range(4)]
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]], got
    # cores outside the process's own affinity mask are never chosen
    assert tb._cpus_near_gpu(0, 4, {2, 3, 9}, sysfs=str(sysfs)) == [2]
    # one rank per node: nothing to do
    assert tb.pin_host_thread(0, 1) is None
