#!/usr/bin/env python3
"""Re-fill the numbers of BASELINE.md section 4, README.md and DESIGN.md section 7 from the round-6 evidence set in profiles/
(r06_bench.json, r06_rocprofv3_stats.txt, r06_pmc_traffic.json, r06_pytest_gpu.txt).  Text around the numbers is left alone."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(ROOT)
b = json.loads(open('profiles/r06_bench.json').read().strip().splitlines()[-1])
r = b['roofline']
A = r['all']
ss = r['single_stream']
W = b['workloads']
pmc = json.load(open('profiles/r06_pmc_traffic.json'))
stats = {}
for ln in open('profiles/r06_rocprofv3_stats.txt'):
    m = re.match(r'(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.e+-]+\s*$', ln)
    if m:
        stats[m.group(1)] = float(m.group(4))
def avg(prefix):
    return next(v for k, v in stats.items() if k.startswith(prefix))
dw_us, fwd_us, bwd_us, comp_us = avg('k_dw_all<256>'), avg('k_mlp_fwd<256'), avg('_Z9k_mlp_bwdILi256'), avg('k_composite_resample')
suite = re.search(r'(\d+) passed.* in ([\d.]+)s', open('profiles/r06_pytest_gpu.txt').read())
ver = pmc['lib_version']
flop_dw = A['mlp_dw_256']['achieved'] * 1e12 * A['mlp_dw_256']['us'] * 1e-6          # algorithmic FLOPs of the launch
frac_rocprof = flop_dw / (dw_us * 1e-6) / 2.5e15
GB = lambda k: pmc[k]['total_bytes'] / 1e9
# (a bench line made before its own version's PMC file existed carries no counter columns: the same arithmetic here)
for k_ in ('mlp_dw_256', 'mlp_fwd_256_train', 'mlp_bwd_256'):
    if A[k_].get('traffic_tbs') is None:
        A[k_]['traffic_tbs'] = pmc[k_]['total_bytes'] / (A[k_]['us'] * 1e-6) / 1e12
    if A[k_].get('mfma_executed_frac') is None:
        A[k_]['mfma_executed_frac'] = pmc[k_]['mfma_busy_cycles'] / 32.0 * 32768.0 / (A[k_]['us'] * 1e-6) / 2.5e15
tab = [l for l in subprocess.run([sys.executable, 'tools/make_results_table.py', 'profiles/r06_bench.json'], capture_output=True, text=True).stdout.splitlines() if l.startswith('|')]


def sub_row(s, prefix, newrow):
    i = s.index(prefix)
    j = s.index('\n', i)
    return s[:i] + newrow + s[j:]


# ---- BASELINE.md
s = open('BASELINE.md').read()
i, j = s.index('| workload | k rays/s | ms / step |'), s.index('| beside it | value |')
s = s[:i] + '\n'.join(tab) + '\n\n' + s[j:]
s = re.sub(r'`profiles/r06_bench.json`, library version \d+:', '`profiles/r06_bench.json`, library version %d:' % ver, s)
s = re.sub(r'vendor GEMM \d+ TFLOP/s; box-to-box', 'vendor GEMM %.0f TFLOP/s; box-to-box' % r['board']['vendor_gemm_tflops'], s)
s = sub_row(s, '| CPU baseline of the same run', '| CPU baseline of the same run (`cpu_baseline`: the oracle\'s `train_step`, fp32 PyTorch-CPU restatement, NOT JAX) | %.0f rays/s on %d threads (%s...) |'
            % (b['cpu_baseline']['value'], b['cpu_baseline']['cores'], b['cpu_baseline']['sample'][:70]))
s = sub_row(s, '| rocprofv3 average of the dominant kernel', '| rocprofv3 average of the dominant kernel, same command (`profiles/r06_rocprofv3_stats.txt`) | `k_dw_all<256>` %.1f us -> %.4f TFLOP / %.1f us / 2.5 PF = **%.3f** (line: %.3f from %.1f us of HIP events: %.1f %% apart) |'
            % (dw_us, flop_dw / 1e12, dw_us, frac_rocprof, r['frac'], r['launch_us'], 100 * (dw_us / r['launch_us'] - 1)))
s = sub_row(s, '| PMC bytes per launch', '| PMC bytes per launch (`profiles/r06_pmc_traffic.json`; FETCH x2, KiB; a line made with this file in the tree carries it as `roofline.traffic` -- the evidence line was made just before it: null there) | dW %.2f GB (%.2f TB/s: %.2f of the 6.29 TB/s a copy achieves), forward %.2f GB, backward %.2f GB, composite %.0f MB |'
            % (GB('mlp_dw_256'), A['mlp_dw_256']['traffic_tbs'], A['mlp_dw_256']['traffic_tbs'] / 6.29, GB('mlp_fwd_256_train'), GB('mlp_bwd_256'), GB('composite_resample') * 1e3))
s = sub_row(s, '| MFMA executed (`SQ_VALU_MFMA_BUSY_CYCLES`', '| MFMA executed (`SQ_VALU_MFMA_BUSY_CYCLES` / 32 x 32 768 FLOP; `mfma_executed_frac`) | dW %.3f of peak; forward %.3f; backward %.3f |'
            % (A['mlp_dw_256']['mfma_executed_frac'], A['mlp_fwd_256_train']['mfma_executed_frac'], A['mlp_bwd_256']['mfma_executed_frac']))
s = sub_row(s, '| GPU suite on the evidence box', '| GPU suite on the evidence box (`profiles/r06_pytest_gpu.txt`) | %s passed, 0 skipped (%.0f min); CPU suite 129 passed |' % (suite.group(1), float(suite.group(2)) / 60))
open('BASELINE.md', 'w').write(s)

# ---- README.md
s = open('README.md').read()
i, j = s.index('| workload | k rays/s | ms / step |'), s.index('* The headline (cfg3, 4096 rays')
s = s[:i] + '\n'.join(tab) + '\n\n' + s[j:]
s = re.sub(r'`profiles/r06_\*`, library version \d+\)', '`profiles/r06_*`, library version %d)' % ver, s)
open('README.md', 'w').write(s)

# ---- DESIGN.md section 7
s = open('DESIGN.md').read()
s = re.sub(r'\*\*Round-6 evidence set\*\*: `profiles/r06_\*` \(library version \d+,', '**Round-6 evidence set**: `profiles/r06_*` (library version %d,' % ver, s)
s = re.sub(r'`r06_pmc_traffic.json`; vendor GEMM \d+ TFLOP/s\)', '`r06_pmc_traffic.json`; vendor GEMM %.0f TFLOP/s)' % r['board']['vendor_gemm_tflops'], s)
d, f, w = A['mlp_dw_256'], A['mlp_fwd_256_train'], A['mlp_bwd_256']
s = sub_row(s, '| `k_dw_all<256>` (dominant) |', '| `k_dw_all<256>` (dominant) | %.0f (%.0f) | %.0f | HBM | **%.3f** (%.3f on the rocprof duration) | %.3f | %.3f | %.3f | %.2f GB -> %.2f (%.2f) |'
            % (d['us'], ss['mlp_dw_256']['us'], dw_us, d['frac'], frac_rocprof, d['hbm_dataflow_frac'], d['mfma_launched_frac'], d['mfma_executed_frac'], GB('mlp_dw_256'), d['traffic_tbs'], d['traffic_tbs'] / 6.29))
s = sub_row(s, '| `k_mlp_fwd<256,true,8,ENC>` (contains the encode) |', '| `k_mlp_fwd<256,true,8,ENC>` (contains the encode) | %.0f (%.0f) | %.0f | HBM by a nose; in fact neither (section 9) | %.3f (%.3f) | %.2f | %.2f | %.3f | %.2f GB -> %.1f (%.2f) |'
            % (f['us'], ss['mlp_fwd_256_train']['us'], fwd_us, f['frac'], ss['mlp_fwd_256_train']['frac'], f['hbm_dataflow_frac'], f['mfma_launched_frac'], f['mfma_executed_frac'], GB('mlp_fwd_256_train'), f['traffic_tbs'], f['traffic_tbs'] / 6.29))
s = sub_row(s, '| `k_mlp_bwd<256>` |', '| `k_mlp_bwd<256>` | %.0f (%.0f) | %.0f | as above | %.3f (%.3f) | %.2f | %.2f | %.3f | %.2f GB -> %.1f (%.2f) |'
            % (w['us'], ss['mlp_bwd_256']['us'], bwd_us, w['frac'], ss['mlp_bwd_256']['frac'], w['hbm_dataflow_frac'], w['mfma_launched_frac'], w['mfma_executed_frac'], GB('mlp_bwd_256'), w['traffic_tbs'], w['traffic_tbs'] / 6.29))
s = sub_row(s, '| `k_composite_resample` |', '| `k_composite_resample` | %.1f | %.1f | latency | %.3f of HBM | | | | %.1f MB (2.1x algorithmic) |'
            % (A['composite_resample']['us'], comp_us, A['composite_resample']['frac'], GB('composite_resample') * 1e3))
s = sub_row(s, '| whole step |', '| whole step | **%.3f ms = %.1f k rays/s** | | | step_mlp_frac %.2f | | | | ~17 GB -> 4.1 TB/s |' % (b['ms_per_step'], b['value'] / 1e3, r['step_mlp_frac']))
s = re.sub(r'`non_mlp_ms_per_step` [\d.]+ \(r05:', '`non_mlp_ms_per_step` %.3f (r05:' % r['non_mlp_ms_per_step'], s)
s = re.sub(r'CPU restatement: \d+ rays/s \(16 threads\)', 'CPU restatement: %.0f rays/s (16 threads)' % b['cpu_baseline']['value'], s)
k = lambda name: (W[name]['rays_per_s'] / 1e3, W[name]['ms_per_step'])
s = sub_row(s, '| cfg1 (BASELINE configs[0]: N = 64, K = 0, 512 rays) |', '| cfg1 (BASELINE configs[0]: N = 64, K = 0, 512 rays) | %.1f | %.3f | 13 |' % k('cfg1'))
s = sub_row(s, '| cfg3 at the reference\'s own batch (`--rays 512`, `configs/waymo.gin:17`) |',
            '| cfg3 at the reference\'s own batch (`--rays 512`, `configs/waymo.gin:17`) | **%.1f** (780-837 by box over the day) [r05 driver 757; mix off -> on +5.0 %%; role rotation +1.5 %%; encode split +1.5 %%; read ring +0.6 %%] | %.3f | 15 (18) |' % k('cfg3_512rays'))
s = sub_row(s, '| cfg2 (CARLA, K = 1) |', '| cfg2 (CARLA, K = 1) | %.1f | %.3f | 21 |' % k('cfg2'))
s = sub_row(s, '| cfg4 (pose optimisation, hit rays in exact fp32, 1024 rays) |', '| cfg4 (pose optimisation, hit rays in exact fp32, 1024 rays) | %.1f | %.3f | 29 |' % k('cfg4'))
s = sub_row(s, '| cfg4 with `obj_precision = \'bf16x3\'` |', '| cfg4 with `obj_precision = \'bf16x3\'` | %.1f [+0.9 %% interleaved] | %.3f | 29 |' % k('cfg4_bf16x3'))
s = sub_row(s, '| cfg5 (K = 8, 1024 rays) |', '| cfg5 (K = 8, 1024 rays) | **%.1f** (850-904 by box) [r05 driver 837; mix off -> on +3.9 %%; encode split +0.9 %%] | %.3f | 15 (18) |' % k('cfg5'))
s = sub_row(s, '| eval (`render_image`, 320 x 480, chunk 8192, one `durf_render_image` call per image) |',
            '| eval (`render_image`, 320 x 480, chunk 8192, one `durf_render_image` call per image) | %.0f | %.1f / image | |' % k('eval'))
s = sub_row(s, '| cfg3 in exact fp32 (`--precision f32`) |', '| cfg3 in exact fp32 (`--precision f32`) | %.1f | %.1f | |' % k('cfg3_f32'))
open('DESIGN.md', 'w').write(s)
s = open('profiles/README.md').read()
s = re.sub(r'\*\*Round 6\*\* \(library version \d+;', '**Round 6** (library version %d;' % ver, s)
open('profiles/README.md', 'w').write(s)
print('docs updated from the evidence set of library version', ver)
