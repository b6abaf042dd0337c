#!/usr/bin/env python3
"""One table of current numbers from a default `python bench.py` JSON line (README):
    python tools/make_results_table.py profiles/r06_bench.json [BENCH_r05.json ...]
First file = the line the table is made of; further files (driver records `BENCH_rNN.json` or earlier bench lines) add a
column each with that file's rays/s per workload."""
import json
import sys


def load(path):
    txt = open(path).read().strip()
    d = json.loads(txt.splitlines()[-1]) if not txt.startswith('{\n') else json.loads(txt)
    if 'tail' in d and 'parsed' in d:         # a driver record (BENCH_rNN.json): the bench line is the last JSON line of its `tail`
        for ln in reversed(str(d['tail']).splitlines()):
            if ln.startswith('{"metric"'):
                return json.loads(ln)
        return d['parsed']
    return d


def rows_of(d):
    r = d['roofline']
    out = [('cfg3 (headline: 4096 rays x 128 x 2, K = 3)', d['value'], d['ms_per_step'], r['kernel'], r['bound'], r['frac'],
            r.get('hbm_dataflow_frac'), r.get('mfma_executed_frac'), r.get('hbm_frac_of_achievable'), r.get('non_mlp_ms_per_step'))]
    names = dict(cfg1='cfg1 (512 rays x 64 x 2, K = 0)', cfg2='cfg2 (4096 rays, K = 1)', cfg3_512rays='cfg3 at the reference\'s 512 rays',
                 cfg4='cfg4 (1024 rays, pose optimisation, hit rays fp32)', cfg5='cfg5 (1024 rays, K = 8)',
                 cfg3_f32='cfg3 in exact fp32 (`--precision f32`)', eval='eval: render_image 320 x 480, one C call per image')
    for k, w in (d.get('workloads') or {}).items():
        if 'error' in w:
            out.append((names.get(k, k), None, None, w['error'], None, None, None, None, None, None))
            continue
        out.append((names.get(k, k), w['rays_per_s'], w['ms_per_step'], w.get('dominant') or w.get('host_path'), w.get('bound'), w.get('frac'),
                    w.get('hbm_dataflow_frac'), None, w.get('hbm_frac_of_achievable'), w.get('non_mlp_ms_per_step')))
    return out


def main():
    d = load(sys.argv[1])
    others = [(p, load(p)) for p in sys.argv[2:]]
    f = lambda x, fmt: '' if x is None else fmt % x
    hdr = ['workload', 'k rays/s', 'ms / step', 'dominant kernel', 'nearer roof', 'frac (8d: FLOPs / 2.5 PF)', 'data-flow bytes / 8 TB/s',
           'MFMA executed', 'HBM of 6.29 TB/s', 'ms outside the 3 MLP kernels'] + [p for p, _ in others]
    print('| ' + ' | '.join(hdr) + ' |')
    print('|' + '---|' * len(hdr))
    keys = ['headline'] + list((d.get('workloads') or {}).keys())
    for key, row in zip(keys, rows_of(d)):
        cells = [row[0], f(row[1] and row[1] / 1e3, '%.1f'), f(row[2], '%.3f'), str(row[3] or ''), str(row[4] or ''), f(row[5], '%.3f'),
                 f(row[6], '%.3f'), f(row[7], '%.3f'), f(row[8], '%.2f'), f(row[9], '%.3f')]
        for _, o in others:
            v = o['value'] if key == 'headline' else ((o.get('workloads') or {}).get(key) or {}).get('rays_per_s')
            cells.append(f(v and v / 1e3, '%.1f'))
        print('| ' + ' | '.join(cells) + ' |')
    b = (d['roofline'].get('board') or {})
    print('\nboard: vendor GEMM %.0f TFLOP/s; cpu_baseline: %s' % (b.get('vendor_gemm_tflops', float('nan')),
                                                                  json.dumps(d.get('cpu_baseline'))))


if __name__ == '__main__':
    main()
