#!/usr/bin/env python3
"""GPU idle time inside the timed region of bench.py from a rocprofv3 --kernel-trace directory: how much of a step the GPU
spends between kernels (what a HIP graph could remove).   gpu_idle_gaps.py <dir> <steps> <warmup>"""
import csv
import glob
import os
import sys

d, steps, warm = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
path = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# one k_adam per step: cut the trace after the warm-up steps' last adam
adam = [i for i, e in enumerate(ev) if e[2].startswith('k_adam')]
assert len(adam) >= steps + warm, (len(adam), steps, warm)
first = adam[warm - 1] + 1 if warm else 0
last = adam[warm + steps - 1]
seg = ev[first:last + 1]
span = seg[-1][1] - seg[0][0]
busy, gaps, cur_end = 0, [], seg[0][0]
where = []                      # (gap, step within the segment, kernel before, kernel after)
step_i, prev_n = 0, ''
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, n))
        where.append((s - cur_end, step_i, prev_n[:36], n[:36]))
        cur_end = s
    if n.startswith('k_adam'):
        step_i += 1
    prev_n = n
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
print('steps %d: span %.3f ms/step, GPU busy %.3f ms/step, idle %.3f ms/step (%.1f %%), %d launches/step' % (
    steps, span / steps / 1e6, busy / steps / 1e6, (span - busy) / steps / 1e6, 100.0 * (span - busy) / span, len(seg) / steps))
gaps.sort(reverse=True)
print('largest gaps (us, before kernel):', [(round(g / 1e3, 1), n[:28]) for g, n in gaps[:8]])
import collections
c = collections.Counter()
for g, n in gaps:
    c[n[:40]] += g
print('idle by following kernel (us/step):', [(k, round(v / steps / 1e3, 1)) for k, v in c.most_common(8)])
where.sort(reverse=True)
print('largest gaps (us, step, after kernel -> before kernel):')
for g, st, a, b in where[:6]:
    print('   %10.1f  step %3d  %s -> %s' % (g / 1e3, st, a, b))
