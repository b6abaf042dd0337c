#!/usr/bin/env python3
"""GPU idle share of the inference path from a rocprofv3 --kernel-trace of `bench.py --mode eval`: the second half of the
trace (warm-up images excluded) -- busy time, idle time between kernels, launches.   eval_idle_gaps.py <dir>"""
import csv, glob, os, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True))[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path)))
seg = ev[len(ev) // 2:]
span = seg[-1][1] - seg[0][0]
busy, cur = 0, seg[0][0]
for s, e, n in seg:
    cur = max(cur, s)
    if e > cur:
        busy += e - cur
        cur = e
fw = [e - s for s, e, n in seg if n.startswith('void k_mlp_fwd<256') or n.startswith('k_mlp_fwd<256')]
print('launches %d, span %.2f ms, GPU busy %.2f ms, idle %.2f ms (%.1f %%); background forward: %d launches, %.2f ms (%.1f %% of the span)' % (
    len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, len(fw), sum(fw) / 1e6, 100.0 * sum(fw) / span))
