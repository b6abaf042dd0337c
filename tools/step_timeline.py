#!/usr/bin/env python3
"""One training step as a timeline from a rocprofv3 --kernel-trace directory: every launch in order with its duration
and the idle gap before it.   step_timeline.py <dir> <step index among the traced steps>"""
import csv, glob, os, sys
d, which = sys.argv[1], int(sys.argv[2])
path = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path)))
adam = [i for i, e in enumerate(ev) if e[2].startswith('k_adam')]
seg = ev[adam[which - 1] + 1: adam[which] + 1]
t0, prev_end = seg[0][0], seg[0][0]
tot_gap = 0
for s, e, n in seg:
    gap = max(0, s - prev_end)
    tot_gap += gap
    print('%9.1f us  +%6.1f gap  %8.1f us  %s' % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, n.replace('void ', '')[:70]))
    prev_end = max(prev_end, e)
print('step span %.1f us, %d launches, idle %.1f us' % ((prev_end - t0) / 1e3, len(seg), tot_gap / 1e3))
