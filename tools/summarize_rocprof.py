#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small text tables (committed under profiles/).

usage: summarize_rocprof.py <dir> [--steps K]
  *_kernel_stats.csv        -> per-kernel calls / total / average duration
  *_counter_collection.csv  -> per-kernel average of every collected counter
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=70):
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main():
    d = sys.argv[1]
    for path in sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)):
        rows = list(csv.DictReader(open(path)))
        print('== kernel stats: %s' % os.path.relpath(path, d))
        print('%-72s %8s %12s %12s %7s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'pct'))
        for r in rows:
            name = r.get('Name') or r.get('KernelName') or ''
            calls = int(float(r.get('Calls', 0)))
            tot = float(r.get('TotalDurationNs', 0)) / 1e6
            avg = float(r.get('AverageNs', 0)) / 1e3
            pct = r.get('Percentage', '')
            print('%-72s %8d %12.3f %12.2f %7s' % (short(name), calls, tot, avg, pct[:6]))
    for path in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(path)):
            k = r.get('Kernel_Name') or r.get('KernelName') or ''
            c = r.get('Counter_Name') or r.get('CounterName') or ''
            v = float(r.get('Counter_Value') or r.get('CounterValue') or 0)
            a = acc[k][c]
            a[0] += v
            a[1] += 1
        print('== counters (average per dispatch): %s' % os.path.relpath(path, d))
        for k in sorted(acc):
            print('%-72s %s' % (short(k), '  '.join('%s=%.4g (n=%d)' % (c, a[0] / a[1], a[1]) for c, a in sorted(acc[k].items()))))


if __name__ == '__main__':
    main()
