"""gaps before / after every k_big variant in a rocprofv3 --kernel-trace of tools/probes/probe_gaps (see there)"""
import csv
import glob
import sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
acc = defaultdict(lambda: [[], [], []])
seen = defaultdict(int)
for i in range(1, len(rows) - 1):
    n = rows[i]['Kernel_Name']
    if 'k_big' not in n:
        continue
    seen[n] += 1
    key = '%s #%d' % (n[:40], (seen[n] - 1) // 40)          # the same instantiation is used by several sequences
    before = int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])
    after = int(rows[i + 1]['Start_Timestamp']) - int(rows[i]['End_Timestamp'])
    acc[key][0].append(before); acc[key][1].append(after)
    acc[key][2].append(int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp']))
for k, (b, a, d) in acc.items():
    b, a, d = sorted(b), sorted(a), sorted(d)
    print('%-48s n %3d  gap before %6.2f us  after %6.2f us  duration %7.2f us (medians)' % (
        k, len(b), b[len(b) // 2] / 1e3, a[len(a) // 2] / 1e3, d[len(d) // 2] / 1e3))
