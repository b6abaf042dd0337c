"""rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE ... output dir -> per kernel: duration, cycles, effective clock."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
dur = {}
for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r.get('Dispatch_Id')] = (r.get('Kernel_Name'), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
acc = defaultdict(lambda: defaultdict(list))
hdr = None
for p in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    rd = csv.DictReader(open(p))
    hdr = rd.fieldnames
    for r in rd:
        k = r['Kernel_Name']
        if 'k_mlp' not in k:
            continue
        did = r['Dispatch_Id']
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if did in dur:
            acc[k]['us:' + did] = [dur[did][1]]
        elif 'Start_Timestamp' in r and r.get('End_Timestamp'):
            acc[k]['us:' + did] = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3]
print('columns:', hdr)
for k, c in sorted(acc.items()):
    us = [v[0] for n, v in c.items() if n.startswith('us:')]
    line = k[:60]
    mean = lambda x: sum(x) / max(len(x), 1)
    u = mean(us)
    line += '  us=%.1f (n=%d)' % (u, len(us))
    for n, v in sorted(c.items()):
        if not n.startswith('us:'):
            line += '  %s=%.4g' % (n, mean(v))
    if 'GRBM_GUI_ACTIVE' in c and u > 0:
        line += '  clock=%.3f GHz (GRBM/8/us)' % (mean(c['GRBM_GUI_ACTIVE']) / 8 / u / 1e3)
    print(line)
