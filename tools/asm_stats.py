"""Counts waits / memory ops / MFMAs in one kernel of a hipcc -S dump (kernel tuning aid).
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s; python tools/asm_stats.py x.s <name-substring>"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and sub in l]
for st in start:
    end = next(i for i in range(st, len(lines)) if '.end_amdhsa_kernel' in lines[i] or (i > st and re.match(r'^_Z\S*:', lines[i])))
    body = lines[st:end]
    print(lines[st].split(':')[0][:60], 'lines', len(body))
    c = Counter()
    for l in body:
        m = re.search(r's_waitcnt (.*)', l)
        if m and 'vmcnt' in m.group(1):
            c[re.sub(r'\s*;.*', '', m.group(1)).strip()] += 1
    print('  vmcnt waits:', c.most_common(16))
    def n(pat):
        return sum(1 for l in body if re.search(pat, l))
    print('  scratch_load %d scratch_store %d  lds-dma %d  global_store %d  global_load %d  buffer_load %d  mfma %d  s_barrier %d  ds_read %d' % (
        n(r'scratch_load'), n(r'scratch_store'), n(r'buffer_load.* lds'), n(r'global_store'), n(r'global_load'),
        n(r'buffer_load'), n(r'v_mfma'), n(r's_barrier'), n(r'ds_read')))
