#!/usr/bin/env python3
"""profiles/<tag>_rocprofv3_{fetch,write}.txt (tools/collect_profiles.sh) -> <tag>_pmc_traffic.json: HBM bytes per
launch of the kernels bench.py's roofline reports, corrected as MI355X_MICROARCH.md prescribes (counter unit KiB;
FETCH_SIZE x2 on gfx950 for wide streaming reads; WRITE_SIZE as reported).  bench.py quotes the file only when its
lib_version / workload / rays_per_gpu match the running build.
    make_pmc_traffic.py <dir> <tag> <workload> <rays_per_gpu> <lib_version>"""
import json
import os
import re
import sys

KERNELS = {'mlp_fwd_256_train': r'k_mlp_fwd<256, true[,>]', 'mlp_bwd_256': r'k_mlp_bwdILi256|k_mlp_bwd<256',
           'mlp_dw_256': r'k_dw_all<256>', 'encode_bkgd': r'k_encode_lane<false>|k_encode_oct<false>',
           'composite_resample': r'k_composite_resample'}


def read(path, counter, scale=1024.0):
    out = {}
    for ln in open(path):
        m = re.search(counter + r'=([0-9.e+]+) \(n=(\d+)\)', ln)
        if not m:
            continue
        for name, pat in KERNELS.items():
            if re.search(pat, ln):
                out[name] = float(m.group(1)) * scale
    return out


def main():
    d, tag, workload, rays, ver = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    fetch = read(os.path.join(d, 'rocprofv3_fetch.txt'), 'FETCH_SIZE')
    write = read(os.path.join(d, 'rocprofv3_write.txt'), 'WRITE_SIZE')
    mfma_path = os.path.join(d, 'rocprofv3_mfma.txt')
    busy = read(mfma_path, 'SQ_VALU_MFMA_BUSY_CYCLES', scale=1.0) if os.path.exists(mfma_path) else {}
    out = dict(source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/%s_rocprofv3_{fetch,write}.txt); KiB -> '
                      'bytes, FETCH_SIZE x2 per the gfx950 correction; bytes per launch (the dW launch covers both '
                      'levels); mfma_busy_cycles = SQ_VALU_MFMA_BUSY_CYCLES of the %s_rocprofv3_mfma.txt pass' % (tag, tag),
               lib_version=ver, workload=workload, rays_per_gpu=rays)
    for k in KERNELS:
        if k in fetch and k in write:
            out[k] = dict(fetch_bytes=2 * fetch[k], write_bytes=write[k], total_bytes=2 * fetch[k] + write[k])
            if k in busy:       # SQ_VALU_MFMA_BUSY_CYCLES per launch (cycles; / 32 x 32 768 FLOP = executed bf16 MFMA work)
                out[k]['mfma_busy_cycles'] = busy[k]
    json.dump(out, open(os.path.join(d, 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
