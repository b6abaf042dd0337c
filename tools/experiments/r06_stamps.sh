#!/bin/bash
# in-kernel timeline of the object items (tools/experiments/ms_stamps.py), mixed launch and stand-alone launches, for the probe
# variants given (default: stamps):  stampsns = without the items' stash / mask stores, stampsow = one weight fragment per stage
# (timing probes, wrong results); build:  tools/build_variant.sh stampsns -DDURF_MS_STAMPS -DMS_PROBE_NOSTORE=1   (one -D per argument)
out=gpurun_out/r06k; mkdir -p $out
[ $# -eq 0 ] && set -- stamps
for rep in 1 2; do
for v in "$@"; do
  export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so
  { echo "== $v: cfg3 @ 512 rays, mixed"; timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
    echo "== $v: cfg3 @ 512 rays, stand-alone object launches (DURF_OBJ_MIX=0)"; DURF_OBJ_MIX=0 timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
  } > $out/$v.$rep.txt 2>&1
done
done
grep -E "==|item:" $out/*.[12].txt
