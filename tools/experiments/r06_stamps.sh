#!/bin/bash
# in-kernel timeline of the object items (tools/experiments/ms_stamps.py) -- mixed launch and stand-alone launches;
# `stampsns`: the same without the items' stash / mask stores (timing probe, wrong results)
out=gpurun_out/r06k; mkdir -p $out
for v in stamps stampsns; do
  export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so
  { echo "== $v: cfg3 @ 512 rays, mixed"; timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
    echo "== $v: cfg3 @ 512 rays, stand-alone object launches (DURF_OBJ_MIX=0)"; DURF_OBJ_MIX=0 timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
  } > $out/$v.txt 2>&1
done
grep -E "==|item:" $out/stamps.txt $out/stampsns.txt
