#!/bin/bash
# in-kernel timeline of the BACKWARD object items (variant stamps), mixed and stand-alone
out=gpurun_out/r06k; mkdir -p $out
export DURF_LIB_PATH=durf_amd/variants/libdurf_stamps.so
{ echo "== mixed"; timeout 200 python tools/experiments/ms_stamps.py --bwd --config cfg3 --rays 512
  echo "== stand-alone (DURF_OBJ_MIX=0)"; DURF_OBJ_MIX=0 timeout 200 python tools/experiments/ms_stamps.py --bwd --config cfg3 --rays 512
} > $out/bwd.txt 2>&1
grep -v amdgpu $out/bwd.txt
