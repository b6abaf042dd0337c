#!/bin/bash
# the inside of ONE stage of the object items (variant stamps2 = -DDURF_MS_STAMPS -DMS_STAMP_STAGE=2), mixed and stand-alone
out=gpurun_out/r06k; mkdir -p $out
export DURF_LIB_PATH=durf_amd/variants/libdurf_stamps2.so MS_STAMP_STAGE=2
{ echo "== mixed"; timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
  echo "== stand-alone (DURF_OBJ_MIX=0)"; DURF_OBJ_MIX=0 timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
} > $out/stage2.txt 2>&1
grep -E "==|stage 2:|item:|stage 2 " $out/stage2.txt
