#!/bin/bash
# the object encode dealt to the four waves of an M-split item (MS_ENC_SPLIT): bit-identity tests, item timeline, same-box A/B
out=gpurun_out/r06l; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fused_encode.py tests/test_gpu_mix.py tests/test_gpu_dispatch_matrix.py tests/test_gpu_stages.py -x -q > $out/tests.txt 2>&1; tail -3 $out/tests.txt
( export DURF_LIB_PATH=durf_amd/variants/libdurf_stamps.so
  { echo "== cfg3 @ 512 rays, mixed"; timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
    echo "== cfg3 @ 512 rays, stand-alone object launches (DURF_OBJ_MIX=0)"; DURF_OBJ_MIX=0 timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
  } > $out/stamps.txt 2>&1 )
bash tools/experiments/r06_mix_ab.sh r06l main ring0 ring6 ring8 nosplit
cat $out/ab_variants.txt
