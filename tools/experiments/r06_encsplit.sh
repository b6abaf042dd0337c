#!/bin/bash
# the object items of the M-split kernels -- encode dealt to the four waves (MS_ENC_SPLIT), B fragments through an LDS read ring
# (MS_RING): bit-identity tests, item timeline (variant `stamps`), same-box A/B against variants built with the switches off
out=gpurun_out/r06l; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_fused_encode.py tests/test_gpu_mix.py tests/test_gpu_dispatch_matrix.py tests/test_gpu_stages.py -x -q > $out/tests.txt 2>&1; tail -3 $out/tests.txt
if [ -f durf_amd/variants/libdurf_stamps.so ]; then
( export DURF_LIB_PATH=durf_amd/variants/libdurf_stamps.so
  { echo "== cfg3 @ 512 rays, mixed"; timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
    echo "== cfg3 @ 512 rays, stand-alone object launches (DURF_OBJ_MIX=0)"; DURF_OBJ_MIX=0 timeout 200 python tools/experiments/ms_stamps.py --config cfg3 --rays 512
  } > $out/stamps.txt 2>&1 )
fi
bash tools/experiments/r06_mix_ab.sh r06l "$@"
cat $out/ab_variants.txt
