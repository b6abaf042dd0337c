#!/bin/bash
# same-box interleaved A/B of library variants on the small steps (tools/build_variant.sh):  r06_mix_ab.sh <tag> v1 v2 ...
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out
one() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f rays/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = main ]; then unset DURF_LIB_PATH; else export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so; fi
    echo "$v 512: $(one --rays 512)   cfg5: $(one --config cfg5)   cfg2@512: $(one --config cfg2 --rays 512)"
  done
done > $out/ab_variants.txt 2>&1
