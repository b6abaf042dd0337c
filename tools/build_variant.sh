#!/bin/bash
# Build an A/B variant of libdurf_hip.so with extra -D flags, for kernel tuning:
#   tools/build_variant.sh nt "-DDW_DMA_AUX=2"   ->  durf_amd/variants/libdurf_nt.so
# Run with DURF_LIB_PATH=durf_amd/variants/libdurf_nt.so python tools/bench_mlp_train.py
set -e
cd "$(dirname "$0")/../durf_amd/csrc"
name=$1; shift
SRCS=$(sed -n "s/^SRCS = //p" Makefile | sed "s/\.hip//g")
out=../variants/build_$name
mkdir -p $out
for f in $SRCS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o $out/$f.o &
done
wait; for f in $SRCS; do test -f $out/$f.o || { echo "compile of $f failed"; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC $out/*.o -o ../variants/libdurf_$name.so
rm -rf $out
echo built durf_amd/variants/libdurf_$name.so
