#!/bin/bash
# Builds the round-5 store-dissection variants of libdurf_hip.so (tools/experiments/store_overlap.patch applied to a scratch
# copy of csrc/; only mlp_fwd.hip / mlp_bwd.hip are recompiled, the other objects come from durf_amd/csrc/build -- run make first).
#   tools/experiments/store_overlap_build.sh            -> durf_amd/variants/libdurf_{xbase,xnost,xscr,xscrnt,xplain,xsc1,xsc01,xntsc1,xburst}.so
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/durf_amd $tmp/include $root/durf_amd/variants
cp -r $root/durf_amd/csrc $tmp/durf_amd/csrc; rm -rf $tmp/durf_amd/csrc/build
cp $root/include/durf_hip.h $tmp/include/
(cd $tmp && patch -p0 -s < $root/tools/experiments/store_overlap.patch)
build() {
  name=$1; shift
  out=$tmp/build_$name; mkdir -p $out
  for f in mlp_fwd mlp_bwd; do
    (cd $tmp/durf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $f.hip -o $out/$f.o) &
  done
  wait
  objs=""
  for f in $(sed -n "s/^SRCS = //p" $root/durf_amd/csrc/Makefile | sed "s/\.hip//g"); do
    case $f in mlp_fwd|mlp_bwd) ;; *) objs="$objs $root/durf_amd/csrc/build/$f.o";; esac
  done
  hipcc --offload-arch=gfx950 -shared -fPIC $out/mlp_fwd.o $out/mlp_bwd.o $objs -ldl -o $root/durf_amd/variants/libdurf_$name.so
  echo built $name
}
build xbase
build xnost -DX_NOSTORE
build xscr -DX_SCRATCH -DX_PLAIN
build xscrnt -DX_SCRATCH
build xplain -DX_PLAIN
build xsc1 "-DX_ASM_STORE=sc1"
build xsc01 "-DX_ASM_STORE=sc0 sc1"
build xntsc1 "-DX_ASM_STORE=sc1 nt"
build xburst -DX_BURST
rm -rf $tmp
