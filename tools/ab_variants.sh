#!/bin/bash
# Interleaved same-box A/B of variant libraries (tools/build_variant.sh):  tools/ab_variants.sh <log> <rounds> <cmd...> -- v1 v2 ...
# "" (written as main) = the shipped durf_amd/libdurf_hip.so
log=$1; rounds=$2; shift 2
cmd=()
while [ "$1" != "--" ]; do cmd+=("$1"); shift; done
shift
for rep in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset DURF_LIB_PATH; else export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so; fi
    echo "variant=$v" >> $log
    "${cmd[@]}" 2>&1 | tail -${TAIL:-1} >> $log
  done
done
