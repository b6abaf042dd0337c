#!/bin/bash
# Effective shader clock of the fused kernels per store variant: GRBM_GUI_ACTIVE (cycles) and the dispatch's duration from ONE
# profiled run (rocprofv3 --kernel-trace --pmc), plus the pure store ceilings of the board (tools/probes/probe_hbm).
out=${1:-gpurun_out/clock_under_stores}; shift
root=$(pwd); mkdir -p $out
(cd tools/probes && hipcc --offload-arch=gfx950 -O3 -w probe_hbm.hip -o probe_hbm 2>/dev/null; ./probe_hbm) > $out/probe_hbm.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in ${@:-xbase xnost xscr xscrnt}; do
  export DURF_LIB_PATH=$root/durf_amd/variants/libdurf_$v.so
  rm -rf /tmp/cus_$v
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/cus_$v -- python3 $root/tools/experiments/bench_fwd_only.py > /dev/null 2>/tmp/cus_$v.err
  python3 $root/tools/experiments/clock_from_trace.py /tmp/cus_$v > $root/$out/clock_$v.txt 2>&1
done
