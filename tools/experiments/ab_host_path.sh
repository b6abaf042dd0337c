#!/bin/bash
# bench.py through durf_train_step (default) against bench.py --python-step (train_step's Python-issued launches), every
# workload, interleaved, same box:  tools/experiments/ab_host_path.sh  ->  gpurun_out/ab_host_path.txt
cd /root/repo
out=gpurun_out/ab_host_path.txt
ver=$(python3 -c "from durf_amd import _lib; print(_lib.lib().durf_version())")
echo "# bench.py <workload> --no-workloads --no-cpu-baseline [--python-step], interleaved, one box (MI355X), library version $ver: k rays/s (ms/step)" > $out
for w in "--config cfg3" "--config cfg3 --rays 512" "--config cfg1" "--config cfg2" "--config cfg4" "--config cfg5"; do
  for rep in 1 2 3; do
    for m in "" "--python-step"; do
      python3 bench.py $w --no-workloads --no-cpu-baseline $m 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-28s %-14s %8.1f k  (%.4f ms)  vendor GEMM %4.0f TFLOP/s' % ('$w', d['config']['host_path'][:12], d['value'] / 1e3, d['ms_per_step'], d['roofline']['board']['vendor_gemm_tflops']))" >> $out
    done
  done
done
