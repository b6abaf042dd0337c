#!/bin/bash
# how often is a 512-ray bench run slow, and is it one stall or every step?  (step_ms: p50 / p90 / max / slow_steps)
out=gpurun_out/r06n; mkdir -p $out
for i in $(seq 1 70); do
  timeout 120 python bench.py --rays 512 --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f rays/s  %.4f ms  step_ms %s' % (d['value'], d['ms_per_step'], json.dumps(d['step_ms'])))"
done > $out/runs.txt 2>&1
sort -n $out/runs.txt | head -4; sort -n $out/runs.txt | tail -2
