#!/bin/bash
# round 6: obj_precision = 'bf16x3' -- gates, then cfg4 exact vs split operands, interleaved:  gpurun_out/<tag>/
tag=${1:-r06_x3}; out=gpurun_out/$tag; mkdir -p $out
(timeout 600 python -m pytest tests/test_gpu_x3.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $out/pytest_x3.txt
one() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f rays/s  %.4f ms  loss %.6f' % (d['value'], d['ms_per_step'], d['loss']))"; }
for i in 1 2 3; do
  echo "cfg4 exact fp32 objects: $(one --config cfg4)"; echo "cfg4 bf16x3 objects:     $(one --config cfg4 --precision bf16x3)"
done > $out/ab_x3.txt 2>&1
timeout 300 bash tools/timeline.sh $tag/tlcfg4_x3 --config cfg4 --precision bf16x3
timeout 300 bash tools/timeline.sh $tag/tlcfg4 --config cfg4
