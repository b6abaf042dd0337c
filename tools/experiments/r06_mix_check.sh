#!/bin/bash
# round 6: the mixed launches -- bit-identity tests, step timelines and an interleaved A/B (DURF_OBJ_MIX=1 / 0), every command
# under its own timeout:  gpurun -- 'bash tools/experiments/r06_mix_check.sh <tag>'   -> gpurun_out/<tag>/
tag=${1:-r06_mix}; out=gpurun_out/$tag; mkdir -p $out
(timeout 420 python -m pytest tests/test_gpu_mix.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -40) > $out/pytest_mix.txt
(timeout 900 python -m pytest tests/test_gpu_dispatch_matrix.py tests/test_gpu_train_call.py tests/test_gpu_fused_encode.py tests/test_gpu_forward_call.py -x -q -m gpu 2>&1 | grep -v "^  File\|^Extension" | tail -30) > $out/pytest_b.txt
timeout 300 bash tools/timeline.sh $tag/tl512 --rays 512
timeout 300 bash tools/timeline.sh $tag/tlcfg5 --config cfg5
one() { timeout 120 python bench.py "$@" --no-cpu-baseline --no-calibration --no-workloads 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f rays/s  %.4f ms' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  echo "mix   512 rays: $(one --rays 512)";  echo "nomix 512 rays: $(DURF_OBJ_MIX=0 one --rays 512)"
  echo "mix   cfg5:     $(one --config cfg5)"; echo "nomix cfg5:     $(DURF_OBJ_MIX=0 one --config cfg5)"
  echo "mix   cfg2@512: $(one --config cfg2 --rays 512)"; echo "nomix cfg2@512: $(DURF_OBJ_MIX=0 one --config cfg2 --rays 512)"
done > $out/ab.txt 2>&1
timeout 200 python bench.py --mode eval --image-call > $out/eval_image.json 2> $out/eval_image.err
timeout 200 python bench.py --mode eval --one-call > $out/eval_onecall.json 2> $out/eval_onecall.err
