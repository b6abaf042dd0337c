python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -s -k "reference_batch_and_128" 2>&1 | grep -E "rel err|passed|failed|Error|assert" | head -20
python -m pytest tests/test_golden_ref_model.py -q -m gpu -s 2>&1 | grep -E "max abs|measured|passed|failed" | head -40
for rep in 1 2; do
for m in 0 2 3 1; do
  for c in "cfg3 --rays 512" "cfg3 --rays 1024" "cfg5" "cfg1"; do
    echo -n "overlap=$m $c: "; DURF_OVERLAP_OBJECTS=$m python bench.py --config $c --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,1), 'k', round(d['ms_per_step'],4))"
  done
done
done
