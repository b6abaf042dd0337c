# workgroups per object of the M-split object launches (DURF_MS_CAP; default 384 / K clamped to [16, 128])
for c in "cfg5" "cfg3 --rays 512" "cfg3 --rays 1024" "cfg5 --rays 512"; do
  for cap in 0 128 64 48 32 24 16; do
    echo -n "MS_CAP=$cap $c: "
    DURF_MS_CAP=$cap python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,1), round(d['ms_per_step'],4))"
  done
done
