# bench.py with the roofline HIP events on every step (--time-every 1, the behaviour up to round 3) against the sampled default
for rep in 1 2; do
for c in "cfg3" "cfg3 --rays 512" "cfg5" "cfg1" "cfg4" "cfg2"; do
  for e in 1 0; do
    echo -n "time-every=$e  $c: "
    python bench.py --config $c --no-cpu-baseline --time-every $e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.1f k rays/s  %.4f ms/step  dW %.1f us  fwd %.1f us  bwd %.1f us  non-MLP %.3f ms  timed steps %s' % (d['value']/1e3, d['ms_per_step'], r['all']['mlp_dw_256']['us'], r['all']['mlp_fwd_256_train']['us'], r['all']['mlp_bwd_256']['us'], r['non_mlp_ms_per_step'], r.get('timed_steps')))"
  done
done
done
