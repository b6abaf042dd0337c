cd /root/repo
for rep in 1 2 3; do
  for v in main noroll; do
    if [ "$v" = main ]; then unset DURF_LIB_PATH; else export DURF_LIB_PATH=durf_amd/variants/libdurf_$v.so; fi
    echo -n "variant=$v  "; python tools/time_fwd.py 2>&1 | tail -1
    python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-calibration 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('   bench %.1f k rays/s  %.3f ms/step  fwd %.0f us  bwd %.0f us  dW %.0f us  non-MLP %.3f ms' % (d['value']/1e3, d['ms_per_step'], r['all']['mlp_fwd_256_train']['us'], r['all']['mlp_bwd_256']['us'], r['all']['mlp_dw_256']['us'], r['non_mlp_ms_per_step']))"
  done
done
