#!/bin/bash
# End-of-milestone evidence run (on the GPU box):  tools/collect_profiles.sh <tag> [config]
#   gpurun_out/<tag>/bench.json            default bench.py line (with cpu_baseline)
#   gpurun_out/<tag>/bench_op_times.txt    per-op table of a separate --profile-ops run
#   gpurun_out/<tag>/rocprofv3_stats.txt   rocprofv3 --kernel-trace --stats of bench.py --steps 10
#   gpurun_out/<tag>/rocprofv3_{fetch,write,mfma}.txt   separate --pmc passes (bench.py --steps 2)
#   gpurun_out/<tag>/pmc_traffic.json      bytes per launch derived from them (tools/make_pmc_traffic.py)
tag=${1:-rXX}
cfg=${2:-cfg3}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
python3 bench.py --config $cfg > $out/bench.json 2> $out/bench.err      # the default line, exactly as the driver runs it
python3 bench.py --config $cfg --steps 50 --profile-ops --no-cpu-baseline --no-workloads > $out/bench_all_ops_timed.json 2> $out/bench_op_times.txt   # full per-op table (every op timed)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 /root/repo/bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-calibration --no-workloads > $out/bench_under_rocprof.json 2>/dev/null
( echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-calibration   (MI355X)"; python3 /root/repo/tools/summarize_rocprof.py /tmp/p_stats ) > $out/rocprofv3_stats.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  name=$(echo $c | cut -d_ -f1 | tr 'A-Z' 'a-z'); [ "$name" = "sq" ] && name=mfma
  rm -rf /tmp/p_$name
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d /tmp/p_$name -- python3 /root/repo/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-workloads > /dev/null 2>&1
  ( echo "# rocprofv3 --kernel-trace --output-format csv --pmc $c -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-calibration"; python3 /root/repo/tools/summarize_rocprof.py /tmp/p_$name | grep -v "kernel stats\|^kernel \|^[a-zA-Z_:<>0-9 ,()*&\[\].~-]* [0-9]* *[0-9.]* *[0-9.]* *[0-9.]*$" ) > $out/rocprofv3_$name.txt
done
cd /root/repo
rays=$(python3 -c "import bench; print(bench.WORKLOADS['$cfg'][3])")
ver=$(python3 -c "from durf_amd import _lib; print(_lib.lib().durf_version())")
python3 tools/make_pmc_traffic.py $out $tag $cfg $rays $ver > /dev/null
