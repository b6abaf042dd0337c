#!/bin/bash
# review item 7: the fused per-ray launch at 4096 and 32 768 rays, with rocprofv3 durations and PMC bytes -> gpurun_out/composite_32k.txt
out=$(pwd)/gpurun_out/composite_32k.txt; root=$(pwd)
python3 tools/time_composite.py 4096 32768 > $out 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c_st /tmp/c_f /tmp/c_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c_st -- python3 $root/tools/time_composite.py 32768 > /dev/null 2>&1
echo "--- rocprofv3 --kernel-trace --stats -- python3 tools/time_composite.py 32768" >> $out
python3 $root/tools/summarize_rocprof.py /tmp/c_st | grep -E "composite|reduce_rows|kernel " >> $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/c_p
  rocprofv3 --pmc $c --output-format csv -d /tmp/c_p -- python3 $root/tools/time_composite.py 32768 > /dev/null 2>&1
  echo "--- rocprofv3 --pmc $c (KiB per dispatch; FETCH_SIZE x 2 on gfx950 for streaming reads)" >> $out
  python3 $root/tools/summarize_rocprof.py /tmp/c_p | grep -E "composite" >> $out
done
