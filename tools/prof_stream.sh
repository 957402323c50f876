#!/bin/bash
# kernel trace and memory-side byte counters of the three stream kernels at device-filling sizes:
#   bash tools/prof_stream.sh        (results under gpurun_out/pstream)
set -e
root=$(pwd)
out=$root/gpurun_out/pstream
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/trace" -o stream --output-format csv -- \
    python3 "$root/tools/bench_stream.py" "$out/bench.json" --large-only > "$out/bench.txt" 2> "$out/err.txt"
python3 "$root/tools/prof_summary.py" "$out/trace" 12 > "$out/stats.txt"
find "$out" -name '*kernel_trace.csv' -delete
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c -d "$out/$c" -o stream --output-format csv -- \
        python3 "$root/tools/bench_stream.py" --large-only > /dev/null 2>> "$out/err.txt"
    python3 "$root/tools/pmc_summary.py" "$out/$c" $c | grep -E "kernel|drm::" > "$out/$c.txt"
    rm -rf "$out/$c"
done
cat "$out/bench.txt" "$out/stats.txt" "$out/FETCH_SIZE.txt" "$out/WRITE_SIZE.txt"
