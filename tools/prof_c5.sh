#!/bin/bash
# kernel-trace stats of the scaled model's world-model loss + backward: bash tools/prof_c5.sh <name> <wm|wm_bf16>
set -e
name=$1; which=$2
root=$(pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
export C5_ONLY=$which
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out" -o "$name" --output-format csv -- \
    python3 "$root/tools/bench_c5.py" > "$out/bench.txt" 2> "$out/err.txt"
python3 "$root/tools/prof_summary.py" "$out" 40 > "$out/stats.txt"
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/stats.txt"
