#!/bin/bash
# kernel-trace stats of one bench workload: bash tools/prof_quick.sh <name> [bench args...]
set -e
name=$1; shift
root=$(pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o "$name" --output-format csv -- \
    python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --train-steps 0 --profile-steps 0 "$@" > "$out/bench.json" 2> "$out/err.txt"
python3 "$root/tools/prof_summary.py" "$out" 30 > "$out/stats.txt"
python3 "$root/tools/prof_bygrid.py" "$out" > "$out/bygrid.txt" 2>/dev/null || true
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/stats.txt"
