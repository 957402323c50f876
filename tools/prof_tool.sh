#!/bin/bash
# kernel-trace stats of one diagnostic script: bash tools/prof_tool.sh <name> <script.py> [pattern]
set -e
name=$1; script=$2
root=$(pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out" -o "$name" --output-format csv -- python3 "$root/$script" > "$out/out.txt" 2> "$out/err.txt"
python3 "$root/tools/prof_bygrid.py" "$out" 1.0 "${3:-}" > "$out/bygrid.txt"
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/bygrid.txt"
