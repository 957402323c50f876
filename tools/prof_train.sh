#!/bin/bash
# kernel-trace stats of the training update: bash tools/prof_train.sh <name> [kernel-name pattern for a per-grid table]
set -e
name=${1:-ptrain}
root=$(pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
export DRM_BENCH_NO_ACTING=1   # the window at the end of the trace must hold training updates only
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out" -o "$name" --output-format csv -- \
    python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --steps 2 --warmup 1 --profile-steps 0 --train-steps 30 > "$out/bench.json" 2> "$out/err.txt"
python3 "$root/tools/prof_window.py" "$out" 0.02 110 > "$out/stats.txt"
if [ -n "$2" ]; then python3 "$root/tools/prof_bygrid.py" "$out" 0.02 "$2" > "$out/bygrid.txt"; fi
find "$out" -name '*kernel_trace.csv' -delete
cat "$out/stats.txt"
