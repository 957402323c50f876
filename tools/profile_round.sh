#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   kernel-trace stats of both bench workloads, then FETCH_SIZE and WRITE_SIZE passes (one pass each).
# Usage (on the box, from the repo root):  bash tools/profile_round.sh r1
set -e
tag=${1:-r1}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp

stats() {  # name, bench args...
    local name=$1; shift
    # kernel durations are read per kernel: the large-batch block runs as one chain of launches here (two side by side
    # would overlap kernels of the two chains in the trace); the .clean.bench.json line is the shipped configuration
    # the same command without the profiler first: its JSON line is the one to quote (a kernel trace slows short kernels)
    timeout -k 10 300 python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --no-extras --train-steps 0 "$@" > "$out/$name.clean.bench.json" 2> /dev/null
    DRM_NO_TWO_CHAINS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/$name" -o "$name" --output-format csv -- \
        python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --no-extras --train-steps 0 "$@" > "$out/$name.bench.json" 2> "$out/$name.err"
    python3 "$root/tools/prof_summary.py" "$out/$name" 30 > "$out/$name.stats.txt"
    find "$out/$name" -name '*kernel_trace.csv' -delete
}

pmc() {  # name, counter, bench args...
    local name=$1; local ctr=$2; shift; shift
    DRM_NO_TWO_CHAINS=1 timeout -s KILL 200 rocprofv3 --pmc "$ctr" -d "$out/$name.$ctr" -o "$name" --output-format csv -- \
        python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --no-extras --train-steps 0 --profile-steps 0 "$@" > /dev/null 2> "$out/$name.$ctr.err"
    python3 "$root/tools/pmc_summary.py" "$out/$name.$ctr" "$ctr" > "$out/$name.$ctr.txt"
    rm -rf "$out/$name.$ctr"
}

# several counters of one block in one pass (at most 8 SQ_ counters): name, file tag, "CTR1 CTR2", bench args...
pmc_multi() {
    local name=$1; local tag=$2; local ctrs=$3; shift; shift; shift
    DRM_NO_TWO_CHAINS=1 timeout -s KILL 200 rocprofv3 --pmc $ctrs -d "$out/$name.$tag" -o "$name" --output-format csv -- \
        python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --no-extras --train-steps 0 --profile-steps 0 "$@" > /dev/null 2> "$out/$name.$tag.err"
    python3 "$root/tools/pmc_summary.py" "$out/$name.$tag" $ctrs > "$out/$name.$tag.txt"
    rm -rf "$out/$name.$tag"
}

stats imagine16 --workload imagine16
stats imagine4096 --workload imagine4096 --steps 20 --warmup 3
pmc imagine4096 FETCH_SIZE --workload imagine4096 --steps 3 --warmup 1
pmc imagine4096 WRITE_SIZE --workload imagine4096 --steps 3 --warmup 1
pmc imagine16 FETCH_SIZE --workload imagine16 --steps 10 --warmup 2
pmc imagine16 WRITE_SIZE --workload imagine16 --steps 10 --warmup 2
stats imagine4096_bf16 --workload imagine4096_bf16 --steps 20 --warmup 3
stats imagine4096_split --workload imagine4096_split --steps 20 --warmup 3
pmc imagine4096_split FETCH_SIZE --workload imagine4096_split --steps 3 --warmup 1
pmc imagine4096_split WRITE_SIZE --workload imagine4096_split --steps 3 --warmup 1
pmc_multi imagine4096 SQ_counters "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" --workload imagine4096 --steps 3 --warmup 1
pmc_multi imagine4096_split SQ_counters "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" --workload imagine4096_split --steps 3 --warmup 1
python3 "$root/tools/pmc_to_json.py" "$out" "$out/pmc.json"
echo done
