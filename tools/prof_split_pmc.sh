#!/bin/bash
# counters of the large-row split kernels over the B = 4096 block, one rocprofv3 pass per counter group:
# bash tools/prof_split_pmc.sh <name>
set -e
name=${1:-split_pmc}
root=$(pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
export DRM_NO_TWO_CHAINS=1
pass() {  # tag, counters
    local tag=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc $* -d "$out/$tag" -o blk --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-stream-kernels --no-extras --train-steps 0 --profile-steps 0 --workload imagine4096_split --steps 3 --warmup 1 > "$out/$tag.out" 2> "$out/$tag.err"
    python3 "$root/tools/pmc_summary.py" "$out/$tag" $* | grep -E "kernel|k_linear_big|k_table_add" > "$out/$tag.txt"
}
pass sq1 SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES && \
pass sq2 SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
find "$out" -name '*counter_collection.csv' -delete
cat "$out"/*.txt
