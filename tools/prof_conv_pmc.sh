#!/bin/bash
# counters of the tower convolution kernels at the training shapes, one rocprofv3 pass per counter group:
# bash tools/prof_conv_pmc.sh <name>
set -e
name=${1:-conv_pmc}
root=$(pwd)
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
export CONV_TIMES_NO_BACKEND=1
pass() {  # tag, counters
    local tag=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc $* -d "$out/$tag" -o conv --output-format csv -- python3 "$root/tools/conv_native_times.py" > "$out/$tag.out" 2> "$out/$tag.err"
    python3 "$root/tools/pmc_summary.py" "$out/$tag" $* | grep -E "kernel|k_conv" > "$out/$tag.txt"
}
pass sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS && \
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM && \
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
find "$out" -name '*counter_collection.csv' -delete
cat "$out"/*.txt
