#!/bin/bash
# The default bench under runtime settings that change how kernel arguments and recorded graphs reach the GPU.
set -e
run() {
    name=$1; shift
    env "$@" timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --no-stream-kernels --train-steps 0 \
        > gpurun_out/env_$name.json 2> gpurun_out/env_$name.err
    echo "$name: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/env_$name.json)"
}
run base DRM_X=0
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
