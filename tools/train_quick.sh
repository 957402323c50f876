#!/bin/bash
# training-update time only (no profiler): bash tools/train_quick.sh <name>
name=${1:-tq}
export DRM_BENCH_NO_ACTING=1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stream-kernels --no-extras --steps 2 --warmup 1 --profile-steps 0 --train-steps 40 > gpurun_out/$name.json 2> gpurun_out/$name.err
rc=$?
python3 - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}.json"))
t = d.get("train") or {}
print({k: t.get(k) for k in ("ms_per_update", "host_ms_per_update", "updates_per_sec")})
PY
exit $rc
