#!/bin/bash
# A/B of one environment switch on a bench workload, alternating runs: bash tools/ab_env.sh <VAR> <workload> [rounds]
var=$1; wl=${2:-imagine4096_split}; n=${3:-3}
for i in $(seq 1 $n); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    timeout -k 10 200 python -u bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --no-stream-kernels --no-extras --train-steps 0 --profile-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['ms_per_step'],3))" || exit 1
  done
done
