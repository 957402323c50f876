#!/bin/bash
# Copies the summaries of gpurun_out/prof_<tag> (written by tools/profile_round.sh) into profiles/<tag>_*.
set -e
tag=${1:-r1}
src=gpurun_out/prof_$tag
mkdir -p profiles
for f in "$src"/*.stats.txt "$src"/*_SIZE.txt "$src"/*.SQ_counters.txt "$src"/*.bench.json; do cp "$f" "profiles/${tag}_$(basename "$f")"; done
for d in "$src"/*/; do n=$(basename "$d"); cp "$d/${n}_kernel_stats.csv" "profiles/${tag}_${n}_kernel_stats.csv"; done
cp "$src/pmc.json" "profiles/${tag}_pmc.json"
ls profiles
