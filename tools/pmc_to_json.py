"""Collect the per-kernel FETCH_SIZE / WRITE_SIZE tables written by tools/profile_round.sh into one JSON:
    python tools/pmc_to_json.py gpurun_out/prof_r1 profiles/r1_pmc.json
-> {workload: {kernel: {counter: {"dispatches": n, "total": KiB}}}}"""
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {}
for path in sorted(glob.glob(os.path.join(src, "*.*_SIZE.txt"))):
    workload, counter = os.path.basename(path)[:-4].split(".")
    for line in open(path).read().splitlines()[1:]:
        parts = line.split()
        if len(parts) < 5:
            continue
        total, _mean, n, ctr = float(parts[-1]), float(parts[-2]), int(parts[-3]), parts[-4]
        kernel = " ".join(parts[:-4])
        out.setdefault(workload, {}).setdefault(kernel, {})[ctr] = {"dispatches": n, "total": total}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, {w: len(k) for w, k in out.items()})
