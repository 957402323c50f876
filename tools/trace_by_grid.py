"""Diagnostic: group a rocprofv3 kernel-trace CSV by (kernel, grid) and print count / mean duration / share."""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows[(name[:60], r["Grid_Size_X"], r["Grid_Size_Y"])].append(d)
tot = sum(sum(v) for v in rows.values())
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("%-60s grid=%8s x %4s n=%5d mean=%8.1f us min=%8.1f share=%5.1f%%" % (k[0], k[1], k[2], len(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
if len(sys.argv) > 3:      # the last N dispatches in time order
    seq = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            if "drm::" not in r["Kernel_Name"]:
                continue
            seq.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("drm::", "")[:40], r["Grid_Size_X"]))
    seq.sort()
    ends = [i for i, q in enumerate(seq) if q[2].startswith("k_lambda_return")]      # the last imagination block
    if ends:
        seq = seq[: ends[-1] + 1]
    seq = seq[-int(sys.argv[3]):]
    for i, (s, e, n, g) in enumerate(seq):
        gap = (s - seq[i - 1][1]) / 1e3 if i else 0.0
        print("%8.1f us  gap %6.1f  %-48s %s" % ((e - s) / 1e3, gap, n, g))
