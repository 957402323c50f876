"""Aggregate a rocprofv3 kernel trace over its last FRACTION of wall time (steady state)."""
import csv, glob, sys, collections
d, frac = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= cut:
        a = agg[r["Kernel_Name"]]; a[0] += e - s; a[1] += 1
tot = sum(v[0] for v in agg.values())
print("window ms", (t1 - cut) / 1e6, "kernel ms in window", tot / 1e6, "launches", sum(v[1] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%9.2f ms %6.2f%% %7d calls %9.1f us  %s" % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], v[0] / v[1] / 1e3, k[:130]))
