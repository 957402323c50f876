"""Per-(kernel, grid) average durations from a rocprofv3 kernel trace, over the last fraction of the run."""
import csv, glob, sys, collections
d, frac = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
pat = sys.argv[3] if len(sys.argv) > 3 else ""
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) < cut or pat not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size_X"], r["Grid_Size_Y"], r["VGPR_Count"], r["LDS_Block_Size"])
    a = agg[key]; a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(v[0] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.2f ms %5.1f%% %6d calls %9.1f us  %s grid %sx%s vgpr %s lds %s" % (v[0] / 1e6, 100.0 * v[0] / tot, v[1], v[0] / v[1] / 1e3, *k))
