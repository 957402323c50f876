"""Diagnostic: gaps between consecutive kernels in the last part of a rocprofv3 kernel trace (where the device waits)."""
import csv, glob, sys
d, frac = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))))
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= cut]
busy = sum(e - s for s, e, _ in rows)
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)]
tot_gap = sum(max(g[0], 0) for g in gaps)
print("window %.2f ms, kernels %d, busy %.2f ms, gaps %.2f ms" % ((t1 - cut) / 1e6, len(rows), busy / 1e6, tot_gap / 1e6))
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 10**9)):
    sel = [g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
    print("gaps %4d-%-6s us: %6d, total %.2f ms" % (lo, hi if hi < 10**9 else "inf", len(sel), sum(sel) / 1e6))
print("largest gaps:")
for g in sorted(gaps, reverse=True)[:12]:
    print("  %8.1f us after %s before %s" % (g[0] / 1e3, g[1], g[2]))
