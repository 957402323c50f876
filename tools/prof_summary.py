"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV directory."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:n]:
    print("%9.2f ms %6.2f%% %7d calls %9.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6, 100.0 * int(r["TotalDurationNs"]) / tot,
                                                    int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:120]))
