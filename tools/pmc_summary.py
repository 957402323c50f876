"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: dispatches, mean counter value per dispatch.

    python tools/pmc_summary.py <dir> [counter ...]
"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
want = set(sys.argv[2:])
files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
if not files:
    sys.exit("no counter_collection.csv under " + d)
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        c = r["Counter_Name"]
        if want and c not in want:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        a = acc[(name[:90], c)]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
print("%-90s %-14s %9s %16s %16s" % ("kernel", "counter", "dispatches", "mean/dispatch", "total"))
for (name, c), (n, tot) in rows[:40]:
    print("%-90s %-14s %9d %16.1f %16.1f" % (name, c, n, tot / n, tot))
