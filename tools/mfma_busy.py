"""MFMA-pipe busy fraction per kernel from the round's rocprofv3 passes:
SQ_VALU_MFMA_BUSY_CYCLES per dispatch (prof_<tag>/<workload>.SQ_counters.txt; cycles summed over the SIMDs) divided by
(average duration per dispatch (prof_<tag>/<workload>.stats.txt) x 2.4 GHz x 1024 SIMDs).
Usage: python tools/mfma_busy.py gpurun_out/prof_r3 imagine4096 imagine4096_split"""
import re, sys

d = sys.argv[1]
print("MFMA-pipe busy fraction of the large-row kernels at B = 4096 (one chain), from the rocprofv3 passes of this directory:")
print("busy = SQ_VALU_MFMA_BUSY_CYCLES per dispatch (<workload>.SQ_counters.txt; counts cycles, summed over the SIMDs)")
print("       / (average duration per dispatch (<workload>.stats.txt) x 2.4 GHz x 1024 SIMDs)\n")
for wl in sys.argv[2:]:
    busy = {}
    for line in open(f"{d}/{wl}.SQ_counters.txt"):
        m = re.match(r"(\S.*?)\s+SQ_VALU_MFMA_BUSY_CYCLES\s+(\d+)\s+([\d.]+)", line)
        if m:
            busy[m.group(1).strip()] = float(m.group(3))
    for line in open(f"{d}/{wl}.stats.txt"):
        m = re.match(r"\s*[\d.]+ ms\s+[\d.]+%\s+(\d+) calls\s+([\d.]+) us\s+(?:void )?(.*?)\(", line)
        if not m:
            continue
        name, us = m.group(3).strip(), float(m.group(2))
        if name in busy:
            frac = busy[name] / (us * 1e-6 * 2.4e9 * 1024)
            print(f"{wl:18s} {name:48s} {busy[name] / 1e6:8.1f} M busy cycles / dispatch, {us:7.1f} us -> busy {frac:.2f}")
