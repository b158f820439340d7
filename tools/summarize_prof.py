"""Summarise a tools/profile.sh output directory: per-kernel duration stats + PMC counters per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage")})

# raw durations of the dominant kernel from the trace
for f in find("*kernel_trace.csv"):
    durs = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            durs[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in durs.items():
        v.sort()
        print("trace: %-60s n=%d min=%.1fus med=%.1fus max=%.1fus" % (k[:60], len(v), v[0] / 1e3, v[len(v) // 2] / 1e3, v[-1] / 1e3))

print("== PMC counters (mean per dispatch of the FFT kernel) ==")
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "fft1024" in row["Kernel_Name"] or "k_pass" in row["Kernel_Name"]:
                acc[(row["Kernel_Name"][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-42s %-24s n=%d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
