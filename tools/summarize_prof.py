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

# machine-readable digest for bench.py's roofline.traffic (HBM bytes per launch of the dominant kernel)
import json
digest = {}
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "fft1024" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for c, v in acc.items():
        digest[c] = sum(v) / len(v)
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "fft1024" in row["Name"]:
                digest["kernel_avg_ns"] = float(row["AverageNs"])
                digest["kernel_calls"] = int(row["Calls"])
                digest["kernel_min_ns"] = float(row["MinNs"])
                digest["kernel_max_ns"] = float(row["MaxNs"])
# the timed region of bench.py = the LAST 50 launches of the run (--steps 50); the stats row above also averages
# the 400 clock-ramp launches and the 5 warm-up launches
for f in find("*kernel_trace.csv"):
    rows = []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "fft1024" in row["Kernel_Name"]:
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    rows.sort()
    last = [d for _, d in rows[-50:]]
    if last:
        digest["kernel_timed_region_avg_ns"] = sum(last) / len(last)
        digest["kernel_timed_region_launches"] = len(last)
        print("timed region (last %d launches): avg %.1f us, min %.1f, max %.1f" % (len(last), sum(last) / len(last) / 1e3, min(last) / 1e3, max(last) / 1e3))
if "FETCH_SIZE" in digest and "WRITE_SIZE" in digest:
    # FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE counts the 128-B requests of a coalesced
    # stream as 64 B -> double it (MI355X_MICROARCH.md, HBM section); cross-check: TCC_EA0_RDREQ_sum x 128 B.
    digest["hbm_read_bytes"] = digest["FETCH_SIZE"] * 1024 * 2
    digest["hbm_write_bytes"] = digest["WRITE_SIZE"] * 1024
    digest["hbm_bytes_per_launch"] = digest["hbm_read_bytes"] + digest["hbm_write_bytes"]
    if "TCC_EA0_RDREQ_sum" in digest:
        digest["crosscheck_rdreq_x128B"] = digest["TCC_EA0_RDREQ_sum"] * 128
        digest["crosscheck_wrreq_x64B"] = digest.get("TCC_EA0_WRREQ_sum", 0) * 64
with open(os.path.join(root, "digest.json"), "w") as fh:
    json.dump(digest, fh, indent=1, sort_keys=True)
print("digest:", json.dumps(digest, sort_keys=True))
