#!/bin/bash
# tools/profile_configs.sh <tag> -- rocprofv3 --kernel-trace --stats over tools/bench_configs.py for the other BASELINE
# configurations and the new kernel families (run on the GPU box via gpurun); compact per-kernel CSV in gpurun_out/.
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_cfg_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $REPO/tools/bench_configs.py \
    C3 C4 C5 C5fwd C5inv C2inv C2r C2u 7:16:16:0 7:16:16:0:0:PAIR 7:16:16:1 11:16:16:0 13:16:16:0 14:16:16:0 13:16:16:0:0:INV 14:16:16:0:0:INV 13:16:16:0:0:PAIR 14:16:16:0:0:PAIR 15:16:16:0 16:16:16:0 16:16:16:0:0:INV 16:16:16:0:0:PAIR 12:16:16:1 16:16:16:1 16:16:16:1:0:INV 10:12:16:0 12:16:16:0:1 12:16:16:0:1:PAIR 10:16:16:0:1:INV 17:16:16:0 18:16:16:0 17:16:16:0:0:INV 18:16:16:0:0:INV 17:16:16:0:0:PAIR 17:16:16:0:1 18:16:16:0:1 20:16:16:0:1 20:16:16:0:1:INV 20:16:16:0:0:FWD:10 21:16:16:0:0:FWD:10 22:16:16:0:0:FWD:10 23:16:16:0:0:FWD:10 20:16:16:0:0:FWD:8 20:16:16:0:0:INV:10 19:16:16:0 19:16:16:0:0:INV 20:16:16:0:0:INV 16:24:24:1:0:INV 14:24:24:1:0:INV 10:32:16:1 7:32:16:1 10:26:16:1:0:INV 10:34:24:1:0:INV 10:16:16:1:0:PAIR 10:40:16:0 13:16:16:0:1 14:16:16:0:1 13:16:16:0:1:INV 14:16:16:0:1:INV 13:16:16:0:1:PAIR 14:16:16:0:1:PAIR 10:18:16:0:0:PAIR 12:24:24:0:1:PAIR 14:18:16:0:0:PAIR 16:32:16:1 16:32:16:1:0:INV 13:32:16:1 14:32:16:1:0:INV 16:28:16:1 19:16:16:0:1 17:16:16:1 20:16:16:1 17:24:24:1 17:18:18:0 20:18:18:0 17:18:18:0:0:INV 18:32:24:0 > "$OUT/bench.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, os
root = sys.argv[1]
f = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "intfft" in r["Name"]]
with open(os.path.join(root, "kernel_stats_compact.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
print(open(os.path.join(root, "kernel_stats_compact.csv")).read())
PY
grep -v amdgpu.ids "$OUT/bench.log" | cut -c1-200
rm -rf "$OUT/trace"  # raw trace: gpurun_out/ is merged back only below 64 MiB
