#!/bin/bash
# tools/pmc_lds.sh <bench_configs specs...> -- LDS instruction and bank-conflict counters per kernel (one --pmc pass, no trace domains)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_lds
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/p" -- python $REPO/tools/bench_configs.py "$@" > "$OUT/log.txt" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, os, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "intfft" in r["Kernel_Name"] and "twiddle" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void intfft::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-62s %10s %12s %8s %10s" % ("kernel", "LDS insts", "conflict cyc", "cyc/inst", "VALU insts"))
for k, d in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    li = m.get("SQ_INSTS_LDS", 0)
    print("%-62s %10.3g %12.3g %8.2f %10.3g" % (k, li, m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_BANK_CONFLICT", 0) / li if li else 0, m.get("SQ_INSTS_VALU", 0)))
PY
