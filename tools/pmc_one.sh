#!/bin/bash
# tools/pmc_one.sh <bench_configs spec> -- a few SQ counters of one configuration's kernels (separate --pmc passes, no trace domains)
set -u
SPEC=$1
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_one
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/$name" -- python $REPO/tools/bench_configs.py $SPEC > "$OUT/$name.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, os, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "intfft" in r["Kernel_Name"] and "twiddle" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
