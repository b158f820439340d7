#!/bin/bash
# tools/profile.sh <tag> -- rocprofv3 evidence for bench.py (run on the GPU box via gpurun).
# Pass 1: --kernel-trace --stats (durations).  Passes 2..: PMC counters, each in its own run
# (never combined with other trace domains -- see the task notes).
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" ; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1
done
cd "$REPO"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep what profiles/ gets (stats csv, digest, summary); drop the raw traces (gpurun_out/ is merged back only below 64 MiB)
cp $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT"/trace "$OUT"/pmc_*
