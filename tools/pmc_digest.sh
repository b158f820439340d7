#!/bin/bash
# tools/pmc_digest.sh <tag> <bench_configs spec> [more specs]  -- per-kernel PMC digest of one configuration (run on the GPU box via gpurun).
# One `rocprofv3 --pmc` pass per counter set (never combined with trace domains), plus one --kernel-trace pass for durations;
# writes gpurun_out/pmc_<tag>/<spec>_pmc_digest.json: per kernel {mean counters per launch, launches, avg ns, HBM bytes derived as in
# MI355X_MICROARCH.md (FETCH_SIZE x 2 on gfx950; cross-check TCC_EA0_RDREQ x 128 B / WRREQ x 64 B; round 4: the request-size split
# TCC_EA0_RDREQ_128B / _64B / _32B and WRREQ_64B, calibrated with tools/reqbench.hip)} and the config's algorithmic bytes.
set -u
TAG=$1; shift
REPO=$(pwd)
export TMPDIR=/tmp
for SPEC in "$@"; do
  SAFE=$(echo "$SPEC" | tr ':' '_')
  OUT=$REPO/gpurun_out/pmc_$TAG/$SAFE
  rm -rf "$OUT"; mkdir -p "$OUT"
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python $REPO/tools/bench_configs.py $SPEC > "$OUT/trace.log" 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$i" -- python $REPO/tools/bench_configs.py $SPEC > "$OUT/pmc_$i.log" 2>&1
  done
  cd "$REPO"
  python tools/pmc_digest.py "$OUT" "$SPEC" > "$REPO/gpurun_out/pmc_$TAG/${SAFE}_pmc_digest.json"
  rm -rf "$OUT"  # the raw rocprofv3 output (tens of MiB per configuration): gpurun_out/ is merged back only below 64 MiB
  cat "$REPO/gpurun_out/pmc_$TAG/${SAFE}_pmc_digest.json" | python -c "
import json,sys
d=json.load(sys.stdin)
print(d['config'], 'algorithmic bytes per launch-set:', d['algorithmic_bytes_per_call'])
for k,v in d['kernels'].items():
    print('  %-34s n=%-4d avg %.1f us  hbm rd %.1f MB wr %.1f MB  VALU %.3g  LDS %.3g  conflict %.3g' % (k[:34], v['launches'], v['avg_ns']/1e3, v.get('hbm_read_bytes',0)/1e6, v.get('hbm_write_bytes',0)/1e6, v.get('SQ_INSTS_VALU',0), v.get('SQ_INSTS_LDS',0), v.get('SQ_LDS_BANK_CONFLICT',0)))
print('  traffic / algorithmic = %s' % d['traffic_over_algorithmic'])
"
done
