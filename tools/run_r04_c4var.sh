#!/bin/bash
# round 4: C4 pass variants (build/variants/libintfft_<name>.so): Gsample/s on two streams / one stream + the L2 memory-side request counters per variant
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_c4var.jsonl
: > $OUT
for v in base "$@"; do
  if [ "$v" = base ]; then unset INTFFT_LIB; else export INTFFT_LIB=$GRAFT_REPO_ROOT/build/variants/libintfft_$v.so; fi
  echo "{\"variant\": \"$v\", \"streams\": 2}" >> $OUT
  python tools/bench_configs.py C4 >> $OUT 2>&1
  echo "{\"variant\": \"$v\", \"streams\": 1}" >> $OUT
  INTFFT_ONE_STREAM=1 python tools/bench_configs.py C4 >> $OUT 2>&1
  BENCH_STEPS=3 BENCH_RAMP_S=0.02 tools/pmc_sets.sh r04_c4var_$v "k_big" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C4
done
grep -v "^W\|^E\|amdgpu.ids" $OUT
