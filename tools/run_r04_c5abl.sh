#!/bin/bash
# round 4: where the C5 pair's time beyond its VALU issue goes -- ablated variants of k_fft4096_i16 (INTFFT_4K_ABL bits, intfft_fast4096.hip;
# their results are wrong by construction, their time and LDS counters are the measurement)
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_c5abl.txt
: > $OUT
for v in base "$@"; do
  if [ "$v" = base ]; then unset INTFFT_LIB; else export INTFFT_LIB=$GRAFT_REPO_ROOT/build/variants/libintfft_c5abl$v.so; fi
  echo "== variant $v" >> $OUT
  python tools/bench_configs.py C5 C5 2>/dev/null | grep '^{' | cut -c1-220 >> $OUT
  bash tools/pmc_lds.sh C5 2>/dev/null | grep -v "^W\|^E" >> $OUT
done
cat $OUT
