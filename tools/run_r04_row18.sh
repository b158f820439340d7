#!/bin/bash
# round 4: k_fft4096_i16 with 18-dword LDS rows + b64 row reads (conflict-free under the per-instruction banking) against the shipped 20-dword rows
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_row18.txt
: > $OUT
for v in base "$@"; do
  if [ "$v" = base ]; then unset INTFFT_LIB; else export INTFFT_LIB=$GRAFT_REPO_ROOT/build/variants/libintfft_$v.so; fi
  echo "== variant $v" >> $OUT
  python tools/bench_configs.py C5 C5fwd C5inv 11:16:16:0 11:16:16:0:0:PAIR 12:16:16:0:1 12:16:16:0:1:PAIR C5 2>/dev/null | grep '^{' | cut -c1-120 >> $OUT
  bash tools/pmc_lds.sh C5 C5fwd C5inv 2>/dev/null | grep -v "^W\|^E" >> $OUT
  if [ "$v" != base ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "4096 or 2048 or single_pass or pair or round or config5 or C5" 2>&1 | tail -3 >> $OUT; fi
done
cat $OUT
