#!/bin/bash
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_n_bench.jsonl
: > $OUT
run() { echo "{\"variant\": \"$1\"}" >> $OUT; shift; env "$@" python tools/bench_configs.py C4 >> $OUT 2>&1; }
for rep in 1 2; do
run base X=1
run groups32 INTFFT_2XA_GROUPS=32
run groups16 INTFFT_2XA_GROUPS=16
run setprio INTFFT_LIB=$GRAFT_REPO_ROOT/build/variants/libintfft_setprio.so
run mb96 INTFFT_SCRATCH_MB=96
run mb160 INTFFT_SCRATCH_MB=160
run mb192 INTFFT_SCRATCH_MB=192
done
grep -v "^W\|^E\|amdgpu.ids" $OUT | python -c "
import sys, json
cur=None
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    if 'variant' in d: cur=d['variant']; continue
    print('%-10s %-8s %.1f Gs/s  parity=%s' % (cur, d['config'], d['Gsample/s'], d['parity_prefix_ok']))
"
