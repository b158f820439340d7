#!/bin/bash
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_o_bench.jsonl
: > $OUT
run() { echo "{\"variant\": \"$1\"}" >> $OUT; shift; env "$@" python tools/bench_configs.py C4 >> $OUT 2>&1; }
for rep in 1 2; do
run base X=1
run bcu3 INTFFT_2XB_BLOCKS_PER_CU=3
run bcu4 INTFFT_2XB_BLOCKS_PER_CU=4
run bcu8 INTFFT_2XB_BLOCKS_PER_CU=8
run onestream INTFFT_ONE_STREAM=1
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
