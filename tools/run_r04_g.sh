#!/bin/bash
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_g_bench.jsonl
: > $OUT
for v in base ockl base ockl; do
  if [ "$v" = base ]; then unset INTFFT_LIB; else export INTFFT_LIB=$GRAFT_REPO_ROOT/build/variants/libintfft_$v.so; fi
  echo "{\"variant\": \"$v\"}" >> $OUT
  python tools/bench_configs.py C4 14:16:16:0 13:16:16:0 14:16:16:0:0:INV 17:16:16:0 20:16:16:0:0:FWD:10 20:16:16:0:0:INV >> $OUT 2>&1
done
unset INTFFT_LIB
grep -v "^W\|^E\|amdgpu.ids" $OUT | python -c "
import sys, json
cur=None
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    if 'variant' in d: cur=d['variant']; continue
    print('%-6s %-22s %.1f Gs/s  parity=%s' % (cur, d['config'], d['Gsample/s'], d['parity_prefix_ok']))
"
