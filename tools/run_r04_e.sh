#!/bin/bash
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r04_e_bench.jsonl
: > $OUT
for rep in 1 2; do
for v in pipe nopipe; do
  echo "{\"variant\": \"$v\"}" >> $OUT
  if [ "$v" = pipe ]; then python tools/bench_configs.py C4 19:16:16:0 >> $OUT 2>&1; else INTFFT_NO_PIPE_AB=1 python tools/bench_configs.py C4 19:16:16:0 >> $OUT 2>&1; fi
done; done
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
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cabi.py -m gpu -q -x -k "two_pass or chunked or fullsize or large or capture or stream" 2>&1 | tail -5
