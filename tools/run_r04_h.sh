#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py C5 C5fwd C5inv 11:16:16:0:0:PAIR 12:16:16:0:1:PAIR 12:16:16:0:1 C5 > gpurun_out/r04_h_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_h_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line); continue
    d=json.loads(line)
    print('%-24s %-4s %-30s %.1f Gs/s %.1f us parity=%s' % (d['config'], d['dir'], d['kernel'][:30], d['Gsample/s'], d['ms']*1e3, d['parity_prefix_ok']))
"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "4096 or pair or config5 or narrow or round" 2>&1 | tail -3
