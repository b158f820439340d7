#!/bin/bash
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wide_two_pass" 2>&1 | tail -8
python tools/bench_configs.py 16:24:24:1:0:INV 16:24:16:1:0:INV 14:24:24:1:0:INV C3 > gpurun_out/r04_k_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_k_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line); continue
    d=json.loads(line)
    print('%-24s %-4s %-30s %.1f Gs/s %.1f us parity=%s' % (d['config'], d['dir'], d['kernel'][:30], d['Gsample/s'], d['ms']*1e3, d['parity_prefix_ok']))
"
