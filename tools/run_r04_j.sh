#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py 20:16:16:0:1 20:16:16:0:1:INV 19:16:16:0:1 18:16:16:0:1 17:16:16:0:1 18:16:16:0:1:INV 20:14:16:0:1 > gpurun_out/r04_j_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_j_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line); continue
    d=json.loads(line)
    print('%-24s %-4s %-30s %.1f Gs/s %.1f us parity=%s' % (d['config'], d['dir'], d['kernel'][:30], d['Gsample/s'], d['ms']*1e3, d['parity_prefix_ok']))
"
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "round_mode" 2>&1 | tail -5
