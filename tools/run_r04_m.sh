#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py 15:16:16:0 16:16:16:0 17:16:16:0 18:16:16:0 16:16:16:0:0:INV 17:16:16:0:0:INV 18:16:16:0:0:INV 16:16:16:0:0:PAIR 18:16:16:0:0:PAIR 16:16:16:0:1 21:16:16:0:0:FWD:10 22:16:16:0:0:FWD:10 23:16:16:0:0:FWD:10 > gpurun_out/r04_m_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_m_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line); continue
    d=json.loads(line)
    print('%-24s %-4s %-34s %.1f Gs/s %.1f us parity=%s' % (d['config'], d['dir'], d['kernel'][:34], d['Gsample/s'], d['ms']*1e3, d['parity_prefix_ok']))
"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r04_m_suite.txt
cat gpurun_out/r04_m_suite.txt
