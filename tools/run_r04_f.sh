#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py C4 C3 13:16:16:0 14:16:16:0 13:16:16:0:0:INV 14:16:16:0:0:INV 20:16:16:0:0:INV 19:16:16:0 20:16:16:0:0:FWD:10 20:16:16:0:0:INV:10 16:16:16:0 17:16:16:0 18:16:16:0 17:16:16:0:0:INV 16:16:16:0:0:PAIR 20:16:16:0:1 > gpurun_out/r04_f_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_f_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line); continue
    d=json.loads(line)
    print('%-24s %-3s %-34s %.1f Gs/s  parity=%s' % (d['config'], d['dir'], d['kernel'][:34], d['Gsample/s'], d['parity_prefix_ok']))
"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r04_f_suite.txt
cat gpurun_out/r04_f_suite.txt
