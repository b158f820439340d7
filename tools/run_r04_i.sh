#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py 13:16:16:0:0:PAIR 14:16:16:0:0:PAIR 14:12:16:0:0:PAIR > gpurun_out/r04_i_bench.jsonl 2>&1
INTFFT_NO_FAST16K=1 python tools/bench_configs.py 13:16:16:0:0:PAIR 14:16:16:0:0:PAIR >> gpurun_out/r04_i_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_i_bench.jsonl | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): print(line); continue
    d=json.loads(line)
    print('%-24s %-4s %-30s %.1f Gs/s %.1f us parity=%s' % (d['config'], d['dir'], d['kernel'][:30], d['Gsample/s'], d['ms']*1e3, d['parity_prefix_ok']))
"
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "single_pass_n8192 or three_pass_pair or small_batches or narrow_data_multi" 2>&1 | tail -3
