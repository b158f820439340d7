#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py C4 13:16:16:0 14:16:16:0 13:16:16:0:0:INV 14:16:16:0:0:INV 14:12:16:0 > gpurun_out/r04_b_bench.jsonl 2>&1
INTFFT_NO_FAST16K=1 python tools/bench_configs.py 13:16:16:0 14:16:16:0 13:16:16:0:0:INV 14:16:16:0:0:INV >> gpurun_out/r04_b_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_b_bench.jsonl
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_b_suite.txt
cat gpurun_out/r04_b_suite.txt
