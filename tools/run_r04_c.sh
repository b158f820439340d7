#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/bench_configs.py C4 C3 C5 13:16:16:0 14:16:16:0 14:16:16:0:0:INV 20:16:16:0:0:INV 20:16:16:0:0:FWD:10 > gpurun_out/r04_c_bench.jsonl 2>&1
grep -v "^W\|^E\|amdgpu.ids" gpurun_out/r04_c_bench.jsonl
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04_c_suite.txt
cat gpurun_out/r04_c_suite.txt
