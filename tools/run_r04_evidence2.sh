#!/bin/bash
# round 4 evidence, part 2: the contract's bench line for every BASELINE configuration (with the r04 digests in profiles/), the new tests, the GPU suite
set -u
mkdir -p gpurun_out
: > gpurun_out/r04_other_configs_bench.jsonl
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
tail -c 600 gpurun_out/r04_bench_default.json
for c in C3 C4 C5; do
  python bench.py --config $c --no-cpu-baseline 2> gpurun_out/r04_bench_$c.err | tee -a gpurun_out/r04_other_configs_bench.jsonl | cut -c1-700
done
timeout 900 python -m pytest tests/test_gpu_xcd.py tests/test_gpu_parity.py tests/test_gpu_cabi.py -k "xcd or two_stream_switch or outside or reentrant" -x -q 2>&1 | tail -5
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite.txt 2>&1
tail -5 gpurun_out/r04_gpu_suite.txt
