#!/bin/bash
cd /root/repo
export INTFFT_DIAG=1
for s in 20:16:16:0:0:FWD:10 21:16:16:0:0:FWD:10 22:16:16:0:0:FWD:10; do
  python tools/bench_configs.py $s 2>&1 | grep -o '"config".*"kernel": "[^"]*"' | cut -c1-260
  INTFFT_ONE_STREAM=1 python tools/bench_configs.py $s 2>&1 | grep -o '"config".*"kernel": "[^"]*"' | cut -c1-260
done
python -m pytest tests -m gpu -q -x > gpurun_out/gpu_suite.log 2>&1; tail -3 gpurun_out/gpu_suite.log
