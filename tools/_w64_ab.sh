#!/bin/bash
cd /root/repo
export INTFFT_DIAG=1
python -m pytest tests/test_gpu_parity.py -q -x -k "64_bit or pair_of_dedicated" 2>&1 | tail -5
for s in 10:32:16:1 7:32:16:1 7:32:16:1:0:PAIR 7:39:16:1:0:INV; do
  python tools/bench_configs.py $s 2>&1 | grep -o '"config".*"kernel": "[^"]*"' | cut -c1-260
done
