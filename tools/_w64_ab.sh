#!/bin/bash
cd /root/repo
export INTFFT_DIAG=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide128.py -q -x -k "64_bit or pair_of_dedicated or fuzz or wide or regime" 2>&1 | tail -5
for s in 10:32:16:1 10:32:24:1 10:34:24:1:0:INV 10:24:24:1:0:PAIR 16:24:24:1:0:INV 12:32:24:1; do
  python tools/bench_configs.py $s 2>&1 | grep -o '"config".*"kernel": "[^"]*"' | cut -c1-260
done
