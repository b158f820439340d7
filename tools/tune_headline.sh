#!/bin/bash
# tools/tune_headline.sh -- blocks per CU of the persistent wave kernels (INTFFT_BLOCKS_PER_CU overrides the planner's cap)
for rep in 1 2 3; do for b in 4 6 8; do
  echo -n "rep=$rep blocks=$b: "
  INTFFT_BLOCKS_PER_CU=$b python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['kernel_ms']*1000,2),'us', round(d['roofline']['frac'],4))"
done; done
for b in 4 6 8; do
  echo "blocks=$b:"; INTFFT_BLOCKS_PER_CU=$b python tools/bench_configs.py C2r 7:16:16:0 9:16:16:0 C2inv C2pair C2u 7:16:16:0:0:PAIR 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ',d['config'],d['dir'],round(d['Gsample/s'],1),d['kernel'])"
done
