#!/bin/bash
# tools/tune_headline.sh -- grid size of the persistent headline kernel (INTFFT_BLOCKS_PER_CU overrides the planner's choice)
for rep in 1 2 3; do for b in 0 4 5 6 8 12 16 32; do
  echo -n "rep=$rep blocks/CU=$b: "
  INTFFT_DIAG=1 INTFFT_BLOCKS_PER_CU=$b python bench.py --no-cpu-baseline --no-extras --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['kernel_ms']*1000,2),'us', round(d['roofline']['frac'],4))"
done; done
