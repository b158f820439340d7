set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_fuzz.py tests/test_gpu_perf_floor.py -x -q 2>&1 | tail -15
for s in 10:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 10:16:16:0:0:INV:0:BITREV_LANES:NATURAL 10:16:16:0:1:FWD:0:HALVES:BITREV_LANES 12:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 12:16:16:0:0:INV:0:BITREV_LANES:NATURAL 16:24:24:1:0:FWD:0:NATURAL:BITREV_LANES 10:24:24:1:0:FWD:0:NATURAL:BITREV_LANES 14:16:16:0:0:FWD:0:NATURAL:BITREV_LANES; do
  python tools/bench_configs.py $s 2>&1 | grep "^{" | tee -a gpurun_out/r06_lanes_rates.jsonl
  INTFFT_NO_LANES_COMPOSITE=1 INTFFT_GENERIC_ONLY=1 python tools/bench_configs.py $s 2>&1 | grep "^{" | sed 's/^{/{"generic_only": true, /' | tee -a gpurun_out/r06_lanes_rates.jsonl
done
