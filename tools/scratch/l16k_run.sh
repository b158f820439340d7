cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
FUZZ_R6=1 python tools/fuzz_soak.py 120 6621 2>&1 | grep -v amdgpu.ids | head -4
for s in 13:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 14:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 14:16:16:0:0:INV:0:BITREV_LANES:NATURAL 14:16:16:0:1:FWD:0:HALVES:BITREV_LANES 14:16:16:0:0:FWD:0:NATURAL:BITREV; do
  python tools/bench_configs.py $s 2>&1 | grep "^{" | tee -a gpurun_out/r06_lanes16k_rates.jsonl
done
