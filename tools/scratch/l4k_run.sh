cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_bypass.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4
FUZZ_R6=1 python tools/fuzz_soak.py 150 6611 2>&1 | grep -v amdgpu.ids | head -12
for s in 12:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 12:16:16:0:0:INV:0:BITREV_LANES:NATURAL 11:16:16:0:1:FWD:0:HALVES:BITREV_LANES 12:16:16:0:0:FWD:0:NATURAL:BITREV; do
  python tools/bench_configs.py $s 2>&1 | grep "^{" | tee -a gpurun_out/r06_lanes4k_rates.jsonl
done
