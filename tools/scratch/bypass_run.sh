cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bypass.py tests/test_gpu_parity.py -k "bypass or use_fly" -x -q 2>&1 | tail -8
for s in 10:16:16:0:0:FWD:0:NATURAL:NATURAL:0 10:16:16:0:0:FWD:0:NATURAL:BITREV:0 10:16:16:1:0:FWD:0:HALVES:BITREV:0 16:24:24:1:0:INV:0:BITREV:NATURAL:0; do
  python tools/bench_configs.py $s 2>&1 | grep "^{" | tee -a gpurun_out/r06_bypass_rates.jsonl
  INTFFT_NO_BYPASS_COPY=1 python tools/bench_configs.py $s 2>&1 | grep "^{" | sed 's/^{/{"generic": true, /' | tee -a gpurun_out/r06_bypass_rates.jsonl
done
