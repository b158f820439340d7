python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_cabi.py -x -q -k "config3 or config4 or large_configs or two_pass or three_pass or in_place or exec_host or sharded or multi_pass_lengths" 2>&1 | tail -3
for spec in C4 C3 16:16:16:0 18:16:16:0 14:16:16:0 20:16:16:0:0:INV 16:16:16:0:0:PAIR 16:16:16:1 21:16:16:0:0:FWD:10; do
a=$(python tools/bench_configs.py $spec 2>&1 | grep -o '"Gsample/s": [0-9.]*\|"parity_prefix_ok": [a-z]*' | paste - -)
b=$(INTFFT_ONE_STREAM=1 python tools/bench_configs.py $spec 2>&1 | grep -o '"Gsample/s": [0-9.]*' )
echo "$spec two-stream: $a | one-stream: $b"
done
