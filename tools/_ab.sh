export BENCH_INPUT_MB=2048
for spec in 16:16:16:0 14:16:16:0 18:16:16:0 16:16:16:0:0:INV 20:16:16:0:0:INV 16:16:16:0:0:PAIR 20:16:16:0:0:PAIR 16:16:16:1 16:16:16:0:1 20:16:16:0:1; do
a=$(python tools/bench_configs.py $spec 2>&1 | grep -o '"Gsample/s": [0-9.]*' | sed 's/"Gsample\/s": //')
b=$(INTFFT_TWO_STREAMS=1 python tools/bench_configs.py $spec 2>&1 | grep -o '"Gsample/s": [0-9.]*\|"parity_prefix_ok": [a-z]*' | paste - - | sed 's/"Gsample\/s": //; s/"parity_prefix_ok": //')
echo "$spec (2 GiB in) default: $a | two streams forced: $b"
done
