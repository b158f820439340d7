R=$(pwd)
for spec in C2 C2r C2inv C2pair C2u tb7u C5 C5fwd C5inv C3 C4 7:16:16:0 16:16:16:0 14:16:16:0:0:INV 18:16:16:0 20:16:16:0:0:INV 16:16:16:0:0:PAIR 12:16:16:1 10:24:24:1 10:32:16:1 16:16:16:1 13:24:24:1 20:16:16:0:0:FWD:10 22:16:16:0:0:FWD:10 5:16:16:0; do
a=$(python tools/bench_configs.py $spec 2>&1 | grep -o '"Gsample/s": [0-9.]*\|"parity_prefix_ok": [a-z]*' | paste - - | sed 's/"Gsample\/s": //; s/"parity_prefix_ok": //')
b=$(INTFFT_LIB=$R/build/variants/libintfft_ntld.so python tools/bench_configs.py $spec 2>&1 | grep -o '"Gsample/s": [0-9.]*' | sed 's/"Gsample\/s": //')
echo "$spec plain-loads: $a | nt-loads: $b"
done
