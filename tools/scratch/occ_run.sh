cd $GRAFT_REPO_ROOT
run() { python tools/bench_configs.py "$@" 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'], d['kernel'], round(d['Gsample/s'],1), d['parity_prefix_ok'])"; }
for rep in 1 2; do
echo "== base"; run C2pair C2inv 10:16:16:0:1 10:16:16:0:1:INV 10:16:16:0:1:PAIR C5 C2
echo "== base FAST_PIPE=0"; INTFFT_FAST_PIPE=0 run 10:16:16:0:1 C2
echo "== xwpe5"; INTFFT_LIB=$PWD/build/variants/libintfft_xwpe5.so run C2pair C2inv 10:16:16:0:1:INV 10:16:16:0:1:PAIR
echo "== xwpe6"; INTFFT_LIB=$PWD/build/variants/libintfft_xwpe6.so run C2pair C2inv
echo "== fwpe5"; INTFFT_LIB=$PWD/build/variants/libintfft_fwpe5.so run 10:16:16:0:1 C2
done
