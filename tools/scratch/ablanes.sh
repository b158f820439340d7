cd $GRAFT_REPO_ROOT
run() { python tools/bench_configs.py "$@" 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'], d['kernel'], round(d['Gsample/s'],1), d['parity_prefix_ok'])"; }
S="C2native 10:16:16:0:0:INV:0:BITREV:HALVES 12:16:16:0:0:FWD:0:HALVES:BITREV 12:16:16:0:0:INV:0:BITREV:NATURAL 10:16:16:0:1:FWD:0:NATURAL:BITREV C2 C5"
for rep in 1 2 3; do
echo "== new"; run $S
echo "== old"; INTFFT_LIB=$PWD/build/variants/libintfft_oldlanes.so run $S
done
