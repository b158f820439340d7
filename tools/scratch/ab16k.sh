cd $GRAFT_REPO_ROOT
run() { python tools/bench_configs.py "$@" 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'], d['kernel'], round(d['Gsample/s'],1), d['parity_prefix_ok'])"; }
for rep in 1 2 3; do
echo "== new"; run 14:16:16:0:0:FWD:0:NATURAL:BITREV 14:16:16:0:0:INV:0:BITREV:NATURAL 13:16:16:0:0:FWD:0:HALVES:BITREV 14:16:16:0 14:16:16:0:0:FWD:0:NATURAL:BITREV_LANES
echo "== old"; INTFFT_LIB=$PWD/build/variants/libintfft_old16k.so run 14:16:16:0:0:FWD:0:NATURAL:BITREV 14:16:16:0:0:INV:0:BITREV:NATURAL 13:16:16:0:0:FWD:0:HALVES:BITREV 14:16:16:0
done
