cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_fuzz.py tests/test_gpu_perf_floor.py -x -q 2>&1 | tail -3
run() { python tools/bench_configs.py "$@" 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   ', d['config'], d['kernel'], round(d['Gsample/s'],1), d['parity_prefix_ok'])"; }
S="C2native 12:16:16:0:0:FWD:0:HALVES:BITREV 12:16:16:0:0:INV:0:BITREV:NATURAL 10:16:16:0:1:FWD:0:NATURAL:BITREV 14:16:16:0:0:FWD:0:NATURAL:BITREV 14:16:16:0:0:INV:0:BITREV:NATURAL"
for rep in 1 2 3; do
echo "== new"; run $S
echo "== old"; INTFFT_LIB=$PWD/build/variants/libintfft_oldlanes.so run $S
done
echo "== new LANES"; run 10:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 10:16:16:0:0:INV:0:BITREV_LANES:NATURAL 12:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 12:16:16:0:0:INV:0:BITREV_LANES:NATURAL 13:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 14:16:16:0:0:FWD:0:NATURAL:BITREV_LANES 14:16:16:0:0:INV:0:BITREV_LANES:NATURAL 14:16:16:0:1:FWD:0:HALVES:BITREV_LANES
