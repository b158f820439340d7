for i in 1 2; do
  echo "--- short (library)"; python tools/bench_configs.py 10:18:18:0:1 10:18:18:0:1:INV 12:18:18:0:1 16:24:24:0:1 12:18:18:0:1:INV 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['kernel'], round(d['Gsample/s'],1), d['parity_prefix_ok'])"
  echo "--- long (variant)"; INTFFT_LIB=$PWD/build/variants/libintfft_rhu2long.so python tools/bench_configs.py 10:18:18:0:1 10:18:18:0:1:INV 12:18:18:0:1 16:24:24:0:1 12:18:18:0:1:INV 2>&1 | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['kernel'], round(d['Gsample/s'],1), d['parity_prefix_ok'])"
done
