for i in 1 2 3; do
  echo "--- skip7"; PROBE_REPS=300 python tools/power_probe.py 0 FWD 2>&1 | grep -E "15bit|16bit" | cut -c1-150
  echo "--- noskip"; INTFFT_LIB=$PWD/build/variants/libintfft_noskip.so PROBE_REPS=300 python tools/power_probe.py 0 FWD 2>&1 | grep -E "15bit|16bit" | cut -c1-150
done
