cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# round-6 soaks on the library $(python -c 'from intfftk_amd import _capi as c; print(c.lib().intfft_version().decode())' 2>/dev/null)"
echo "## FUZZ_R6=1 python tools/fuzz_soak.py 420 6601"; FUZZ_R6=1 python tools/fuzz_soak.py 420 6601
echo "## python tools/fuzz_soak.py 300 6602"; python tools/fuzz_soak.py 300 6602
echo "## FUZZ_BIG=1 python tools/fuzz_soak.py 240 6603"; FUZZ_BIG=1 python tools/fuzz_soak.py 240 6603
echo "## FUZZ_NATIVE=1 python tools/fuzz_soak.py 180 6604"; FUZZ_NATIVE=1 python tools/fuzz_soak.py 180 6604
echo "## FUZZ_LONG=1 python tools/fuzz_soak.py 90 6605"; FUZZ_LONG=1 python tools/fuzz_soak.py 90 6605
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_fuzz_final.txt
tail -5 gpurun_out/r06_fuzz_final.txt; grep -c MISMATCH gpurun_out/r06_fuzz_final.txt
