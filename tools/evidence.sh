#!/bin/bash
# tools/evidence.sh <round-tag> <part> [args]   -- every measurement that ends up under profiles/ comes from ONE of these parts, run on the GPU box
# (gpurun -- 'tools/evidence.sh r05 all').  Outputs go to gpurun_out/<tag>_*; copy what is to be judged into profiles/.
#
#   headline            tools/profile.sh <tag>: bench.py under rocprofv3 --kernel-trace --stats + one --pmc pass per counter set
#   digests [spec...]   tools/pmc_digest.sh <tag> <spec...>: per-kernel PMC digests (default: C2 C3 C4 C5, the shipped 2-D plans, N = 2^14, the 24-bit inverse)
#   configs             tools/profile_configs.sh <tag>: kernel stats + bench_configs lines of the other configurations
#   matrix              tools/bench_matrix.py -> <tag>_coverage_matrix.md
#   bench               the contract's line (python bench.py) and --config C3 / C4 / C5 lines
#   suite               python -m pytest tests -m gpu
#   calib               tools/reqbench.hip under the L2 memory-side request counters (what a request counts on gfx950)
#   variants <config> <name...>   <config> of tools/bench_configs.py on experimental builds build/variants/libintfft_<name>.so (tools/build_variant.sh,
#                       e.g. `tools/build_variant.sh map1 intfft_big2x.hip -DINTFFT_2XA_MAP=1`): Gsample/s on two streams / one stream, the one-stream
#                       kernel durations (rocprofv3 --kernel-trace --stats) and, with EVIDENCE_PMC=1, the L2 request counters per variant.  "base" = the library as built.
#                       Variant flags that exist at HEAD: intfft_big2x.hip -DINTFFT_2XA_MAP=1|2, -DINTFFT_2XA_PLAIN, -DINTFFT_2XB_PLAIN; intfft_fast4096.hip -DINTFFT_4K_ABL=<bits>
#   at32check           every translation unit rebuilt with -DINTFFT_AT32_CHECK (at32 / at32b trap when their base pointer is not wave-uniform:
#                       intfft_device.hpp) as build/variants/libintfft_at32chk.so (build it HERE first: tools/evidence.sh <tag> at32build), then the parity
#                       suites on that library
#   tilebench           build/tilebench (tools/tilebench.hip) at 64 and 256 frames
#   collect             (run HERE, after gpurun merged gpurun_out/) copy the outputs of `all` into profiles/ under the names profiles/README.md lists
#   all                 headline digests configs matrix bench suite
set -u
TAG=${1:?round tag}; PART=${2:-all}; shift; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp

part_headline() { bash tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -5 gpurun_out/${TAG}_profile.log; }
part_digests() {
  local specs=("$@")
  [ ${#specs[@]} -eq 0 ] && specs=(C2 C3 C4 C5 20:16:16:0:0:FWD:10 21:16:16:0:0:FWD:10 21:16:16:0:0:INV:10 14:16:16:0 16:24:24:1:0:INV 19:16:16:0 16:32:16:1 17:16:16:1 17:18:18:0)
  BENCH_STEPS=6 BENCH_RAMP_S=0.1 bash tools/pmc_digest.sh $TAG "${specs[@]}" > gpurun_out/${TAG}_pmc_digest.log 2>&1
  tail -20 gpurun_out/${TAG}_pmc_digest.log
}
part_configs() { bash tools/profile_configs.sh $TAG > gpurun_out/${TAG}_profile_configs.log 2>&1; tail -3 gpurun_out/${TAG}_profile_configs.log; }
part_matrix() {
  python tools/bench_matrix.py > gpurun_out/${TAG}_coverage_matrix.md 2> gpurun_out/${TAG}_coverage_matrix.err
  tail -3 gpurun_out/${TAG}_coverage_matrix.md; echo "mismatches: $(grep -c MISMATCH gpurun_out/${TAG}_coverage_matrix.md)"
}
part_bench() {
  python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
  tail -c 400 gpurun_out/${TAG}_bench_default.json; echo
  : > gpurun_out/${TAG}_other_configs_bench.jsonl
  for c in C3 C4 C5; do python bench.py --config $c --no-cpu-baseline 2> gpurun_out/${TAG}_bench_$c.err >> gpurun_out/${TAG}_other_configs_bench.jsonl; done
}
# collect: run HERE after the gpurun call -- what gpurun merged into gpurun_out/ goes to profiles/ under the names profiles/README.md lists
part_collect() {
  local P=profiles G=gpurun_out
  cp $G/prof_$TAG/kernel_stats.csv $P/${TAG}_kernel_stats.csv
  cp $G/prof_$TAG/digest.json $P/${TAG}_k_fft1024_pmc_digest.json
  cp $G/prof_$TAG/summary.txt $P/${TAG}_k_fft1024_rocprofv3_summary.txt
  declare -A nm=([C2]=C2 [C3]=C3 [C4]=C4 [C5]=C5 [20_16_16_0_0_FWD_10]=2d_n2pow20 [21_16_16_0_0_FWD_10]=2d_n2pow21 [21_16_16_0_0_INV_10]=2d_n2pow21_inv
                 [14_16_16_0]=n2pow14_fwd [16_24_24_1_0_INV]=n2pow16_24bit_inv [19_16_16_0]=n2pow19_fwd [16_32_16_1]=n2pow16_32bit_fwd [17_16_16_1]=n2pow17_16bit_unscaled_fwd [17_18_18_0]=n2pow17_18bit_scaled_fwd)
  for k in "${!nm[@]}"; do [ -f $G/pmc_$TAG/${k}_pmc_digest.json ] && cp $G/pmc_$TAG/${k}_pmc_digest.json $P/${TAG}_${nm[$k]}_pmc_digest.json; done
  cp $G/prof_cfg_$TAG/kernel_stats_compact.csv $P/${TAG}_other_configs_kernel_stats.csv
  grep '^{' $G/prof_cfg_$TAG/bench.log > $P/${TAG}_other_configs_rates.jsonl
  for f in coverage_matrix.md bench_default.json other_configs_bench.jsonl gpu_suite.txt; do cp $G/${TAG}_$f $P/${TAG}_$f; done
  ls -la $P/${TAG}_* | wc -l
}
part_suite() { timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_suite.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_suite.txt; }
part_calib() {
  [ -x build/reqbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o build/reqbench tools/reqbench.hip
  local sets=("TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum")
  tools/pmc_sets.sh ${TAG}_reqbench "k_" "${sets[@]}" -- $ROOT/build/reqbench 1024
}
part_variants() {
  local cfg=${1:?config}; shift
  local out=gpurun_out/${TAG}_${cfg}_variants.jsonl
  : > $out
  for v in base "$@"; do
    if [ "$v" = base ]; then unset INTFFT_LIB; else export INTFFT_LIB=$ROOT/build/variants/libintfft_$v.so; fi
    echo "{\"variant\": \"$v\", \"streams\": 2}" >> $out
    python tools/bench_configs.py $cfg 2>&1 | grep "^{" >> $out
    echo "{\"variant\": \"$v\", \"streams\": 1}" >> $out
    INTFFT_ONE_STREAM=1 python tools/bench_configs.py $cfg 2>&1 | grep "^{" >> $out
    (cd /tmp && rm -rf /tmp/var_$v && INTFFT_ONE_STREAM=1 BENCH_STEPS=5 BENCH_RAMP_S=0.1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/var_$v -o t -- python $ROOT/tools/bench_configs.py $cfg > /tmp/var_$v.log 2>&1
     f=$(find /tmp/var_$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-300 $f | grep -v "k_twiddle\|k_pack\|Cijk\|at::\|elementwise" | head -8 > $ROOT/gpurun_out/${TAG}_${cfg}_variant_${v}_stats.csv)
    if [ "${EVIDENCE_PMC:-0}" = 1 ]; then
      BENCH_STEPS=3 BENCH_RAMP_S=0.02 tools/pmc_sets.sh ${TAG}_${cfg}_var_$v "k_" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" -- python $ROOT/tools/bench_configs.py $cfg
    fi
  done
  unset INTFFT_LIB
  cat $out
}
part_at32build() { # (CPU: cross-compiles)
  tools/build_variant_multi.sh at32chk "-DINTFFT_AT32_CHECK" $(cd intfftk_amd/csrc && ls *.hip)
}
part_at32check() {
  INTFFT_LIB=$ROOT/build/variants/libintfft_at32chk.so timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_2d.py tests/test_gpu_fullsize.py tests/test_gpu_widelong.py -m gpu -x -q > gpurun_out/${TAG}_at32check.txt 2>&1
  tail -3 gpurun_out/${TAG}_at32check.txt
}
part_tilebench() {
  [ -x build/tilebench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o build/tilebench tools/tilebench.hip
  ./build/tilebench 64 20 > gpurun_out/${TAG}_tilebench_64.txt 2>&1; ./build/tilebench 256 10 > gpurun_out/${TAG}_tilebench_256.txt 2>&1
  tail -3 gpurun_out/${TAG}_tilebench_256.txt
}
case $PART in
  all) part_headline; part_digests; part_configs; part_matrix; part_bench; part_suite ;;
  headline|digests|configs|matrix|bench|suite|calib|variants|tilebench|at32build|at32check|collect) part_$PART "$@" ;;
  *) echo "unknown part $PART"; exit 2 ;;
esac
