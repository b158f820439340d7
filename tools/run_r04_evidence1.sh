#!/bin/bash
# round 4 evidence, part 1: headline profile, PMC digests per configuration, kernel stats of the other configurations, coverage matrix
set -u
mkdir -p gpurun_out
bash tools/profile.sh r04 > gpurun_out/r04_profile.log 2>&1
BENCH_STEPS=6 BENCH_RAMP_S=0.1 bash tools/pmc_digest.sh r04 C2 C3 C4 C5 20:16:16:0:0:FWD:10 21:16:16:0:0:FWD:10 14:16:16:0 16:24:24:1:0:INV > gpurun_out/r04_pmc_digest.log 2>&1
tail -40 gpurun_out/r04_pmc_digest.log
bash tools/profile_configs.sh r04 > gpurun_out/r04_profile_configs.log 2>&1
python tools/bench_matrix.py > gpurun_out/r04_coverage_matrix.md 2> gpurun_out/r04_coverage_matrix.err
tail -5 gpurun_out/r04_coverage_matrix.md
grep -c MISMATCH gpurun_out/r04_coverage_matrix.md
