#!/bin/bash
# round 4, first GPU call: what do the L2's memory-side counters count on gfx950?  (tools/reqbench.hip: known byte counts per pattern)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r04_counters_list.txt 2>&1)
grep -o "TCC_[A-Z0-9_]*" gpurun_out/r04_counters_list.txt | sort -u > gpurun_out/r04_tcc_counters.txt
SINGLES=("TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum" "TCC_EA0_WRREQ_64B_sum"
         "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_DRAM_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RD_UNCACHED_32B_sum" "TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_64B_sum"
         "TCC_EA0_WR_UNCACHED_32B_sum" "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "TCC_EA0_RDREQ_GMI_sum" "TCC_EA0_RDREQ_IO_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_NC_READ_REQ_sum")
tools/pmc_sets.sh r04_reqbench "k_" "${SINGLES[@]}" -- $GRAFT_REPO_ROOT/build/reqbench 1024
BENCH_STEPS=4 BENCH_RAMP_S=0.05 tools/pmc_sets.sh r04_C4_req "k_big" "${SINGLES[@]:0:9}" -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C4
python tools/bench_configs.py C4 C5 > gpurun_out/r04_base_bench.jsonl 2>&1
cat gpurun_out/r04_base_bench.jsonl
cat gpurun_out/r04_reqbench_logs/failed.txt 2>/dev/null
