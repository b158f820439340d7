#!/bin/bash
# round 4, final collection on the library as committed: parts 1 and 2 back to back
set -u
bash tools/run_r04_evidence1.sh > gpurun_out/r04_ev1.log 2>&1
bash tools/run_r04_evidence2.sh > gpurun_out/r04_ev2.log 2>&1
tail -3 gpurun_out/r04_ev1.log; tail -12 gpurun_out/r04_ev2.log | cut -c1-300
