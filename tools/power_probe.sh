#!/bin/bash
# tools/power_probe.sh <tag> [rnd] [dir]: tools/power_probe.py plain, then under rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -> shader clock per phase
set -u
TAG=${1:-r06}; RND=${2:-1}; DIR=${3:-FWD}
REPO=$(pwd); OUT=$REPO/gpurun_out/${TAG}_power_${RND}_${DIR}; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python tools/power_probe.py $RND $DIR > $OUT/plain.jsonl 2>&1
cd /tmp
PROBE_REPS=200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof -- python $REPO/tools/power_probe.py $RND $DIR > $OUT/prof.jsonl 2>&1
cd $REPO
python - $OUT <<'PY' > $OUT/clock.txt
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/prof/**/*counter_collection.csv', recursive=True)[0]
t = glob.glob(out + '/prof/**/*kernel_trace.csv', recursive=True)[0]
dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(t))}
rows = [r for r in csv.DictReader(open(f)) if 'fft1024' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
vals = [(float(r['Counter_Value']) / 8 / dur[r['Dispatch_Id']], dur[r['Dispatch_Id']] / 1e3) for r in rows if r['Dispatch_Id'] in dur]
# phases: 200 ramp, then 4 x (50 + 200)
names = ['ramp', 'zeros', 'uniform_13bit', 'uniform_15bit', 'uniform_16bit']
bounds = [0, 200, 450, 700, 950, 1200]
for i, n in enumerate(names):
    ch = vals[bounds[i] + (50 if i else 100):bounds[i + 1]]
    if ch:
        print("%-14s dispatches %4d  shader clock %.3f GHz  kernel %.1f us (serialised by the counter pass)" % (n, len(ch), sum(c for c, _ in ch) / len(ch), sum(d for _, d in ch) / len(ch)))
PY
rm -rf $OUT/prof
cat $OUT/plain.jsonl | cut -c1-200; cat $OUT/clock.txt
