#!/bin/bash
# kernel-trace of one tools/bench_configs.py config: per-launch durations
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/trace_$1; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $REPO/tools/bench_configs.py $1 > $OUT/log.txt 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'intfft' in r['Kernel_Name'] and 'twiddle' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-12:]
for r in tail:
    print("%-40s grid=%s lds=%s dur=%.1f us"%(r['Kernel_Name'][:40], r.get('Grid_Size_X',r.get('Grid_Size')), r.get('LDS_Block_Size'), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
