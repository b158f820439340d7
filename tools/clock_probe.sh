#!/bin/bash
# effective shader clock of the FFT kernel at steady state: GRBM_GUI_ACTIVE / 8 XCDs / duration
export TMPDIR=/tmp; REPO=$(pwd); OUT=$REPO/gpurun_out/clock; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python $REPO/tools/sustain.py 800 > $OUT/log.txt 2>&1
cd $REPO
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/clock/**/*counter_collection.csv',recursive=True)[0]
t=glob.glob('gpurun_out/clock/**/*kernel_trace.csv',recursive=True)
rows=[r for r in csv.DictReader(open(f)) if 'fft1024' in r['Kernel_Name']]
print(rows[0].keys())
dur={}
if t:
    for r in csv.DictReader(open(t[0])):
        dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
vals=[]
for r in rows:
    d=dur.get(r['Dispatch_Id'])
    if d: vals.append((float(r['Counter_Value'])/8/d, d/1e3))
for i in range(0,len(vals),50):
    ch=vals[i:i+50]
    print("dispatch %4d: clock %.2f GHz  dur %.1f us"%(i, sum(c for c,_ in ch)/len(ch), sum(d for _,d in ch)/len(ch)))
PY
