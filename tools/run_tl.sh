#!/bin/bash
# step timeline (HIP events) + multi-stream kernel trace of the current build
cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 python tools/step_timeline.py 12 > gpurun_out/step_timeline.txt 2>&1; echo rc=$?; cat gpurun_out/step_timeline.txt
OUT=/root/repo/gpurun_out/prof_tl; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap -- python3 /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-kernel-timing --config c3 > $OUT/overlap.log 2>&1 && python3 tools/trace_overlap.py $OUT/overlap 5 > $OUT/overlap.txt 2>&1; cat $OUT/overlap.txt | head -8
# keep the last 2 steps of the trace: compress
f=$(find $OUT/overlap -name '*_kernel_trace.csv' | head -1); python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
keep=rows[-1400:]
w=csv.writer(open('/root/repo/gpurun_out/trace_tail.csv','w'))
w.writerow(['name','queue','stream','start','end'])
t0=int(keep[0]['Start_Timestamp'])
for r in keep: w.writerow([r['Kernel_Name'][:60],r.get('Queue_Id',''),r.get('Stream_Id',''),int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0])
PY
find $OUT -name '*_kernel_trace.csv' -delete
