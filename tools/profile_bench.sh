#!/bin/bash
# rocprofv3 passes over bench.py itself: tools/profile_bench.sh <c3|c2> [bench args...]
#   0. --kernel-trace, multi-stream     -> how the four streams overlap in the timed configuration (tools/trace_overlap.py)
#   1. --kernel-trace --stats, --serialize -> per-kernel durations on one stream (must agree with bench.py's HIP-event numbers)
#   2. --kernel-trace --pmc FETCH_SIZE  -> HBM read bytes per dispatch  (own pass, as the MI355X guide prescribes)
#   3. --kernel-trace --pmc WRITE_SIZE  -> HBM write bytes per dispatch
# Results land in gpurun_out/prof_<tag>/; tools/prof_bench_report.py condenses them for profiles/.
export TMPDIR=/tmp
TAG=$1; shift
OUT=/root/repo/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-kernel-timing --serialize --config $TAG $*"
CARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-kernel-timing --config $TAG $*"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/overlap -- python3 /root/repo/bench.py $CARGS > $OUT/overlap.log 2>&1 || { echo "overlap pass failed"; tail -5 $OUT/overlap.log; exit 1; }
python3 tools/trace_overlap.py $OUT/overlap 5 > $OUT/overlap.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 /root/repo/bench.py $ARGS > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 /root/repo/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 /root/repo/bench.py $ARGS > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/write.log; exit 1; }
python3 tools/prof_bench_report.py $OUT 5 > $OUT/report.txt 2>&1
cat $OUT/overlap.txt $OUT/report.txt
# keep only the condensed files small enough to merge back
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
