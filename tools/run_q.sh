#!/bin/bash
# step time against HIP's hardware-queue cap (GPU_MAX_HW_QUEUES; default 4)
cd /root/repo
for q in 4 8 2 6; do GPU_MAX_HW_QUEUES=$q timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-stock-caller > gpurun_out/bench_q$q.json 2> gpurun_out/bench_q$q.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_q$q.json')); print('queues $q: step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))"; done
