#!/bin/bash
# step time against the stream layout: default (PoseNet stream + one weight-gradient stream per network), shared weight-gradient stream, none
cd /root/repo
run() { env "$@" timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-stock-caller > gpurun_out/bench_s.json 2> gpurun_out/bench_s.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_s.json')); print('$*: step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))"; }
run X=default
run DVS_WGRAD_STREAM=shared
run DVS_WGRAD_STREAM=0
run X=default
run DVS_SIDE_PRIORITY=low
run DVS_POSE_PRIORITY=high
