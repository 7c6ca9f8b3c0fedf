#!/bin/bash
# 300 timed steps of the headline workload + a repeat of the deterministic-repeat check (gpurun_out/)
cd /root/repo
timeout -k 10 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-other-configs --no-stock-caller > gpurun_out/bench_soak.json 2> gpurun_out/bench_soak.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_soak.json')); print('soak: step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['loss_check'])"
timeout -k 10 600 python tools/repeat_check.py > gpurun_out/repeat_check.txt 2>&1; echo "repeat rc=$?"; tail -5 gpurun_out/repeat_check.txt
