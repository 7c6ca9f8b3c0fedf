#!/bin/bash
# bf16 mode: its tests, a 40-step bench line and the per-launch table (gpurun_out/)
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/t_bf16.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/t_bf16.log
timeout -k 10 600 python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_bf16.json')); print('bf16 step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['loss_check']['worst_rel_err']); print({k:round(v,3) for k,v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv:-kv[1])[:5]})"
DVS_PRECISION=bf16 timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_bf16.txt 2>&1; echo rc=$?
