#!/bin/bash
# tests of the convolution files + a 40-step fp32 bench line (gpurun_out/bench_step.json)
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_chain_gpu.py -x -q > gpurun_out/t_step.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_step.log
timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_step.json 2> gpurun_out/bench_step.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_step.json')); print('step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['loss_check']['worst_rel_err'])"
