#!/bin/bash
cd /root/repo
for g in 1024 512 2048 100000; do echo "FWD_WGS $g"; DVS_HEAD_FWD_WGS=$g timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin; done > gpurun_out/head_bench_fwd.txt; cat gpurun_out/head_bench_fwd.txt
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -k "head" > gpurun_out/t_head.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_head.log
