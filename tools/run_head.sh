#!/bin/bash
# Disparity heads: tools/head_bench.py with the walking kernels and with the row forms, + the head tests (gpurun_out/)
cd /root/repo
echo "walking kernels (default)"; timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
echo "row forms"; DVS_HEAD_FWD_WALK=0 DVS_HEAD_DGRAD_IN=0 DVS_HEAD_WGRAD_IN=0 timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py -x -q -k head > gpurun_out/t_head.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_head.log
