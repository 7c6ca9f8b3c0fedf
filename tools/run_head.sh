#!/bin/bash
cd /root/repo
echo "input-indexed dgrad"; timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
echo "row-form dgrad"; DVS_HEAD_DGRAD_IN=0 timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_chain_gpu.py -x -q > gpurun_out/t_head.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_head.log
