#!/bin/bash
cd /root/repo
echo "walking forward"; timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
echo "row-form forward"; DVS_HEAD_FWD_WALK=0 timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
timeout -k 10 600 python -m pytest tests/test_conv_gpu.py tests/test_chain_gpu.py -x -q > gpurun_out/t_head.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_head.log
