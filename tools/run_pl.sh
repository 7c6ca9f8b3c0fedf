#!/bin/bash
cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; echo rc=$?
grep -n "^== conv_wgrad" -A30 gpurun_out/per_launch_fp32.txt | sed -n 24,26p
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/t_thin.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_thin.log
