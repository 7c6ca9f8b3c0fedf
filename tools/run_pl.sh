#!/bin/bash
cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; grep -n "^== conv_wgrad" -A54 gpurun_out/per_launch_fp32.txt | grep -E "^[0-9]+-\s+(22|52)  work"
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q -k "stem or planar or conv1" > gpurun_out/t_stem.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_stem.log
