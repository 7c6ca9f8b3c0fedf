#!/bin/bash
cd /root/repo
for c in 0 1; do echo "NTN2 $c"; DVS_THIN_DGRAD_NTN2=$c timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_ntn$c.txt 2>&1; grep -n "^== conv_dgrad" -A66 gpurun_out/per_launch_ntn$c.txt | grep -E "^[0-9]+-\s+(24|25|26|27)  work"; done
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/t_thin.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_thin.log
