#!/bin/bash
# Thin decoder layers: per-launch table of a step (at batch 12 the decoder's data gradients are launches 24-27 of that slot, the thin
# weight gradients 23 / 24 of theirs) + the convolution and full-size tests (gpurun_out/)
cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; echo rc=$?
grep -n "^== conv_dgrad" -A66 gpurun_out/per_launch_fp32.txt | grep -E "^[0-9]+-\s+(24|25|26|27|28)  work"
grep -n "^== conv_wgrad" -A54 gpurun_out/per_launch_fp32.txt | grep -E "^[0-9]+-\s+(23|24)  work"
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py -x -q > gpurun_out/t_thin.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_thin.log
