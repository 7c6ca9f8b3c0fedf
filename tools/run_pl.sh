#!/bin/bash
cd /root/repo
echo default; timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; grep -n "^== conv_dgrad" -A66 gpurun_out/per_launch_fp32.txt | grep -E "^[0-9]+-\s+(24|25|26)  work"
for v in 1 2 16 5 13 29; do echo "THIN_DBG $v"; DVS_LIB=/root/repo/deep-visual-slam_amd/csrc/build/variant_tdbg$v.so timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_tdbg.txt 2>&1; grep -n "^== conv_dgrad" -A66 gpurun_out/per_launch_tdbg.txt | grep -E "^[0-9]+-\s+(24|25|26)  work"; done
