#!/bin/bash
cd /root/repo
for c in 0 1 2 4 3 6; do echo "STEM_DEBUG $c"; DVS_STEM_DEBUG=$c DVS_LIB=/root/repo/deep-visual-slam_amd/csrc/build/variant_timing.so timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_stemdbg.txt 2>&1; grep -n "^== conv_fwd" -A32 gpurun_out/per_launch_stemdbg.txt | grep -E "^[0-9]+-\s+(0|30)  work"; grep -n "^== conv_wgrad" -A54 gpurun_out/per_launch_stemdbg.txt | grep -E "^[0-9]+-\s+(22|52)  work"; done
