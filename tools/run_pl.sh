#!/bin/bash
cd /root/repo
for c in "200 256" "320 512" "320 768" "520 768" "520 1024"; do set -- $c; echo "splitk tiles<$1 target $2"; DVS_SPLITK_TILES=$1 DVS_SPLITK_TARGET=$2 timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_sk.txt 2>&1; grep -E "^== conv_(fwd|dgrad)" gpurun_out/per_launch_sk.txt; grep -n "^== conv_dgrad" -A66 gpurun_out/per_launch_sk.txt | grep -E "^[0-9]+-\s+(1|3|29|32|35|38|41|44)  work"; done
