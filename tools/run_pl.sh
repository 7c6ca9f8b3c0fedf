#!/bin/bash
# Per-launch table of one single-stream step on the GPU box (tools/per_launch.py) -> gpurun_out/per_launch_fp32.txt; with variant builds
# (tools/build_variant.py <name> --flag=-DTHIN_DBG=<bits> --only=conv_thin.hip ...) as arguments, one table per variant beside it:
#   gpurun -- bash tools/run_pl.sh [variant ...]
cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; echo "default rc=$?"
grep "^== " gpurun_out/per_launch_fp32.txt
for v in "$@"; do
    DVS_LIB=/root/repo/deep-visual-slam_amd/csrc/build/variant_$v.so timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_$v.txt 2>&1; echo "$v rc=$?"
    grep "^== " gpurun_out/per_launch_$v.txt
done
