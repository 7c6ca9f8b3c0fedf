#!/bin/bash
# fp32 per-launch table of the current build (gpurun_out/per_launch_fp32.txt)
cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; echo rc=$?
