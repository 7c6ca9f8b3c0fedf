#!/bin/bash
cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_fp32.txt 2>&1; echo rc=$?
grep -n "^== conv_dgrad" -A66 gpurun_out/per_launch_fp32.txt | grep -E "^[0-9]+-\s+(5|6|10|11|15|16|20|21|22|23|46|47|51|52|56|57|61|62|63|64)  work"
timeout -k 10 900 python -m pytest tests/test_wino_gpu.py tests/test_conv_gpu.py -x -q > gpurun_out/t_wino.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_wino.log
