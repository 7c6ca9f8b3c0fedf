#!/bin/bash
# Thin decoder layers: weight-gradient bench + the convolution tests (gpurun_out/)
cd /root/repo
timeout -k 10 300 python tools/thin_bench.py 12 2>&1 | grep name
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py -x -q > gpurun_out/t_thin.log 2>&1; echo rc=$?; tail -3 gpurun_out/t_thin.log
