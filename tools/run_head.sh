#!/bin/bash
cd /root/repo
echo "default"; timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
echo "no final atomics"; DVS_LIB=/root/repo/deep-visual-slam_amd/csrc/build/variant_noatom.so timeout -k 10 300 python tools/head_bench.py 12 2>&1 | grep Cin
