#!/bin/bash
cd /root/repo
echo "default"; timeout -k 10 300 python tools/thin_bench.py 12 2>&1 | grep "up0"
for v in 1 2 4 8 5 13; do echo "THIN_DBG $v"; DVS_LIB=/root/repo/deep-visual-slam_amd/csrc/build/variant_tdbg$v.so timeout -k 10 300 python tools/thin_bench.py 12 2>&1 | grep "up0"; done
