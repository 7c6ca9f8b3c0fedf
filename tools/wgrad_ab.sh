#!/bin/bash
# A/B of the Winograd weight gradient on one box: correctness first (default build), then per-launch times of each variant library.
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_wino_gpu.py -x -q > gpurun_out/wgrad_ab_tests.log 2>&1 || { tail -30 gpurun_out/wgrad_ab_tests.log; exit 1; }
tail -3 gpurun_out/wgrad_ab_tests.log
for v in "" "$@"; do
  lib=""; [ -n "$v" ] && lib=/root/repo/deep-visual-slam_amd/csrc/build/variant_$v.so
  for B in 12 24; do
    echo "== variant '${v:-default}' B=$B" | tee -a gpurun_out/wgrad_ab.txt
    DVS_LIB=$lib WINO_SWEEP=64 WINO_WGRAD=1 timeout -k 10 300 python tools/wino_bench.py $B 2>&1 | grep wgrad | tee -a gpurun_out/wgrad_ab.txt || exit 1
  done
done
