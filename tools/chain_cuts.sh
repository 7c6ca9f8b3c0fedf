#!/bin/bash
# Where does chain_bwd_kernel's time go?  Variant libraries with one phase of the kernel cut out (-DCHAIN_DBG=<bits>: 1 no gathers in the
# staging, 2 no SSIM field phase, 4 no 3x3 field gather, 8 no chain phase, 16 no wave reductions; wrong results by construction), each
# timed alone with tools/chain_bench.py.  Build the variants HERE first (the .so files travel with gpurun), then run this on the box:
#   for d in 1 2 4 8 16; do python tools/build_variant.py cdbg$d --flag=-DCHAIN_DBG=$d --only=loss_chain.hip; done
#   gpurun -- 'bash tools/chain_cuts.sh "" cdbg1 cdbg2 cdbg4 cdbg8 cdbg16'
cd /root/repo
for v in "$@"; do
  lib=""; [ -n "$v" ] && lib=/root/repo/deep-visual-slam_amd/csrc/build/variant_$v.so
  echo "== ${v:-default}"
  DVS_LIB=$lib timeout -k 10 200 python tools/chain_bench.py 12 4 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: round(v['avg_ms'], 4) for k, v in d.items() if isinstance(v, dict)})"
done
