#!/bin/bash
# Where does chain_bwd_kernel's time go?  Variant libraries with parts of the kernel cut out (CHAIN_DBG bits, wrong results), timed alone.
cd /root/repo
for v in b256 cdbg1 cdbg2 cdbg8 cdbg16 cdbg24; do
  lib=""; [ -n "$v" ] && lib=/root/repo/deep-visual-slam_amd/csrc/build/variant_$v.so
  echo "== ${v:-default}"
  DVS_LIB=$lib timeout -k 10 200 python tools/chain_bench.py 12 4 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: round(v['avg_ms'], 4) for k, v in d.items() if isinstance(v, dict)})"
done
