cd /root/repo
timeout -k 10 600 python -m pytest tests/test_chain_gpu.py -x -q > gpurun_out/t_chain.log 2>&1 || { tail -40 gpurun_out/t_chain.log; exit 1; }
tail -2 gpurun_out/t_chain.log
for v in "" b256 "" b256; do
  lib=""; [ -n "$v" ] && lib=/root/repo/deep-visual-slam_amd/csrc/build/variant_$v.so
  echo "== ${v:-default 512}"
  DVS_LIB=$lib timeout -k 10 200 python tools/chain_bench.py 12 4 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: round(v['avg_ms'], 4) for k, v in d.items() if isinstance(v, dict)})"
done
