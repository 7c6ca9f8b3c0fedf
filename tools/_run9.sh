cd /root/repo
timeout -k 10 600 python -m pytest tests/test_wino_gpu.py tests/test_wgrad_ordered_gpu.py -x -q > gpurun_out/t_wino.log 2>&1 || { tail -40 gpurun_out/t_wino.log; exit 1; }
tail -2 gpurun_out/t_wino.log
timeout -k 10 300 python tools/wino_fixed_cost.py 12 2>&1 | grep shape
for B in 12 24; do WINO_SWEEP=64 WINO_WGRAD=1 timeout -k 10 300 python tools/wino_bench.py $B 2>&1 | grep wgrad | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('B=$B %-8s wino %.1f us (%.0f TF-eq) err %.1e  direct %.1f us' % (d['wgrad'], d['wino_ms']*1e3, d['wino_tf_eff'], d['err_wino'], d['direct_ms']*1e3))"; done
