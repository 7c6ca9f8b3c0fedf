cd /root/repo
timeout -k 10 600 python -m pytest tests/test_wino_gpu.py -x -q > gpurun_out/t_wino.log 2>&1 || { tail -40 gpurun_out/t_wino.log; exit 1; }
tail -2 gpurun_out/t_wino.log
for k in 1 0; do DVS_WINO_KSPLIT=$k timeout -k 10 300 python tools/wino_bench.py 12 2>&1 | grep '"name"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('ksplit=$k %-8s fwd %.1f us (%.0f TF-eq) err %.1e stats %.1e dgrad err %.1e' % (d['name'], d['wino_ms']*1e3, d['wino_tf_eff'], d['err_wino'], d['err_stats'], d['err_dgrad_wino']))"; done
