cd /root/repo
timeout -k 10 600 python -m pytest tests/test_wino_gpu.py -x -q > gpurun_out/t_wino.log 2>&1 || { tail -40 gpurun_out/t_wino.log; exit 1; }
tail -2 gpurun_out/t_wino.log
for pz in 1 0; do echo "PERSIST=$pz"; DVS_WINO_PERSIST=$pz timeout -k 10 300 python tools/wino_fixed_cost.py 12 2>&1 | grep shape | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['shape'], 'fixed/wg', d['fixed_us_per_wg'], 'kstep/wg', d['us_per_kstep_per_wg'], d['points'])"; done
for pz in 1 0 1 0; do DVS_WINO_PERSIST=$pz timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-kernel-timing > gpurun_out/bench_p$pz.json 2> gpurun_out/bench_p$pz.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_p$pz.json')); print('persist=$pz', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))"; done
