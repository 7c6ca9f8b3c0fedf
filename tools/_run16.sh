cd /root/repo
timeout -k 10 600 python -m pytest tests/test_wino_gpu.py tests/test_conv_gpu.py -x -q > gpurun_out/t_wino.log 2>&1 || { tail -40 gpurun_out/t_wino.log; exit 1; }
tail -2 gpurun_out/t_wino.log
timeout -k 10 300 python tools/wino_fixed_cost.py 12 2>&1 | grep shape | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['shape'], 'fixed/wg', d['fixed_us_per_wg'], 'kstep/wg', d['us_per_kstep_per_wg'], d['points'])"
timeout -k 10 200 python tools/host_profile_bwd.py 4 > gpurun_out/host_profile_bwd_b4.txt 2>&1; echo "hp rc=$?"
for i in 1 2; do timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-kernel-timing > gpurun_out/bench_e$i.json 2> gpurun_out/bench_e$i.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_e$i.json')); print('epilogue', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))"; done
