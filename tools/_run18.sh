cd /root/repo
timeout -k 10 900 python -m pytest tests/test_caller_gpu.py tests/test_networks_gpu.py tests/test_dp_gpu.py tests/test_wgrad_ordered_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/t_host.log 2>&1 || { tail -40 gpurun_out/t_host.log; exit 1; }
tail -2 gpurun_out/t_host.log
timeout -k 10 200 python tools/host_profile_bwd.py 4 > gpurun_out/host_profile_bwd_b4.txt 2>&1; echo "hp rc=$?"; head -24 gpurun_out/host_profile_bwd_b4.txt
timeout -k 10 1100 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_f.json')); print('step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3)); c=d['other_configs']['configs[1]']; print({k:c[k] for k in c if k in ('ms_per_step','process','in_process','median_ms_per_step')}); print(d.get('stock_caller',{}).get('ms_per_step'))"
