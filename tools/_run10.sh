cd /root/repo
timeout -k 10 900 python -m pytest tests/test_networks_gpu.py tests/test_fullsize_gpu.py tests/test_caller_gpu.py tests/test_dp_gpu.py -x -q > gpurun_out/t_nets.log 2>&1 || { tail -40 gpurun_out/t_nets.log; exit 1; }
tail -2 gpurun_out/t_nets.log
for o in 1 0 1 0; do DVS_DS_STREAM=$o timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-kernel-timing > gpurun_out/bench_ds$o.json 2> gpurun_out/bench_ds$o.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_ds$o.json')); print('ds_stream=$o', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3))"; done
