cd /root/repo
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/t_bf16.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_bf16.log
for db in 600 0 2000; do DVS_BF16_DEEP_BELOW=$db timeout -k 10 600 python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_bf16_d$db.json 2> gpurun_out/bench_bf16_d$db.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_bf16_d$db.json')); print('deep_below=$db bf16 step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3)); print({k:round(v,3) for k,v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv:-kv[1])[:4]})"; done
