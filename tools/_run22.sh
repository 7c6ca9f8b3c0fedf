cd /root/repo
for pr in bf16 fp32; do timeout -k 10 600 python bench.py --precision $pr --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_$pr.json 2> gpurun_out/bench_$pr.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_$pr.json')); print('$pr step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['loss_check']); print({k:round(v,3) for k,v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv:-kv[1])[:14]})"; done
