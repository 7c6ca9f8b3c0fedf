cd /root/repo
timeout -k 10 900 python -m pytest tests/test_chain_gpu.py tests/test_pipeline_gpu.py tests/test_bf16_gpu.py -x -q > gpurun_out/t_chain.log 2>&1 || { tail -40 gpurun_out/t_chain.log; exit 1; }
tail -2 gpurun_out/t_chain.log
bash tools/profile_bench.sh c3 > gpurun_out/prof_c3_r3f.txt 2>&1; echo "rc=$?"; grep -E "per step|chain_bwd_kernel|chain_fwd_kernel|conv_wgrad_kernel " gpurun_out/prof_c3_r3f.txt
for i in 1 2; do timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_h$i.json 2> gpurun_out/bench_h$i.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_h$i.json')); print('step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k:v for k,v in d['kernels_ms_per_step'].items() if 'chain' in k})"; done
