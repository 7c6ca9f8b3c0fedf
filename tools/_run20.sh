cd /root/repo
timeout -k 10 900 python -m pytest tests/test_chain_gpu.py tests/test_pipeline_gpu.py tests/test_caller_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q > gpurun_out/t_chain.log 2>&1 || { tail -40 gpurun_out/t_chain.log; exit 1; }
tail -2 gpurun_out/t_chain.log
for i in 1 2; do timeout -k 10 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_g$i.json 2> gpurun_out/bench_g$i.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_g$i.json')); print('step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k:v for k,v in d['kernels_ms_per_step'].items() if 'chain' in k}, d.get('roofline_chain'))"; done
bash tools/pmc_chain.sh > gpurun_out/pmc_chain_r3.log 2>&1; tail -34 gpurun_out/pmc_chain_r3.log
