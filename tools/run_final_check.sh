#!/bin/bash
# End-of-round check on the GPU box: full `-m gpu` suite, the driver's bench command, smoke (outputs under gpurun_out/)
cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/final_pytest_gpu.log
date +%s > gpurun_out/t0
timeout -k 10 1100 python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - $(cat gpurun_out/t0) ))s"
python3 -c "
import json; d=json.load(open('gpurun_out/final_bench_default.json')); print(d['value'], d['ms_per_step'], d['median_ms_per_step']); print(d['roofline']['frac'], d['roofline_chain']['chain_bwd_kernel']['ms_per_step']); print(d['stock_caller']['ms_per_step'], d['cpu_baseline']['value']); print({k:(v.get('ms_per_step'), v.get('value')) for k,v in d['other_configs'].items()})"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; tail -3 gpurun_out/final_smoke.log
