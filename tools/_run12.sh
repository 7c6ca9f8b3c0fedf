cd /root/repo
timeout -k 10 1100 python bench.py > gpurun_out/bench_default_r3c.json 2> gpurun_out/bench_default_r3c.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_default_r3c.json')); print(d['value'], d['ms_per_step'], d['median_ms_per_step']); print(d['roofline']['frac'], d['roofline_chain']); print(d['stock_caller']['ms_per_step'], d['cpu_baseline']['value']); print({k:(v.get('ms_per_step'), v.get('value')) for k,v in d['other_configs'].items()})"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r3c.log 2>&1; tail -3 gpurun_out/smoke_r3c.log
