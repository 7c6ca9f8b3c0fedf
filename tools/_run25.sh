cd /root/repo
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/t_bf16.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t_bf16.log
for st in 1 0; do DVS_BF16_STEM=$st timeout -k 10 600 python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/bench_bf16_s$st.json 2> gpurun_out/bench_bf16_s$st.err; python3 -c "
import json; d=json.load(open('gpurun_out/bench_bf16_s$st.json')); print('stem16=$st bf16 step', round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['loss_check']['worst_rel_err']); print({k:round(v,3) for k,v in sorted(d['kernels_ms_per_step'].items(), key=lambda kv:-kv[1])[:5]})"; done
DVS_PRECISION=bf16 timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_bf16.txt 2>&1; echo rc=$?
