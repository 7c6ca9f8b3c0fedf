cd /root/repo
bash tools/pmc_wino.sh l1 wgrad 12 > gpurun_out/pmc_l1_wgrad_wide.txt 2>&1; tail -10 gpurun_out/pmc_l1_wgrad_wide.txt
timeout -k 10 300 python tools/dec_wgrad_bench.py 12 > gpurun_out/dec_wgrad_r3.txt 2>&1; tail -12 gpurun_out/dec_wgrad_r3.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r3a.log 2>&1; tail -5 gpurun_out/pytest_gpu_r3a.log
