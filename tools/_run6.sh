cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r3a.log 2>&1; tail -5 gpurun_out/pytest_gpu_r3a.log
