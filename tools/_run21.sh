cd /root/repo
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q > gpurun_out/t_bf16.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/t_bf16.log
