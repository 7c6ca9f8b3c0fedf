cd /root/repo
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_e_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_e_pytest_gpu.log
