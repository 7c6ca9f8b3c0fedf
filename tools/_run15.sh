cd /root/repo
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_d_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_d_pytest_gpu.log
timeout -k 10 200 python tools/host_profile.py 4 > gpurun_out/host_profile_b4.txt 2>&1; echo "hp rc=$?"
timeout -k 10 200 python tools/graph_probe.py 4 1 > gpurun_out/graph_probe_b4.txt 2>&1; echo "gp rc=$?"; tail -3 gpurun_out/graph_probe_b4.txt
