cd /root/repo
timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_r3a.txt 2>&1; tail -3 gpurun_out/per_launch_r3a.txt
