cd /root/repo
DVS_PRECISION=bf16 timeout -k 10 600 python tools/per_launch.py 12 4 4 > gpurun_out/per_launch_bf16.txt 2>&1; echo rc=$?
grep -c . gpurun_out/per_launch_bf16.txt
