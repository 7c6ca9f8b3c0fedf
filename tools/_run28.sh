cd /root/repo
bash tools/profile_bench.sh c3 > gpurun_out/prof_c3_r3f.txt 2>&1; echo "rc=$?"; tail -45 gpurun_out/prof_c3_r3f.txt
