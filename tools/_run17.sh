cd /root/repo
timeout -k 10 300 python tools/host_profile_st.py 4 > gpurun_out/host_profile_st_b4.txt 2>&1; echo "rc=$?"
timeout -k 10 300 python tools/host_profile_st.py 4 cumtime > gpurun_out/host_profile_st_b4_cum.txt 2>&1; echo "rc=$?"
head -3 gpurun_out/host_profile_st_b4.txt
