bash tools/pmc_wino.sh l1 wgrad 12 > gpurun_out/pmc_l1_wgrad_lean.txt 2>&1; tail -12 gpurun_out/pmc_l1_wgrad_lean.txt
DVS_LIB=/root/repo/deep-visual-slam_amd/csrc/build/variant_noslp.so WINO_SWEEP=64 WINO_WGRAD=1 timeout -k 10 300 python tools/wino_bench.py 12 2>&1 | grep wgrad > gpurun_out/wgrad_noslp_b12.txt; cat gpurun_out/wgrad_noslp_b12.txt
