#!/bin/bash
# bf16 attention forward: its tests and the DA-V2 probe (gpurun_out/)
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_dav2_gpu.py -x -q > gpurun_out/t_att.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/t_att.log
timeout -k 10 300 python tools/dav2_bf16_probe.py > gpurun_out/dav2_probe.json 2> gpurun_out/dav2_probe.err && cat gpurun_out/dav2_probe.json
