#!/bin/bash
# Rehearsal of the N > 1 code path of bench.py on a one-GPU box: two ranks on cuda:0, all-reduce over gloo (numbers mean nothing)
cd /root/repo
DVS_BENCH_REHEARSE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/rehearse2.json 2> gpurun_out/rehearse2.err; echo "rc=$?"
tail -c 600 gpurun_out/rehearse2.json; tail -3 gpurun_out/rehearse2.err
