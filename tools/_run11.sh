cd /root/repo
timeout -k 10 600 python -m pytest tests/test_dav2_gpu.py -x -q > gpurun_out/t_dav2.log 2>&1 || { tail -30 gpurun_out/t_dav2.log; exit 1; }
tail -2 gpurun_out/t_dav2.log
timeout -k 10 600 python -c "
import json, torch, bench
print(json.dumps(bench.dav2_side(torch.device('cuda:0'), False), indent=1))" > gpurun_out/dav2_side.json 2>gpurun_out/dav2_side.err; python3 -c "import json; d=json.load(open(\"gpurun_out/dav2_side.json\")); print({k:(v[\"attention\"][\"roofline\"][\"frac\"], v[\"gemm_and_conv\"][\"roofline\"][\"frac\"], v[\"ms_per_forward\"]) for k,v in d[\"batches\"].items()}); print(d[\"train\"])"
