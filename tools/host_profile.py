"""cProfile of the host side of a few training steps (where do the ~11 ms of Python issue time go?)."""
import cProfile, pstats, sys, io
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tr, flat, sync, opt, sample = bench.build_gpu(B, 1 if B == 4 else 4, dev, 0)
for _ in range(5):
    bench.gpu_step(tr, sync, opt, sample)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.gpu_step(tr, sync, opt, sample)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
