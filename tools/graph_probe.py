"""Feasibility probe: capture one whole training step (4 streams, autograd, Adam) in a HIP graph and replay it."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from deep_visual_slam_amd import gradsink

dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scales = int(sys.argv[2]) if len(sys.argv) > 2 else 1
trainer, flat, sync, opt, sample = bench.build_gpu(batch, scales, dev, 0)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4):
        losses = bench.gpu_step(trainer, sync, opt, sample)
torch.cuda.synchronize()
t = time.perf_counter()
with torch.cuda.stream(s):
    for _ in range(20):
        losses = bench.gpu_step(trainer, sync, opt, sample)
torch.cuda.synchronize()
print("eager: %.2f ms/step, loss %.6f" % ((time.perf_counter() - t) / 20 * 1e3, float(losses["loss"])))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    static_losses = bench.gpu_step(trainer, sync, opt, sample)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graph replay: %.2f ms/step, loss %.6f" % ((time.perf_counter() - t) / 20 * 1e3, float(static_losses["loss"])))
