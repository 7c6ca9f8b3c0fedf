"""Where a training step's time goes with the four streams on, measured with HIP events (no profiler: its
per-launch overhead makes the host the bottleneck and changes the overlap).  tools/step_timeline.py [batch]"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from deep_visual_slam_amd import gradsink

dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 12
trainer, flat, sync, opt, sample = bench.build_gpu(batch, 4, dev, 0)
trainer._timeline = True
main = torch.cuda.current_stream()
pose = trainer.pose_stream
ev = lambda: torch.cuda.Event(enable_timing=True)

def step(marks=None):
    if marks is not None:
        marks["t0"] = ev(); marks["t0"].record(main)
    outputs, losses = trainer.process_batch(sample)
    if marks is not None:
        marks["fwd_end"] = ev(); marks["fwd_end"].record(main)
        marks.update(getattr(trainer, "_marks", {}))
    losses["loss"].backward()
    if marks is not None:
        marks["bwd_main"] = ev(); marks["bwd_main"].record(main)
    sync.finish()
    opt.step(grad_scale=sync.grad_scale, zero_grad=True)
    if marks is not None:
        marks["end"] = ev(); marks["end"].record(main)

for _ in range(5):
    step()
torch.cuda.synchronize()
# host-side cost of issuing one step (GPU idle at the start: pure launch time when the GPU is not the limiter)
t = time.perf_counter(); step(); t_issue = time.perf_counter() - t
torch.cuda.synchronize()
rows = []
t = time.perf_counter()
for _ in range(10):
    m = {}
    step(m)
    rows.append(m)
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / 10
print("host time to issue one step (GPU idle): %.1f ms; steady-state wall %.1f ms/step" % (t_issue * 1e3, wall * 1e3))
for m in rows[-3:]:
    if "pose_done" in m:
        print("  forward: PoseNet done at %.2f ms, DepthNet done at %.2f ms, chain forward done at %.2f ms" % (
            m["t0"].elapsed_time(m["pose_done"]), m["t0"].elapsed_time(m["depth_done"]), m["t0"].elapsed_time(m["fwd_end"])))
    print("  step: forward+chain %.2f ms | backward %.2f ms | join+adam %.2f ms | total %.2f ms" % (
        m["t0"].elapsed_time(m["fwd_end"]), m["fwd_end"].elapsed_time(m["bwd_main"]),
        m["bwd_main"].elapsed_time(m["end"]), m["t0"].elapsed_time(m["end"])))
