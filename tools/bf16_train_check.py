"""Does the bf16 mode TRAIN like fp32?  The same seeded networks and the same synthetic batch stream for N optimiser steps in both
modes (dp.FusedAdam, lr 1e-4); prints the loss trajectories and their relative difference.  tools/bf16_train_check.py [steps] [batch]"""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench
from deep_visual_slam_amd import _lib, gradsink, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
out = {}
for mode in ("fp32", "bf16"):
    _lib.set_precision(mode)
    gradsink.reset_streams()
    trainer, flat, sync, opt, _ = bench.build_gpu(batch, 4, dev, 0)
    losses = []
    for it in range(steps):
        # five different textured batches in turn (sources = the target texture shifted: a scene with real parallax)
        sample = {k: v.to(dev) for k, v in synth.parity_sample(batch, bench.H, bench.W, seed=2024 + it % 5).items()}
        _, l = trainer.process_batch(sample)
        l["loss"].backward()
        sync.finish()
        opt.step(grad_scale=sync.grad_scale, zero_grad=True)
        losses.append(float(l["loss"]))
    out[mode] = losses
    del trainer, flat, sync, opt
_lib.set_precision("fp32")
rel = [abs(a - b) / abs(a) for a, b in zip(out["fp32"], out["bf16"])]
print(json.dumps({"steps": steps, "batch": batch, "fp32": [round(v, 6) for v in out["fp32"]], "bf16": [round(v, 6) for v in out["bf16"]],
                  "max_rel_diff": max(rel), "mean_rel_diff": sum(rel) / len(rel),
                  "fp32_first_last": [out["fp32"][0], out["fp32"][-1]], "bf16_first_last": [out["bf16"][0], out["bf16"][-1]]}))
