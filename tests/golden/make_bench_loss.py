"""Oracle value of bench.py's first-step loss (committed as tests/golden/bench_loss.json).

bench.py checks the loss of its own seeded workload -- networks initialised under torch.manual_seed(0) on the CPU,
synth.throughput_sample(batch, 480, 640, rank=0), tie-break noise zero -- against this number before it times anything,
so the headline run is tied to the oracle (oracle/networks.py + oracle/loss_chain.py), not only the small parity shapes.

    python tests/golden/make_bench_loss.py          # ~2 minutes on 8 cores
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from deep_visual_slam_amd import synth  # noqa: E402
from deep_visual_slam_amd.depthnet import DepthNet  # noqa: E402
from deep_visual_slam_amd.posenet_single import PoseNet  # noqa: E402
from oracle import loss_chain as OL, networks as ON  # noqa: E402

H, W = 480, 640


def first_step_loss(batch, num_scales):
    torch.manual_seed(0)                                   # bench.build_gpu: same seed, same construction order
    sd_d = DepthNet(18, pretrained=False).state_dict()
    sd_p = PoseNet(18, pretrained=False, num_input_images=2).state_dict()
    sample = synth.throughput_sample(batch, H, W, rank=0)
    tgt, left, right = sample[("target_image", 0)], sample[("source_left", 0)], sample[("source_right", 0)]
    with torch.no_grad():
        disp = ON.depthnet(tgt, sd_d, train=True)
        aa_l, t_l = ON.posenet(torch.cat([left, tgt], 1), sd_p, train=True)
        aa_r, t_r = ON.posenet(torch.cat([tgt, right], 1), sd_p, train=True)
        noise = [torch.zeros(batch, 2, H, W) for _ in range(num_scales)]
        _, losses = OL.loss_chain(sample, [disp[("disp", s)] for s in range(num_scales)], (aa_l, t_l, aa_r, t_r), noise,
                                  num_scales=num_scales)
    return {k: float(v) for k, v in losses.items()}


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    out = {"c3": {"batch": 12, "num_scales": 4, "losses": first_step_loss(12, 4)},
           "c2": {"batch": 4, "num_scales": 1, "losses": first_step_loss(4, 1)},
           "note": "oracle (fp32, PyTorch-CPU) loss of bench.py's seeded first step, tie-break noise = 0"}
    with open(os.path.join(HERE, "bench_loss.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
