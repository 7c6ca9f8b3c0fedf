"""A few Depth-Anything-V2 ViT-S forwards for rocprofv3 (per-kernel time of BASELINE configs[4]):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dav2 -- python3 tools/dav2_prof.py [batch] [forwards]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.build()
from deep_visual_slam_amd.depth_anything_v2 import DepthAnythingV2  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
net = DepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384]).cuda().eval()
x = torch.randn(B, 3, 518, 518, device="cuda")
with torch.no_grad():
    for _ in range(n):
        net(x)
torch.cuda.synchronize()
print("done", B, n)
