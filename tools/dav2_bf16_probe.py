"""DA-V2 ViT-S forward (bench.dav2_side) with the token GEMMs / DPT convolutions in the bf16 mode and the attention products; prints the
frames/s of both modes and the relative difference of the depth output.  tools/dav2_bf16_probe.py"""
import json, sys
import torch
sys.path.insert(0, ".")
import bench
from deep_visual_slam_amd import _lib
from deep_visual_slam_amd.depth_anything_v2 import DepthAnythingV2
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = DepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384]).to(dev).eval()
x = torch.rand(8, 3, 518, 518, device=dev)
out = {}
with torch.no_grad():
    for mode in ("fp32", "bf16"):
        _lib.set_precision(mode)
        for _ in range(3):
            y = net(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            y = net(x)
        b.record()
        torch.cuda.synchronize()
        out[mode] = (a.elapsed_time(b) / 10, y.float().clone())
from deep_visual_slam_amd.depth_anything_v2 import attention
qkv = torch.randn(8 * 1370, 3 * 384, device=dev)
att = {}
for mode in ("fp32", "bf16"):
    _lib.set_precision(mode)
    for _ in range(3):
        attention(qkv, 8, 1370, 6, 64)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        attention(qkv, 8, 1370, 6, 64)
    b.record()
    torch.cuda.synchronize()
    att[mode] = a.elapsed_time(b) / 20
_lib.set_precision("fp32")
d = (out["bf16"][1] - out["fp32"][1]).abs().max() / out["fp32"][1].abs().max()
print(json.dumps({"batch": 8, "fp32_ms": out["fp32"][0], "bf16_ms": out["bf16"][0], "fp32_fps": 8e3 / out["fp32"][0], "bf16_fps": 8e3 / out["bf16"][0],
                  "rel_max_diff_of_depth": float(d),
                  "attention_ms": att}))
