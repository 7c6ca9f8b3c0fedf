"""Per-layer times of the bf16 patch kernels (forward, data gradient, weight gradient) at the encoder shapes.  tools/p16_bench.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import _lib, conv as DC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
_lib.set_precision("bf16")
CL = torch.channels_last
for name, c, H, W in (("l1", 64, 120, 160), ("l2", 128, 60, 80), ("l3", 256, 30, 40), ("l4", 512, 15, 20)):
    x = torch.randn(B, c, H, W, device=dev).contiguous(memory_format=CL)
    dy = torch.randn(B, c, H, W, device=dev).contiguous(memory_format=CL)
    w = (torch.randn(c, c, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
    dw = torch.zeros_like(w)
    fns = {"fwd": lambda: DC.conv3x3_p16(x, w), "dgrad": lambda: DC.conv3x3_p16(dy, w, flip=True),
           "wgrad": lambda: DC.conv3x3_p16_wgrad(x, dy, tuple(w.shape), dw_out=dw)}
    out = []
    for k, f in fns.items():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        gf = 2.0 * B * H * W * c * c * 9 / 1e9
        out.append("%s %.1f us (%.0f TF)" % (k, us, gf / us * 1e3))            # 1e9 flop / 1e-6 s = 1e15 flop/s = 1e3 TF
    print("B=%d %s %d ch %dx%d: %s   [bytes x+y %.0f MB]" % (B, name, c, H, W, ", ".join(out), 2 * B * H * W * c * 4 / 1e6))
