"""Run one conv shape repeatedly (for rocprofv3 --pmc).  usage: conv_one.py name [fwd|dgrad|wgrad]"""
import sys, torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
CL = torch.channels_last
dev = torch.device("cuda:0")
SH = {"l1": (12, 64, 64, 3, 1, 1, 120, 160), "l2": (12, 128, 128, 3, 1, 1, 60, 80), "l3": (12, 256, 256, 3, 1, 1, 30, 40),
      "l4": (12, 512, 512, 3, 1, 1, 15, 20), "l2s2": (12, 64, 128, 3, 2, 1, 120, 160), "up0": (12, 16, 16, 3, 1, 1, 480, 640),
      "dec2": (12, 128, 64, 3, 1, 1, 120, 160), "dec1": (12, 96, 32, 3, 1, 1, 240, 320)}   # decoder: reflect pad + ELU
B, ci, co, k, s, p, h, w = SH[sys.argv[1]]
op = sys.argv[2] if len(sys.argv) > 2 else "fwd"
x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL)
wt = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=CL)
dec = sys.argv[1].startswith("dec")
bias = torch.zeros(co, device=dev) if dec else None
y = DC.conv2d_forward(x, wt, bias, s, p, dec, "elu" if dec else None)
dy = torch.randn_like(y)
for _ in range(8):
    if op == "fwd": DC.conv2d_forward(x, wt, bias, s, p, dec, "elu" if dec else None)
    elif op == "dgrad": DC.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, dec, y if dec else None, "elu" if dec else None)
    else: DC.conv2d_wgrad(x, dy, tuple(wt.shape), s, p, dec, dec, y if dec else None, "elu" if dec else None)
torch.cuda.synchronize()
