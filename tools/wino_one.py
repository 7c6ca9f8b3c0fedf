"""Run one Winograd kernel repeatedly (for rocprofv3 --pmc).  usage: wino_one.py <l1|l2|l3|l4> <fwd|wgrad> [B]"""
import sys, torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
CL = torch.channels_last
dev = torch.device("cuda:0")
SH = {"l1": (64, 64, 120, 160), "l2": (128, 128, 60, 80), "l3": (256, 256, 30, 40), "l4": (512, 512, 15, 20)}
ci, co, h, w = SH[sys.argv[1]]
op = sys.argv[2] if len(sys.argv) > 2 else "fwd"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 12
x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL)
wt = (torch.randn(co, ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
dy = torch.randn(B, co, h, w, device=dev).contiguous(memory_format=CL)
sink = torch.zeros(co, ci, 3, 3, device=dev).contiguous(memory_format=CL)
for _ in range(8):
    if op == "fwd": DC.conv3x3_wino(x, wt)
    else: DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3), dw_out=sink)
torch.cuda.synchronize()
