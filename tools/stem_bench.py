"""Timing of the stem (conv1) kernels alone: tools/stem_bench.py [B]"""
import sys, json, time
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ZERO = len(sys.argv) > 2 and sys.argv[2] == "zero"     # all-zero operands: same instruction stream, minimal switching power
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for ci in (3, 6):
    x = torch.rand(B, ci, 480, 640, device=dev)
    if ZERO: x.zero_()
    wt = (torch.randn(64, ci, 7, 7, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    sc = torch.full((ci,), 1 / 0.225, device=dev); sh = torch.full((ci,), -0.45 / 0.225, device=dev)
    y = DC.conv2d_forward(x, wt, None, 2, 3, False, None, in_scale=sc, in_shift=sh, nchw_planar=True)
    dy = torch.randn_like(y)
    if ZERO: dy.zero_(); wt.zero_(); sc.zero_(); sh.zero_()
    fl = 2.0 * B * 64 * 240 * 320 * ci * 49
    t_f = timeit(lambda: DC.conv2d_forward(x, wt, None, 2, 3, False, None, in_scale=sc, in_shift=sh, nchw_planar=True))
    t_w = timeit(lambda: DC.conv2d_wgrad(x, dy, tuple(wt.shape), 2, 3, False, False, in_scale=sc, in_shift=sh, nchw_planar=True))
    print(json.dumps(dict(cin=ci, fwd_ms=t_f * 1e3, fwd_tf=fl / t_f / 1e12, wgrad_ms=t_w * 1e3, wgrad_tf=fl / t_w / 1e12)), flush=True)
