"""Per-shape timing of the hand-written data-gradient / weight-gradient kernels vs MIOpen (torch autograd)."""
import sys, json, time
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CL = torch.channels_last
shapes = [
    ("l1_3x3", 64, 64, 3, 1, 1, 0, 120, 160), ("l2_s2", 64, 128, 3, 2, 1, 0, 120, 160),
    ("l2_ds", 64, 128, 1, 2, 0, 0, 120, 160), ("l2_3x3", 128, 128, 3, 1, 1, 0, 60, 80),
    ("l3_s2", 128, 256, 3, 2, 1, 0, 60, 80), ("l3_3x3", 256, 256, 3, 1, 1, 0, 30, 40),
    ("l4_s2", 256, 512, 3, 2, 1, 0, 30, 40), ("l4_3x3", 512, 512, 3, 1, 1, 0, 15, 20),
    ("up4_0", 512, 256, 3, 1, 1, 1, 15, 20), ("up4_1", 512, 256, 3, 1, 1, 1, 30, 40), ("up3_0", 256, 128, 3, 1, 1, 1, 30, 40),
    ("up3_1", 256, 128, 3, 1, 1, 1, 60, 80), ("up2_0", 128, 64, 3, 1, 1, 1, 60, 80), ("up2_1", 128, 64, 3, 1, 1, 1, 120, 160),
    ("up1_0", 64, 32, 3, 1, 1, 1, 120, 160), ("up1_1", 96, 32, 3, 1, 1, 1, 240, 320),
    ("up0_0", 32, 16, 3, 1, 1, 1, 240, 320), ("up0_1", 16, 16, 3, 1, 1, 1, 480, 640),
    ("pose_sq", 512, 256, 1, 1, 0, 0, 15, 20), ("pose_0", 256, 256, 3, 1, 1, 0, 15, 20),
]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
torch.manual_seed(0)
tot = dict(dg=0, wg=0, mi=0)
for (name, ci, co, k, s, p, refl, h, w) in shapes:
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL).requires_grad_(True)
    wt = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=CL).requires_grad_(True)
    xx = F.pad(x, (p,) * 4, mode="reflect") if refl else x
    y = F.conv2d(xx, wt, None, s, 0 if refl else p)
    dy = torch.randn_like(y)
    fl = 2.0 * B * co * y.shape[2] * y.shape[3] * ci * k * k
    t_dg = timeit(lambda: DC.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, bool(refl)))
    t_wg = timeit(lambda: DC.conv2d_wgrad(x, dy, tuple(wt.shape), s, p, bool(refl), False))
    t_mi = timeit(lambda: torch.autograd.grad(y, (x, wt), dy, retain_graph=True))
    tot["dg"] += t_dg; tot["wg"] += t_wg; tot["mi"] += t_mi
    print(json.dumps(dict(name=name, gflop=fl / 1e9, dgrad_ms=t_dg * 1e3, dgrad_tf=fl / t_dg / 1e12, wgrad_ms=t_wg * 1e3,
                          wgrad_tf=fl / t_wg / 1e12, miopen_both_ms=t_mi * 1e3, miopen_tf=2 * fl / t_mi / 1e12)), flush=True)
for ci in (3, 6):
    x = torch.rand(B, ci, 480, 640, device=dev)
    wt = (torch.randn(64, ci, 7, 7, device=dev) * 0.05).requires_grad_(True)
    y = F.conv2d(x, wt, None, 2, 3); dy = torch.randn_like(y)
    fl = 2.0 * B * 64 * 240 * 320 * ci * 49
    sc = torch.full((ci,), 1 / 0.225, device=dev); sh = torch.full((ci,), -0.45 / 0.225, device=dev)
    t_wg = timeit(lambda: DC.conv2d_wgrad(x, dy, tuple(wt.shape), 2, 3, False, False, in_scale=sc, in_shift=sh, nchw_planar=True))
    t_mi = timeit(lambda: torch.autograd.grad(y, (wt,), dy, retain_graph=True))
    print(json.dumps(dict(name="conv1_%dch" % ci, gflop=fl / 1e9, wgrad_ms=t_wg * 1e3, wgrad_tf=fl / t_wg / 1e12,
                          miopen_wgrad_ms=t_mi * 1e3)), flush=True)
print(json.dumps({k: v * 1e3 for k, v in tot.items()}))
