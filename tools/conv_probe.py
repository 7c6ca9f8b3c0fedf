"""Probe: what does PyTorch-ROCm (MIOpen) deliver for the path's fp32 convolutions on this box?
Used only to size the hand-written MFMA conv work (not part of the product path)."""
import sys, time, json
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes = [  # (name, Cin, Cout, k, stride, pad, Hin, Win)
    ("conv1_3ch", 3, 64, 7, 2, 3, 480, 640), ("conv1_6ch", 6, 64, 7, 2, 3, 480, 640),
    ("l1_3x3", 64, 64, 3, 1, 1, 120, 160), ("l2_s2", 64, 128, 3, 2, 1, 120, 160), ("l2_3x3", 128, 128, 3, 1, 1, 60, 80),
    ("l3_s2", 128, 256, 3, 2, 1, 60, 80), ("l3_3x3", 256, 256, 3, 1, 1, 30, 40),
    ("l4_s2", 256, 512, 3, 2, 1, 30, 40), ("l4_3x3", 512, 512, 3, 1, 1, 15, 20),
    ("up4_0", 512, 256, 3, 1, 0, 17, 22), ("up4_1", 512, 256, 3, 1, 0, 32, 42), ("up3_1", 256, 128, 3, 1, 0, 62, 82),
    ("up2_1", 128, 64, 3, 1, 0, 122, 162), ("up1_1", 96, 32, 3, 1, 0, 242, 322), ("up0_0", 32, 16, 3, 1, 0, 242, 322),
    ("up0_1", 16, 16, 3, 1, 0, 482, 642), ("disp0", 16, 1, 3, 1, 0, 482, 642),
]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
res = []
for cl in (False, True):
    for (name, ci, co, k, s, p, h, w) in shapes:
        x = torch.randn(B, ci, h, w, device=dev, requires_grad=True)
        wt = torch.randn(co, ci, k, k, device=dev, requires_grad=True)
        if cl:
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wt = wt.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = F.conv2d(x, wt, None, s, p)
        gy = torch.randn_like(y)
        ho, wo = y.shape[2:]
        fl = 2.0 * B * co * ho * wo * ci * k * k
        tf = timeit(lambda: F.conv2d(x, wt, None, s, p))
        def bw():
            y = F.conv2d(x, wt, None, s, p)
            torch.autograd.grad(y, (x, wt), gy)
        tb = timeit(bw) - tf
        r = dict(name=name, cl=cl, B=B, gflop=fl / 1e9, fwd_ms=tf * 1e3, fwd_tf=fl / tf / 1e12, bwd_ms=tb * 1e3, bwd_tf=2 * fl / tb / 1e12)
        res.append(r); print(json.dumps(r), flush=True)
# HBM copy + fp32 GEMM yardsticks
a = torch.empty(1 << 28, device=dev); b_ = torch.empty_like(a)
t = timeit(lambda: b_.copy_(a)); print(json.dumps(dict(name="copy_1GiB", GBps=2 * a.numel() * 4 / t / 1e9)), flush=True)
m = torch.randn(8192, 8192, device=dev); n_ = torch.randn(8192, 8192, device=dev)
t = timeit(lambda: m @ n_); print(json.dumps(dict(name="sgemm_8192", TF=2 * 8192**3 / t / 1e12)), flush=True)
