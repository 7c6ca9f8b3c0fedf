"""Token GEMMs of the ViT-S blocks through the implicit-GEMM engine (1x1 convolutions on [1, K, 1, M] maps): TF/s per shape."""
import sys, json, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
dev = torch.device("cuda:0")
CL = torch.channels_last
def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
for M in (1370, 10960):
    for (K, N, act) in ((384, 1152, None), (384, 384, None), (384, 1536, "gelu"), (1536, 384, None), (608, 384, None)):
        x = torch.randn(1, K, 1, M, device=dev).contiguous(memory_format=CL)
        w = (torch.randn(N, K, 1, 1, device=dev) * 0.05).contiguous(memory_format=CL)
        b = torch.randn(N, device=dev) * 0.1
        t = timeit(lambda: DC.conv2d_forward(x, w, b, 1, 0, False, act))
        tm = timeit(lambda: F.linear(x.view(K, M).t(), w.view(N, K), b))
        fl = 2.0 * M * K * N
        print(json.dumps(dict(M=M, K=K, N=N, act=act, us=t * 1e6, tf=fl / t / 1e12, lib_us=tm * 1e6, lib_tf=fl / tm / 1e12)), flush=True)
