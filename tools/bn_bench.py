import sys, time, torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import _lib
l = _lib.lib()
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for (M, C) in ((12*240*320, 64), (12*120*160, 64), (12*60*80, 128), (12*30*40, 256), (12*15*20, 512)):
    dz = torch.randn(M, C, device=dev); z = torch.randn(M, C, device=dev); y = torch.randn(M, C, device=dev)
    du = torch.empty_like(y); dy = torch.empty_like(y)
    mean = torch.zeros(C, device=dev); inv = torch.ones(C, device=dev); g = torch.ones(C, device=dev)
    sums = torch.zeros(2, C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(l.dvs_bn_bwd_workspace(M, C, 1) // 4, device=dev)
    t_red = timeit(lambda: l.dvs_bn_bwd_reduce(dz.data_ptr(), z.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(), du.data_ptr(), sums.data_ptr(), ws.data_ptr(), M, C, 1, st))
    t_noat = 0.0
    t_app = timeit(lambda: l.dvs_bn_bwd_apply(du.data_ptr(), y.data_ptr(), mean.data_ptr(), inv.data_ptr(), g.data_ptr(), sums.data_ptr(), dy.data_ptr(), M, C, None, None, 1, st))
    t_fwd = timeit(lambda: l.dvs_bn_apply_fwd(y.data_ptr(), mean.data_ptr(), inv.data_ptr(), z.data_ptr(), None, None, dy.data_ptr(), M, C, 1, 1, st))
    by = M * C * 4
    print("M=%7d C=%3d  reduce %7.1f us (%.2f TB/s)  no-atomics %7.1f us  apply %7.1f us (%.2f TB/s)  fwd %7.1f us (%.2f TB/s)" % (
        M, C, t_red * 1e6, 4 * by / t_red / 1e12, t_noat * 1e6, t_app * 1e6, 3 * by / t_app / 1e12, t_fwd * 1e6, 3 * by / t_fwd / 1e12))
