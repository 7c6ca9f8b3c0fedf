"""Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) against the direct implicit-GEMM kernel and F.conv2d on the stride-1 3x3
shapes of the ResNet encoder: max error of each against an fp64 convolution, and time.  usage: wino_bench.py [B]"""
import sys, json, time
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC, _lib
from deep_visual_slam_amd._lib import check, ptr

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CL = torch.channels_last
shapes = [("l1_3x3", 64, 64, 120, 160), ("l2_3x3", 128, 128, 60, 80), ("l3_3x3", 256, 256, 30, 40), ("l4_3x3", 512, 512, 15, 20),
          ("odd", 32, 48, 13, 27)]


def timeit(fn, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def wino_weights(w, flip=False):
    co, ci = w.shape[:2]
    wl = w.permute(0, 2, 3, 1).contiguous()                     # [Cout][3][3][Cin]
    u = torch.empty((co if flip else ci) * (ci if flip else co) * 16, device=w.device)
    check(_lib.lib().dvs_wino_weights(wl.data_ptr(), u.data_ptr(), co, ci, int(flip), _lib.stream()), "dvs_wino_weights")
    return u


def wino_fwd(x, u, cout, bias=None, relu=False, stats=None, groups=0, y=None):
    b, ci, h, w = x.shape
    if y is None:
        y = torch.empty((b, cout, h, w), device=x.device, memory_format=CL)
    check(_lib.lib().dvs_conv3x3_wino_fwd(x.data_ptr(), u.data_ptr(), ptr(bias), None, y.data_ptr(), ptr(stats), groups, b, h, w, ci, cout,
                                          int(relu), 0, _lib.stream()), "dvs_conv3x3_wino_fwd")
    return y


torch.manual_seed(0)
import os
if os.environ.get("WINO_SWEEP"):
    shapes = []
for (name, ci, co, h, w) in shapes:
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL)
    wt = (torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (ci * 9)) ** 0.5).contiguous(memory_format=CL)
    y64 = F.conv2d(x.double(), wt.double(), None, 1, 1)
    den = float(y64.abs().max())
    u = wino_weights(wt)
    st = torch.zeros(2, 2, co, device=dev)
    yw = wino_fwd(x, u, co, stats=st, groups=2)
    yd = DC.conv2d_forward(x, wt, None, 1, 1, False, None)
    ym = F.conv2d(x, wt, None, 1, 1)
    e = lambda y: float((y.double() - y64).abs().max()) / den
    s_ref = torch.stack([y64[: B // 2].sum((0, 2, 3)), (y64[: B // 2] ** 2).sum((0, 2, 3))])
    s_err = float((st[0].double() - s_ref).abs().max() / s_ref.abs().max())
    # data gradient of the same convolution through the flipped filter
    dy = torch.randn(B, co, h, w, device=dev).contiguous(memory_format=CL)
    dx64 = F.conv_transpose2d(dy.double(), wt.double(), None, 1, 1)
    uf = wino_weights(wt, flip=True)
    dxw = wino_fwd(dy, uf, ci)
    e_dx = float((dxw.double() - dx64).abs().max() / dx64.abs().max())
    fl = 2.0 * B * co * h * w * ci * 9
    t_w = timeit(lambda: wino_fwd(x, u, co, y=yw))
    t_d = timeit(lambda: DC.conv2d_forward(x, wt, None, 1, 1, False, None))
    t_m = timeit(lambda: F.conv2d(x, wt, None, 1, 1))
    print(json.dumps(dict(name=name, B=B, err_wino=e(yw), err_direct=e(yd), err_miopen=e(ym), err_stats=s_err, err_dgrad_wino=e_dx,
                          gflop=fl / 1e9, wino_ms=t_w * 1e3, wino_tf_eff=fl / t_w / 1e12, direct_ms=t_d * 1e3,
                          direct_tf=fl / t_d / 1e12, miopen_ms=t_m * 1e3, miopen_tf=fl / t_m / 1e12)), flush=True)

# fixed cost vs per-k-step cost: the l2 geometry with a varying number of input channels
import os
if os.environ.get("WINO_SWEEP"):
    for ci in [int(v) for v in os.environ["WINO_SWEEP"].split(",")]:
        x = torch.randn(B, ci, 60, 80, device=dev).contiguous(memory_format=CL)
        wt = torch.randn(128, ci, 3, 3, device=dev).contiguous(memory_format=CL)
        u = wino_weights(wt)
        y = torch.empty((B, 128, 60, 80), device=dev).contiguous(memory_format=CL)
        l = _lib.lib(); st = _lib.stream()
        def go():
            l.dvs_conv3x3_wino_fwd(x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), None, 0, B, 60, 80, ci, 128, 0, 0, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5): go()
        e0.record()
        for _ in range(50): go()
        e1.record(); torch.cuda.synchronize()
        print(json.dumps(dict(sweep_cin=ci, ksteps=ci // 2, us=e0.elapsed_time(e1) * 1e3 / 50)), flush=True)

# weight gradient: Winograd vs the direct kernel vs fp64 autograd
if os.environ.get("WINO_WGRAD"):
    shapes_w = [("l1_3x3", 64, 64, 120, 160), ("l2_3x3", 128, 128, 60, 80), ("l3_3x3", 256, 256, 30, 40), ("l4_3x3", 512, 512, 15, 20),
                ("odd", 64, 96, 13, 27)]
    for (name, ci, co, h, w) in shapes_w:
        x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL)
        dy = torch.randn(B, co, h, w, device=dev).contiguous(memory_format=CL)
        w64 = torch.zeros(co, ci, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double(), w64, None, 1, 1).backward(dy.double())
        ref = w64.grad
        dw = DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3))
        dd, _ = DC.conv2d_wgrad(x, dy, (co, ci, 3, 3), 1, 1, False, False)
        e = lambda t: float((t.double() - ref).abs().max() / ref.abs().max())
        sink = torch.zeros(co, ci, 3, 3, device=dev).contiguous(memory_format=CL)
        t_w = timeit(lambda: DC.conv3x3_wino_wgrad(x, dy, (co, ci, 3, 3), dw_out=sink))
        t_d = timeit(lambda: DC.conv2d_wgrad(x, dy, (co, ci, 3, 3), 1, 1, False, False, dw_out=sink))
        fl = 2.0 * B * co * h * w * ci * 9
        print(json.dumps(dict(wgrad=name, B=B, err_wino=e(dw), err_direct=e(dd), wino_ms=t_w * 1e3, wino_tf_eff=fl / t_w / 1e12,
                              direct_ms=t_d * 1e3, direct_tf=fl / t_d / 1e12)), flush=True)
