"""Weight gradient of the decoder's wide Conv3x3 layers (ReflectionPad2d(1) + [upsample + concat] + 3x3): the Winograd kernel's
reflect / upsample gathers (dvs_conv3x3_wino_wgrad_gen) against the LDS-DMA implicit-GEMM kernel (dvs_conv2d_wgrad), error of
each against fp64 autograd and time; plus the zero-padded kernel on a BasicBlock shape (regression check of MODE 0).
usage: dec_wgrad_bench.py [B]"""
import sys, json
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CL = torch.channels_last
# name, C1, C2 (-1: no upsample), Cout, h, w of x
LAYERS = [("upconv_4_0", 512, -1, 256, 15, 20), ("upconv_4_1", 256, 256, 256, 15, 20), ("upconv_3_0", 256, -1, 128, 30, 40),
          ("upconv_3_1", 128, 128, 128, 30, 40), ("upconv_2_0", 128, -1, 64, 60, 80), ("upconv_2_1", 64, 64, 64, 60, 80),
          ("upconv_1_0", 64, -1, 32, 120, 160), ("upconv_1_1", 32, 64, 32, 120, 160)]


def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): fn()
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


for name, c1, c2, co, h, w in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    up = c2 >= 0
    H, W = (2 * h, 2 * w) if up else (h, w)
    x = torch.randn(B, c1, h, w, device="cuda", generator=g).contiguous(memory_format=CL)
    skip = torch.randn(B, c2, H, W, device="cuda", generator=g).contiguous(memory_format=CL) if c2 > 0 else None
    ci = c1 + max(c2, 0)
    x2 = skip if skip is not None else (DC.UPSAMPLE_ONLY if up else None)
    dz = torch.randn(B, co, H, W, device="cuda", generator=g).contiguous(memory_format=CL)
    shape = (co, ci, 3, 3)
    ref = None
    if B <= 2:
        w64 = torch.zeros(shape, device="cuda", dtype=torch.float64, requires_grad=True)
        xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up else x.double()
        if skip is not None:
            xin = torch.cat([xin, skip.double()], 1)
        F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w64).backward(dz.double())
        ref = w64.grad
    sink = torch.zeros(shape, device="cuda").contiguous(memory_format=CL)
    dw_w, _ = DC.conv3x3_wino_wgrad_gen(x, x2, dz, shape)
    dw_d, _ = DC.conv2d_wgrad(x, dz, shape, 1, 1, True, False, x2=x2)
    t_w = timeit(lambda: DC.conv3x3_wino_wgrad_gen(x, x2, dz, shape, dw_out=sink))
    t_d = timeit(lambda: DC.conv2d_wgrad(x, dz, shape, 1, 1, True, False, x2=x2, dw_out=sink))
    fl = 2.0 * B * H * W * co * ci * 9
    if co == 32:      # the thin layers' real form: ELU derivative and bias gradient inside either kernel
        yo = torch.randn(B, co, H, W, device="cuda", generator=g).contiguous(memory_format=CL)
        bs = torch.zeros(co, device="cuda")
        t_w = timeit(lambda: DC.conv3x3_wino_wgrad_gen(x, x2, dz, shape, dw_out=sink, y_out=yo, act="elu", db_out=bs))
        t_d = timeit(lambda: DC.conv2d_wgrad(x, dz, shape, 1, 1, True, True, yo, "elu", x2=x2, dw_out=sink, db_out=bs))
    print(json.dumps(dict(layer=name, B=B, wino_us=round(t_w * 1e6, 1), direct_us=round(t_d * 1e6, 1), wino_tf=round(fl / t_w / 1e12, 1),
                          direct_tf=round(fl / t_d / 1e12, 1), wino_vs_direct=rel(dw_w, dw_d.double()),
                          err_wino=None if ref is None else rel(dw_w, ref), err_direct=None if ref is None else rel(dw_d, ref))), flush=True)

# MODE 0 (zero padding) on two BasicBlock shapes: must not have moved
for name, c, h, w in [("l1_3x3", 64, 120, 160), ("l3_3x3", 256, 30, 40)]:
    x = torch.randn(B, c, h, w, device="cuda").contiguous(memory_format=CL)
    dy = torch.randn(B, c, h, w, device="cuda").contiguous(memory_format=CL)
    sink = torch.zeros(c, c, 3, 3, device="cuda").contiguous(memory_format=CL)
    t = timeit(lambda: DC.conv3x3_wino_wgrad(x, dy, (c, c, 3, 3), dw_out=sink))
    print(json.dumps(dict(basicblock=name, B=B, wino_us=round(t * 1e6, 1), wino_tf=round(2.0 * B * h * w * c * c * 9 / t / 1e12, 1))), flush=True)
