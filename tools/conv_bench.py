"""Correctness (vs F.conv2d in fp64-on-GPU... fp32 MIOpen) and timing of the hand-written conv kernels
on every convolution shape of the path.  usage: conv_bench.py [B] [fwd|all]"""
import sys, json, time
import torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
CL = torch.channels_last
# name, Cin, Cout, k, stride, pad, reflect, H, W, act, bias
shapes = [
    ("l1_3x3", 64, 64, 3, 1, 1, 0, 120, 160, None, 0), ("l2_s2", 64, 128, 3, 2, 1, 0, 120, 160, None, 0),
    ("l2_ds", 64, 128, 1, 2, 0, 0, 120, 160, None, 0), ("l2_3x3", 128, 128, 3, 1, 1, 0, 60, 80, None, 0),
    ("l3_s2", 128, 256, 3, 2, 1, 0, 60, 80, None, 0), ("l3_3x3", 256, 256, 3, 1, 1, 0, 30, 40, None, 0),
    ("l4_s2", 256, 512, 3, 2, 1, 0, 30, 40, None, 0), ("l4_3x3", 512, 512, 3, 1, 1, 0, 15, 20, None, 0),
    ("up4_0", 512, 256, 3, 1, 1, 1, 15, 20, "elu", 1), ("up3_0", 256, 128, 3, 1, 1, 1, 30, 40, "elu", 1),
    ("up2_0", 128, 64, 3, 1, 1, 1, 60, 80, "elu", 1), ("up1_0", 64, 32, 3, 1, 1, 1, 120, 160, "elu", 1),
    ("up0_0", 32, 16, 3, 1, 1, 1, 240, 320, "elu", 1), ("up0_1", 16, 16, 3, 1, 1, 1, 480, 640, "elu", 1),
    ("disp0", 16, 1, 3, 1, 1, 1, 480, 640, "sigmoid", 1), ("disp3", 128, 1, 3, 1, 1, 1, 60, 80, "sigmoid", 1),
    ("pose_sq", 512, 256, 1, 1, 0, 0, 15, 20, "relu", 1), ("pose_0", 256, 256, 3, 1, 1, 0, 15, 20, "relu", 1),
    ("pose_2", 256, 6, 1, 1, 0, 0, 15, 20, None, 1),
]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
def ref_act(y, act):
    return {None: lambda v: v, "elu": F.elu, "relu": F.relu, "sigmoid": torch.sigmoid}[act](y)
torch.manual_seed(0)
rows = []
for (name, ci, co, k, s, p, refl, h, w, act, has_b) in shapes:
    x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL)
    wt = (torch.randn(co, ci, k, k, device=dev) * (2.0 / (ci * k * k)) ** 0.5).contiguous(memory_format=CL)
    bias = torch.randn(co, device=dev) * 0.1 if has_b else None
    def ref():
        xx = F.pad(x, (p,) * 4, mode="reflect") if refl else x
        return ref_act(F.conv2d(xx, wt, bias, s, 0 if refl else p), act)
    y_ref = ref()
    y = DC.conv2d_forward(x, wt, bias, s, p, bool(refl), act)
    err = float((y - y_ref).abs().max() / (y_ref.abs().max() + 1e-30))
    ho, wo = y.shape[2:]
    fl = 2.0 * B * co * ho * wo * ci * k * k
    t_hip = timeit(lambda: DC.conv2d_forward(x, wt, bias, s, p, bool(refl), act))
    t_ref = timeit(ref)
    r = dict(name=name, relerr=err, gflop=fl / 1e9, hip_ms=t_hip * 1e3, hip_tf=fl / t_hip / 1e12, miopen_ms=t_ref * 1e3,
             miopen_tf=fl / t_ref / 1e12)
    rows.append(r); print(json.dumps(r), flush=True)
# special input paths
# conv1 from planar NCHW with input normalisation
for ci in (3, 6):
    x = torch.rand(B, ci, 480, 640, device=dev)
    wt = (torch.randn(64, ci, 7, 7, device=dev) * 0.05).contiguous(memory_format=CL)
    sc = torch.full((ci,), 1 / 0.225, device=dev); sh = torch.full((ci,), -0.45 / 0.225, device=dev)
    ref = lambda: F.conv2d((x - 0.45) / 0.225, wt, None, 2, 3)
    y_ref = ref()
    y = DC.conv2d_forward(x, wt, None, 2, 3, False, None, in_scale=sc, in_shift=sh, nchw_planar=True)
    err = float((y - y_ref).abs().max() / y_ref.abs().max())
    fl = 2.0 * B * 64 * 240 * 320 * ci * 49
    t_hip = timeit(lambda: DC.conv2d_forward(x, wt, None, 2, 3, False, None, in_scale=sc, in_shift=sh, nchw_planar=True))
    t_ref = timeit(ref)
    print(json.dumps(dict(name="conv1_%dch" % ci, relerr=err, gflop=fl / 1e9, hip_ms=t_hip * 1e3, hip_tf=fl / t_hip / 1e12,
                          miopen_ms=t_ref * 1e3, miopen_tf=fl / t_ref / 1e12)), flush=True)
# upsample + concat fused gather (decoder upconv_i_1), BN-fold on load, stats epilogue
for (name, c1, c2, co, h, w) in (("up4_1", 256, 256, 256, 30, 40), ("up1_1", 32, 64, 32, 240, 320)):
    xa = torch.randn(B, c1, h // 2, w // 2, device=dev).contiguous(memory_format=CL)
    xb = torch.randn(B, c2, h, w, device=dev).contiguous(memory_format=CL)
    wt = (torch.randn(co, c1 + c2, 3, 3, device=dev) * 0.03).contiguous(memory_format=CL)
    bias = torch.randn(co, device=dev) * 0.1
    def ref():
        xx = torch.cat([F.interpolate(xa, scale_factor=2, mode="nearest"), xb], 1)
        return F.elu(F.conv2d(F.pad(xx, (1,) * 4, mode="reflect"), wt, bias))
    y_ref = ref()
    y = DC.conv2d_forward(xa, wt, bias, 1, 1, True, "elu", x2=xb)
    err = float((y - y_ref).abs().max() / y_ref.abs().max())
    fl = 2.0 * B * co * h * w * (c1 + c2) * 9
    t_hip = timeit(lambda: DC.conv2d_forward(xa, wt, bias, 1, 1, True, "elu", x2=xb)); t_ref = timeit(ref)
    print(json.dumps(dict(name=name + "_upcat", relerr=err, gflop=fl / 1e9, hip_ms=t_hip * 1e3, hip_tf=fl / t_hip / 1e12,
                          miopen_ms=t_ref * 1e3, miopen_tf=fl / t_ref / 1e12)), flush=True)
x = torch.randn(B, 64, 60, 80, device=dev).contiguous(memory_format=CL)
wt = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.2
stats = torch.zeros(2, 64, device=dev)
y = DC.conv2d_forward(x, wt, None, 1, 1, False, None, in_scale=sc, in_shift=sh, in_relu=True, stats=stats)
y_ref = F.conv2d(F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), wt, None, 1, 1)
print(json.dumps(dict(name="bnfold+stats", relerr=float((y - y_ref).abs().max() / y_ref.abs().max()),
                      stats_err=float((stats[0] - y_ref.sum((0, 2, 3))).abs().max() / y_ref.sum((0, 2, 3)).abs().max()),
                      sq_err=float((stats[1] - (y_ref ** 2).sum((0, 2, 3))).abs().max() / (y_ref ** 2).sum((0, 2, 3)).abs().max()))))
