"""Forward / data-gradient / weight-gradient timing of the depth decoder's convolutions in their real gather modes
(reflection pad, nearest-upsample + concat input, ELU).  usage: dec_bench.py [B] [layer-prefix]"""
import sys, json, time
import torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
dev = torch.device("cuda:0")
CL = torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
only = sys.argv[2] if len(sys.argv) > 2 else ""
# name, C1, C2 (skip), Cout, H, W (output resolution), upsample
LAYERS = [("up4_0", 512, 0, 256, 15, 20, False), ("up4_1", 256, 256, 256, 30, 40, True),
          ("up3_0", 256, 0, 128, 30, 40, False), ("up3_1", 128, 128, 128, 60, 80, True),
          ("up2_0", 128, 0, 64, 60, 80, False), ("up2_1", 64, 64, 64, 120, 160, True),
          ("up1_0", 64, 0, 32, 120, 160, False), ("up1_1", 32, 64, 32, 240, 320, True),
          ("up0_0", 32, 0, 16, 240, 320, False), ("up0_1", 16, 0, 16, 480, 640, True)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
torch.manual_seed(0)
tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0)
for name, c1, c2, co, H, W, up in LAYERS:
    if not name.startswith(only): continue
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, c1, hs, ws, device=dev).contiguous(memory_format=CL)
    x2 = torch.randn(B, c2, H, W, device=dev).contiguous(memory_format=CL) if c2 else (DC.UPSAMPLE_ONLY if up else None)
    w = (torch.randn(co, c1 + c2, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
    bias = torch.zeros(co, device=dev)
    y = DC.conv2d_forward(x, w, bias, 1, 1, True, "elu", x2=x2)
    dy = torch.randn_like(y)
    dw = torch.zeros_like(w)
    db = torch.zeros(co, device=dev)
    fl = 2.0 * B * H * W * co * (c1 + c2) * 9
    t_f = timeit(lambda: DC.conv2d_forward(x, w, bias, 1, 1, True, "elu", x2=x2))
    t_d = timeit(lambda: DC.conv2d_dgrad(dy, w, (B, c1 + c2, H, W), 1, 1, True, y, "elu", split_c1=c1 if up else 0))
    t_w = timeit(lambda: DC.conv2d_wgrad(x, dy, tuple(w.shape), 1, 1, True, True, y, "elu", x2=x2, dw_out=dw, db_out=db))
    tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
    print(json.dumps(dict(name=name, gflop=round(fl / 1e9, 1), fwd_ms=round(t_f * 1e3, 3), fwd_tf=round(fl / t_f / 1e12, 1),
                          dgrad_ms=round(t_d * 1e3, 3), dgrad_tf=round(fl / t_d / 1e12, 1),
                          wgrad_ms=round(t_w * 1e3, 3), wgrad_tf=round(fl / t_w / 1e12, 1))), flush=True)
print(json.dumps({k: round(v * 1e3, 3) for k, v in tot.items()}))
