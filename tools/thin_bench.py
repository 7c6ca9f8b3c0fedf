"""Weight-gradient timing of the decoder's thin layers in their real gather modes (DVS_CONV_THIN=0 for the generic kernel).
Wall-clock over back-to-back Python calls: below ~120 us per call the figure is the HOST loop, not the kernel -- use tools/per_launch.py
(events around the launch inside a real step) for those, as DESIGN.md section 11-2b does."""
import sys, json, time
import torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
dev = torch.device("cuda:0")
CL = torch.channels_last
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
# name, C1, C2 (skip), Cout, H, W (output resolution), upsample
LAYERS = [("up1_0", 64, 0, 32, 120, 160, False), ("up1_1", 32, 64, 32, 240, 320, True),
          ("up0_0", 32, 0, 16, 240, 320, False), ("up0_1", 16, 0, 16, 480, 640, True)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
torch.manual_seed(0)
for name, c1, c2, co, H, W, up in LAYERS:
    hs, ws = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(B, c1, hs, ws, device=dev).contiguous(memory_format=CL)
    x2 = torch.randn(B, c2, H, W, device=dev).contiguous(memory_format=CL) if c2 else (DC.UPSAMPLE_ONLY if up else None)
    y = torch.randn(B, co, H, W, device=dev).contiguous(memory_format=CL)
    dy = torch.randn_like(y)
    wshape = (co, c1 + c2, 3, 3)
    dw = torch.zeros(wshape, device=dev).contiguous(memory_format=CL)
    db = torch.zeros(co, device=dev)
    fl = 2.0 * B * H * W * co * (c1 + c2) * 9
    t = timeit(lambda: DC.conv2d_wgrad(x, dy, wshape, 1, 1, True, True, y, "elu", x2=x2, dw_out=dw, db_out=db))
    print(json.dumps(dict(name=name, gflop=round(fl / 1e9, 1), wgrad_ms=round(t * 1e3, 3), tf=round(fl / t / 1e12, 1))), flush=True)
