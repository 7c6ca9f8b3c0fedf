"""Fixed cost per workgroup vs cost per k-step of the Winograd forward kernel: time over the number of input channels at a fixed
geometry (line fit), for the two workgroup shapes.  usage: wino_fixed_cost.py [B]"""
import sys, json
import torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import _lib

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CL = torch.channels_last
l = _lib.lib(); st = _lib.stream()
for name, co, h, w in (("l1 <2,2>", 64, 120, 160), ("l2 <1,4>", 128, 60, 80), ("l3 <1,4>", 256, 30, 40)):
    pts = []
    for ci in (32, 64, 128, 256):
        x = torch.randn(B, ci, h, w, device=dev).contiguous(memory_format=CL)
        u = torch.randn(ci * co * 16, device=dev)
        y = torch.empty((B, co, h, w), device=dev).contiguous(memory_format=CL)
        stt = torch.zeros(16, 1, 2, co, device=dev)
        def go():
            l.dvs_conv3x3_wino_fwd_slots(x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), stt.data_ptr(), 1, 16, B, h, w, ci, co, 0, 0, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5): go()
        e0.record()
        for _ in range(40): go()
        e1.record(); torch.cuda.synchronize()
        pts.append((ci // 2, e0.elapsed_time(e1) * 1e3 / 40))
    tiles = B * (h // 2) * (w // 2)
    mt, wc = (64, 2) if co <= 64 else (32, 4)
    wgs = -(-tiles // mt) * -(-co // (32 * wc))
    rounds = wgs / 256.0
    # least squares us = a + b * ksteps
    n = len(pts); sx = sum(k for k, _ in pts); sy = sum(t for _, t in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * t for k, t in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
    print(json.dumps(dict(shape=name, B=B, workgroups=wgs, rounds=round(rounds, 2), points=[(k, round(t, 1)) for k, t in pts],
                          fixed_us_per_launch=round(a, 1), us_per_kstep_per_launch=round(b, 3),
                          fixed_us_per_wg=round(a / max(rounds, 1), 2), us_per_kstep_per_wg=round(b / max(rounds, 1), 3))))
