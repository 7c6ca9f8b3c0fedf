import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC
from deep_visual_slam_amd.posenet_single import PoseNet
dev = torch.device("cuda:0")
CL = torch.channels_last
rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
torch.manual_seed(0)
for B in (3, 6):
    for (c, h, w) in ((64, 24, 32), (128, 12, 16), (256, 6, 8), (512, 3, 4)):
        x = torch.randn(B, c, h, w, device=dev).contiguous(memory_format=CL)
        wt = (torch.randn(c, c, 3, 3, device=dev) * 0.03)
        y64 = F.conv2d(x.double(), wt.double(), None, 1, 1)
        for g in ((1, 2) if B % 2 == 0 else (1,)):
            st = torch.zeros((2, c) if g == 1 else (2, 2, c), device=dev)
            y = DC.conv3x3_wino(x, wt, st, g)
            parts = [y64] if g == 1 else [y64[: B // 2], y64[B // 2:]]
            ref = torch.stack([torch.stack([p.sum((0, 2, 3)), (p * p).sum((0, 2, 3))]) for p in parts])
            ref = ref[0] if g == 1 else ref
            print("B", B, "C", c, h, w, "groups", g, "y", rel(y, y64), "stats", rel(st, ref), flush=True)
for wino in (True, False):
    DC._WINO = wino
    DC._wino_packed.clear()
    torch.manual_seed(21)
    a = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
    b = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
    b.load_state_dict(a.state_dict())
    x1, x2 = torch.rand(3, 6, 96, 128, device=dev), torch.rand(3, 6, 96, 128, device=dev)
    with torch.no_grad():
        aa1, t1 = a(x1); aa2, t2 = a(x2)
        aab, tb = b(torch.cat([x1, x2]), pairs=2)
    print("wino", wino, "pairs vs two calls", rel(aab, torch.cat([aa1, aa2])), rel(tb, torch.cat([t1, t2])))
    if wino:
        keep = (aa1, aa2, aab)
    else:
        print("two calls wino vs direct", rel(keep[0], aa1), rel(keep[1], aa2), "pairs wino vs direct", rel(keep[2], aab))
