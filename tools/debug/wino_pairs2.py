import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from deep_visual_slam_amd import conv as DC, zeropool
from deep_visual_slam_amd.posenet_single import PoseNet
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
if len(sys.argv) > 1 and sys.argv[1] == "pool":
    zeropool.reset(dev)
torch.manual_seed(21)
a = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
b = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
b.load_state_dict(a.state_dict())
c = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
c.load_state_dict(a.state_dict())
B, H, W = 3, 96, 128
x1, x2 = torch.rand(B, 6, H, W, device=dev), torch.rand(B, 6, H, W, device=dev)
w = torch.randn(2 * B, 1, 1, 6, device=dev)
with torch.no_grad():
    r1, _ = c(x1); r2, _ = c(x2)          # reference outputs before anything else (train-mode BN: batch statistics only)
aa1, t1 = a(x1)
aa2, t2 = a(x2)
print("a fwd vs c nograd", rel(aa1, r1), rel(aa2, r2))
(torch.cat([torch.cat([aa1, t1], -1), torch.cat([aa2, t2], -1)]) * w).sum().backward()
torch.cuda.synchronize()
aab, tb = b(torch.cat([x1, x2]), pairs=2)
print("b pairs (after a.backward) vs a", rel(aab, torch.cat([aa1, aa2])))
with torch.no_grad():
    aab2, _ = b(torch.cat([x1, x2]), pairs=2)
print("b pairs nograd again vs a", rel(aab2, torch.cat([aa1, aa2])))
DC._WINO = False
with torch.no_grad():
    aab3, _ = b(torch.cat([x1, x2]), pairs=2)
print("b pairs direct vs a", rel(aab3, torch.cat([aa1, aa2])))
