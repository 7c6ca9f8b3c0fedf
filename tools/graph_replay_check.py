"""Replays of several captured inference graphs against eager results (found the hipMemsetAsync-node ordering problem\nthat the split-K forward now avoids with its own zero-fill kernel)."""
import sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from deep_visual_slam_amd import inference
from deep_visual_slam_amd.depthnet import DepthNet
from deep_visual_slam_amd.posenet_single import PoseNet
from test_inference_gpu import rel
dev = torch.device("cuda:0")
torch.manual_seed(0)
dn = DepthNet(18, pretrained=False).to(dev).eval()
pn = PoseNet(18, pretrained=False, num_input_images=2).to(dev).eval()
dn.inference_scales = (0,)
xg = torch.rand(1, 3, 96, 128, device=dev)
x6 = torch.rand(1, 6, 96, 128, device=dev)
with torch.no_grad():
    d0 = dn(xg)[("disp", 0)].clone()
    aa0 = pn(x6)[0].clone()
gd = inference.Graphed(dn, torch.zeros_like(xg))
print("gd fresh", rel(gd(xg)[("disp", 0)], d0))
# V2: a plain torch graph of an unrelated op captured afterwards
g = torch.cuda.CUDAGraph()
a = torch.zeros(1024, device=dev)
with torch.cuda.graph(g):
    b = a * 2 + 1
print("gd after unrelated capture", rel(gd(xg)[("disp", 0)], d0))
gd2 = inference.Graphed(dn, torch.zeros_like(xg))
print("gd after 2nd DepthNet capture", rel(gd(xg)[("disp", 0)], d0), "gd2", rel(gd2(xg)[("disp", 0)], d0))
gp = inference.Graphed(pn, torch.zeros_like(x6))
print("after PoseNet capture: gd", rel(gd(xg)[("disp", 0)], d0), "gd2", rel(gd2(xg)[("disp", 0)], d0), "gp", rel(gp(x6)[0], aa0))
gd.refresh()
print("gd refreshed", rel(gd(xg)[("disp", 0)], d0), "gp after that", rel(gp(x6)[0], aa0))
