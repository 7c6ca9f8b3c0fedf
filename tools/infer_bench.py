"""Frames/s of the inference path at batch 1 (vo/predict.py's per-frame work: PoseNet on the frame pair, DepthNet on the
target frame, pose matrix, depth): eager vs HIP-graph replay (the library-convolution baseline of round 1 is gone with
the fallback path: profiles/r01_d_infer_bench.txt keeps its numbers).
usage: infer_bench.py [frames] [--cpu]   (--cpu: also time the CPU oracle's eval-mode networks on a few frames)"""
import json, os, sys, time
import torch
sys.path.insert(0, ".")
from deep_visual_slam_amd import inference, nn_ops
from deep_visual_slam_amd.depthnet import DepthNet
from deep_visual_slam_amd.posenet_single import PoseNet
from deep_visual_slam_amd.layers import transformation_from_parameters, disp_to_depth

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
dev = torch.device("cuda:0")
torch.manual_seed(0)
dn, pn = DepthNet(18, pretrained=False).to(dev), PoseNet(18, pretrained=False, num_input_images=2).to(dev)
inference.prepare(dn, pn, scales=(0,))
tgt = torch.rand(1, 3, 480, 640, device=dev)
pair = torch.rand(1, 6, 480, 640, device=dev)


def frame(depth, pose):
    aa, t = pose(pair)
    T = transformation_from_parameters(aa[:, 0], t[:, 0], invert=False)
    disp = depth(tgt)[("disp", 0)]
    _, d = disp_to_depth(disp, 0.1, 10.0)
    return T, d


def timeit(depth, pose, n):
    with torch.no_grad():
        for _ in range(10):
            frame(depth, pose)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            frame(depth, pose)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {"backend": "hip", "frames": n, "image": "640x480", "batch": 1}
t = timeit(dn, pn, n)
res["eager_ms"] = round(t * 1e3, 3); res["eager_fps"] = round(1 / t, 1)
if True:
    gd, gp = inference.Graphed(dn, tgt), inference.Graphed(pn, pair)
    t = timeit(gd, gp, n)
    res["graph_ms"] = round(t * 1e3, 3); res["graph_fps"] = round(1 / t, 1)
if "--cpu" in sys.argv:
    from oracle import networks as ON
    sd_d = {k: v.detach().cpu() for k, v in dn.state_dict().items()}
    sd_p = {k: v.detach().cpu() for k, v in pn.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count()))      # the GPU box's per-GPU CPU share
    with torch.no_grad():
        ON.depthnet(tgt.cpu(), sd_d, train=False); ON.posenet(pair.cpu(), sd_p, train=False)
        t0 = time.perf_counter()
        for _ in range(3):
            ON.depthnet(tgt.cpu(), sd_d, train=False); ON.posenet(pair.cpu(), sd_p, train=False)
        t = (time.perf_counter() - t0) / 3
    res["cpu_oracle_ms"] = round(t * 1e3, 1); res["cpu_oracle_fps"] = round(1 / t, 2); res["cpu_threads"] = min(16, os.cpu_count())
print(json.dumps(res))
