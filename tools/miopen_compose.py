"""A/B baseline OUTSIDE the product: the network primitives composed from PyTorch-ROCm's library ops (MIOpen convolution,
eager pad / upsample / cat / activation, F.batch_norm), monkey-patched over deep_visual_slam_amd.nn_ops for a timing run.

    python tools/miopen_compose.py [batch]        # ms per training step with the library composition

Round 1 shipped this composition inside nn_ops behind DVS_CONV_BACKEND=miopen; a shape outside the hand-written kernels'
coverage then silently left the MI355X-native path.  The package now raises DvsError instead, and the composition lives here.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_visual_slam_amd import nn_ops  # noqa: E402

_ACT = {None: lambda v: v, "relu": F.relu, "elu": F.elu, "sigmoid": torch.sigmoid}


def conv2d(x, weight, bias=None, stride=1, padding=0, reflect_pad=0, act=None, x2=None, upsample=False, planar_norm=None):
    if planar_norm is not None:
        sc, sh = planar_norm
        x = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if upsample or x2 is not None:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if x2 is not None:
        x = torch.cat([x, x2], 1)
    if reflect_pad:
        x = F.pad(x, (reflect_pad,) * 4, mode="reflect")
    return _ACT[act](F.conv2d(x, weight, bias, stride, padding))


def batch_norm(x, bn, relu=False, residual=None):
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def conv_bn_act(x, weight, bn, stride=1, padding=0, relu=True, residual=None, res=None, planar_norm=None):
    y = conv2d(x, weight, None, stride, padding, planar_norm=planar_norm)
    if res is not None:
        residual = batch_norm(conv2d(residual, res[0], None, res[2], 0), res[1])
    return batch_norm(y, bn, relu=relu, residual=residual)


def patch():
    nn_ops.conv2d, nn_ops.conv_bn_act = conv2d, conv_bn_act
    nn_ops.max_pool_3x3_s2 = lambda x: F.max_pool2d(x, 3, 2, 1)


if __name__ == "__main__":
    patch()
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dn, pn = DepthNet(18, pretrained=False).to(dev).train(), PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=480, img_w=640, smoothness_ratio=0.001, auto_mask=True, ssim_ratio=0.85,
                         min_depth=0.1, max_depth=10.0, use_compile=False, pose_pairs_batched=False, arena=False)}
    tr = MonodepthTrainer(dn, pn, cfg, dev)
    opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), lr=1e-4)
    sample = synth.throughput_sample(B, 480, 640, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        _, losses = tr.process_batch(sample)
        losses["loss"].backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print("library composition (MIOpen convs + eager glue, fused loss chain): %.2f ms/step at batch %d" % ((time.perf_counter() - t0) / 10 * 1e3, B))
