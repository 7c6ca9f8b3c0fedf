"""Run-to-run repeatability of one training step's gradient arena (same weights, same data, same noise)."""
import sys, os
sys.path.insert(0, ".")
import torch
from deep_visual_slam_amd import dp, gradsink, synth
from deep_visual_slam_amd.depthnet import DepthNet
from deep_visual_slam_amd.learner_new import MonodepthTrainer
from deep_visual_slam_amd.posenet_single import PoseNet
dev = torch.device("cuda:0")
B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 96, 128
cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                     ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
torch.manual_seed(5)
dn = DepthNet(18, pretrained=False).to(dev).train()
pn = PoseNet(18, pretrained=False, num_input_images=2).to(dev).train()
flat = dp.FlatParams(dp.trainable_parameters(dn, pn))
tr = MonodepthTrainer(dn, pn, cfg, dev)
sample = {k: v.to(dev) for k, v in synth.parity_sample(B, H, W, seed=10).items()}
g = torch.Generator().manual_seed(3)
noise = torch.stack([torch.randn(B, 2, H, W, generator=g) for _ in range(4)]).to(dev)
ref = None
for rep in range(6):
    tr._noise = noise
    _, losses = tr.process_batch(dict(sample))
    losses["loss"].backward()
    gradsink.join()
    torch.cuda.synchronize()
    gcur = flat.grads.clone()
    flat.zero_grad()
    torch.cuda.synchronize()
    if ref is None:
        ref = gcur
    else:
        worst = max((float((gcur[o:o + p.numel()] - ref[o:o + p.numel()]).norm() / (ref[o:o + p.numel()].norm() + 1e-30)), n)
                    for n, p, o in zip(flat.names, flat.tensors, flat.offsets))
        print("rep %d loss %.7f  arena rel diff %.3e  worst tensor %.3e %s" % (rep, float(losses["loss"]),
              float((gcur - ref).norm() / ref.norm()), worst[0], worst[1]), flush=True)
