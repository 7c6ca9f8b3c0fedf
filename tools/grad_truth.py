"""Which fp32 gradient is closer to the truth?  DepthNet / PoseNet weight gradients at a given size from (1) the HIP
path, (2) the CPU oracle in fp32, (3) the CPU oracle in fp64 (the truth), per parameter tensor:

    python tools/grad_truth.py [--net depth|pose] [--B 2] [--H 480] [--W 640] [--repeat 2]

Prints rel-L2 of gpu-vs-f64, cpu32-vs-f64, gpu-vs-cpu32 and gpu-vs-gpu (second run: float-atomics jitter) per tensor in
backward order.  If the GPU is as close to fp64 as the reference's own fp32 CPU arithmetic is, the residual is the
conditioning of the backward pass through ~20 training-mode BatchNorms, not a kernel error.  (Uses oracle/: a tool,
not the product.)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="depth")
    ap.add_argument("--B", type=int, default=2)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--pairs", type=int, default=1, help="pose only: 2 = both frame pairs in one pass of 2B")
    ap.add_argument("--eval-bn", action="store_true", help="BatchNorm in eval mode (running statistics): no batch coupling")
    args = ap.parse_args()
    import __graft_entry__ as g
    g.build()
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.posenet_single import PoseNet
    from oracle import networks as ON
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    torch.manual_seed(0)
    net = DepthNet(18, pretrained=False) if args.net == "depth" else PoseNet(18, pretrained=False, num_input_images=2)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    train = not args.eval_bn
    net = net.to(dev)
    net.train(train)
    torch.manual_seed(3)
    cin = 3 if args.net == "depth" else 6
    nb = args.B * (args.pairs if args.net == "pose" else 1)
    x = torch.rand(nb, cin, args.H, args.W)

    def oracle(dtype):
        s = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and ".fc." not in k and "running" not in k)
             for k, v in sd.items()}
        xx = x.to(dtype)
        if args.net == "depth":
            out = ON.depthnet(xx, s, train=train)
            outs = [out[("disp", i)] for i in range(4)]
        elif args.pairs == 1:
            outs = list(ON.posenet(xx, s, train=train))
        else:
            a1, t1 = ON.posenet(xx[:args.B], s, train=train)
            a2, t2 = ON.posenet(xx[args.B:], s, train=train)
            outs = [torch.cat([a1, a2]), torch.cat([t1, t2])]
        return s, outs

    s64, o64 = oracle(torch.float64)
    torch.manual_seed(9)
    cots = [torch.randn(o.shape, dtype=torch.float64) / max(1.0, o[0].numel() ** 0.5) for o in o64]
    sum((o * c).sum() for o, c in zip(o64, cots)).backward()
    s32, o32 = oracle(torch.float32)
    sum((o * c.float()).sum() for o, c in zip(o32, cots)).backward()

    def gpu_run():
        net.zero_grad(set_to_none=True)
        if args.net == "depth":
            out = net(x.to(dev))
            outs = [out[("disp", i)] for i in range(4)]
        else:
            outs = list(net(x.to(dev), pairs=args.pairs)) if args.pairs > 1 else list(net(x.to(dev)))
        sum((o * c.float().to(dev)).sum() for o, c in zip(outs, cots)).backward()
        torch.cuda.synchronize()
        return outs, {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    og, g1 = gpu_run()
    # ReLU branch flips: elements of the encoder's five feature maps (stem, layer1..4 outputs) whose sign pattern (> 0)
    # differs from the fp64 run -- the discontinuities of the backward pass
    with torch.no_grad():
        pre = "encoder.encoder"
        s64n = {k: v.detach() for k, v in s64.items()}
        s32n = {k: v.detach() for k, v in s32.items()}
        xin = x if args.pairs == 1 or args.net == "depth" else x[:args.B]
        f64 = ON.resnet_encoder(xin.double(), s64n, pre, train)
        f32 = ON.resnet_encoder(xin, s32n, pre, train)
        fg = [f.detach().cpu() for f in net.encoder.features]
        for i, (a, b, c) in enumerate(zip(fg, f32, f64)):
            a = a[:c.shape[0]]
            near = int((c.abs() < 1e-6 * c.abs().max()).sum()) - int((c == 0).sum())
            print("feature %d: %9d elements, ReLU-sign flips vs f64: gpu %d, cpu32 %d; |f64| within 1e-6 of max but nonzero: %d"
                  % (i, c.numel(), int(((a > 0) != (c > 0)).sum()), int(((b > 0) != (c > 0)).sum()), near))
    # restore the buffers the first run updated (running statistics) so that the second run sees the same state
    net.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    _, g2 = gpu_run()
    for i, (a, b, c) in enumerate(zip(og, o32, o64)):
        print("out%d  gpu-f64 %.2e  cpu32-f64 %.2e  gpu-cpu32 %.2e" % (i, rel(a, c), rel(b, c), rel(a, b)))
    print("%-52s %10s %10s %10s %10s" % ("tensor (backward order)", "gpu-f64", "cpu32-f64", "gpu-cpu32", "gpu-gpu"))
    names = [n for n, _ in net.named_parameters() if n in g1][::-1]
    worst = [0.0, 0.0, 0.0, 0.0]
    for n in names:
        e = (rel(g1[n], s64[n].grad), rel(s32[n].grad, s64[n].grad), rel(g1[n], s32[n].grad), rel(g2[n], g1[n]))
        worst = [max(a, b) for a, b in zip(worst, e)]
        print("%-52s %10.2e %10.2e %10.2e %10.2e" % ((n,) + e))
    print("%-52s %10.2e %10.2e %10.2e %10.2e" % (("WORST",) + tuple(worst)))


if __name__ == "__main__":
    main()
