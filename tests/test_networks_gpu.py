"""GPU parity of DepthNet / PoseNet / MonodepthTrainer against the CPU oracle restatement
(same seeded weights through state_dict, training-mode BatchNorm)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def nets(gpu_device):
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.posenet_single import PoseNet
    torch.manual_seed(0)
    dn, pn = DepthNet(18, pretrained=False), PoseNet(18, pretrained=False, num_input_images=2)
    sd_d = {k: v.clone() for k, v in dn.state_dict().items()}
    sd_p = {k: v.clone() for k, v in pn.state_dict().items()}
    return dn.to(gpu_device).train(), pn.to(gpu_device).train(), sd_d, sd_p


def test_depthnet_forward_and_running_stats(gpu_device, nets):
    from oracle import networks as ON
    dn, _, sd_d, _ = nets
    torch.manual_seed(3)
    x = torch.rand(2, 3, 96, 128)
    upd = {}
    ref = ON.depthnet(x, sd_d, train=True, update=upd)
    out = dn(x.to(gpu_device))
    for s in range(4):
        assert out[("disp", s)].shape == ref[("disp", s)].shape
        assert rel(out[("disp", s)], ref[("disp", s)]) < 2e-4     # fp32, ~40 conv layers deep
    sd_new = dn.state_dict()
    for k in ("encoder.encoder.bn1.running_mean", "encoder.encoder.layer3.0.bn2.running_var",
              "encoder.encoder.layer2.0.downsample.1.running_mean"):
        assert rel(sd_new[k], upd[k]) < 1e-4
    assert int(sd_new["encoder.encoder.bn1.num_batches_tracked"]) == 1


def test_posenet_forward_backward(gpu_device, nets):
    from oracle import networks as ON
    _, pn, _, sd_p = nets
    torch.manual_seed(4)
    x = torch.rand(2, 6, 96, 128)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd_p.items()}
    aa_r, t_r = ON.posenet(x, sd, train=True)
    aa, t = pn(x.to(gpu_device))
    assert aa.shape == (2, 1, 1, 3) and t.shape == (2, 1, 1, 3)
    assert rel(aa, aa_r) < 2e-4 and rel(t, t_r) < 2e-4
    cot = torch.randn(2, 1, 1, 3)
    ((aa_r + t_r) * cot).sum().backward()
    pn.zero_grad()
    ((aa + t) * cot.to(gpu_device)).sum().backward()
    for k in ("net.3.weight", "net.1.weight", "encoder.encoder.conv1.weight", "encoder.encoder.layer4.1.conv2.weight"):
        g = dict(pn.named_parameters())[k].grad
        assert rel(g, sd[k].grad) < 2e-3, k


def test_trainer_step_matches_oracle(gpu_device, nets):
    """Whole process_batch (nets + fused chain) vs oracle nets + oracle chain with injected noise."""
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from oracle import loss_chain as OL, networks as ON
    dn, pn, _, _ = nets
    B, H, W = 2, 96, 128
    sd_d = {k: v.detach().cpu().clone() for k, v in dn.state_dict().items()}
    sd_p = {k: v.detach().cpu().clone() for k, v in pn.state_dict().items()}
    sample = synth.parity_sample(B, H, W)
    g = torch.Generator().manual_seed(7)
    noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(4)]
    # oracle
    tgt, left, right = sample[("target_image", 0)], sample[("source_left", 0)], sample[("source_right", 0)]
    disp = ON.depthnet(tgt, sd_d, train=True)
    aa_l, t_l = ON.posenet(torch.cat([left, tgt], 1), sd_p, train=True)
    aa_r, t_r = ON.posenet(torch.cat([tgt, right], 1), sd_p, train=True)
    _, ref_losses = OL.loss_chain(sample, [disp[("disp", s)] for s in range(4)], (aa_l, t_l, aa_r, t_r), noise)
    # product
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                         ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
    tr = MonodepthTrainer(dn, pn, cfg, gpu_device)
    tr._noise = torch.stack(noise).to(gpu_device)
    outputs, losses = tr.process_batch(dict(sample))
    for k in ("loss", "loss/0", "loss/1", "loss/2", "loss/3"):
        assert abs(float(losses[k]) - float(ref_losses[k])) < 2e-4 * abs(float(ref_losses[k])), k
    losses["loss"].backward()
    assert all(p.grad is not None for n, p in dn.named_parameters() if ".fc." not in n)
    assert all(p.grad is None for n, p in dn.named_parameters() if ".fc." in n)
    # lazily materialised outputs keep the reference's schema
    assert outputs[("color", -1, 2)].shape == (B, 3, H, W)
    assert outputs[("sample", 1, 0)].shape == (B, H, W, 2)
    assert outputs[("depth", 3)].shape == (B, 1, H, W)
    assert outputs["identity_selection/1"].shape == (B, 1, H, W)
    assert outputs[("cam_T_cam", 0, -1)].shape == (B, 4, 4)
    with pytest.raises(KeyError):
        outputs[("nope", 0)]


def test_gradient_sinks_match_autograd_accumulation(gpu_device):
    """FlatParams(grad_sinks=True): the kernels add parameter gradients straight into the arena.  The same step
    on identical weights with sinks off (autograd's AccumulateGrad) must give the same arena up to the
    run-to-run noise of the atomics (measured between two sink-less runs), a second backward must accumulate
    (not overwrite), and every parameter's post-accumulate hook fires exactly once per backward."""
    from deep_visual_slam_amd import _lib, dp, gradsink, synth
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    B, H, W = 2, 96, 128
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                         ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
    sample = {k: v.to(gpu_device) for k, v in synth.parity_sample(B, H, W).items()}
    g = torch.Generator().manual_seed(11)
    noise = torch.stack([torch.randn(B, 2, H, W, generator=g) for _ in range(4)]).to(gpu_device)
    arenas, fired = [], None
    # Deterministic forward (dvs_set_deterministic): every run takes the same ReLU / maxpool branches, so what is left
    # between two runs is the smooth rounding noise of the backward's float atomics -- the comparison below can then be
    # tight enough to see a lost or doubled contribution of ANY size (round 1 had to allow 4 x jitter + 5e-3).
    _lib.set_deterministic(True)
    try:
        _sink_runs(gpu_device, cfg, sample, noise, arenas, B, H, W)
    finally:
        _lib.set_deterministic(False)
    fired = arenas.pop()
    (a1, a2, flat), (b1, b2, _), (s1, s2, _) = arenas
    worst = (0.0, 0.0, "")
    for ref, other, got in ((a1, b1, s1), (a2, b2, s2)):
        for n, p, o in zip(flat.names, flat.tensors, flat.offsets):
            r, q, t = (x[o:o + p.numel()].double() for x in (ref, other, got))
            scale = float(r.norm()) + 1e-30
            jitter, err = float((r - q).norm()) / scale, float((r - t).norm()) / scale
            worst = max(worst, (err, jitter, n))
            assert jitter <= 2e-5 and err <= 2e-5, (n, err, jitter, float(r.norm()), float(q.norm()), float(t.norm()), float((q - t).norm()))   # measured 1.8e-6
    print("worst sink-vs-autograd rel-L2 %.2e (run-to-run there %.2e) at %s" % worst)
    assert float((s2 - 2 * s1).norm()) <= 2e-5 * float(s1.norm())
    assert set(fired) == set(flat.names), sorted(set(flat.names) - set(fired))
    assert all(v == 2 for v in fired.values()), {k: v for k, v in fired.items() if v != 2}


def _sink_runs(gpu_device, cfg, sample, noise, arenas, B, H, W):
    from deep_visual_slam_amd import dp, gradsink
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    fired = None
    for sinks in (False, False, True):
        torch.manual_seed(5)
        dn = DepthNet(18, pretrained=False).to(gpu_device).train()
        pn = PoseNet(18, pretrained=False, num_input_images=2).to(gpu_device).train()
        flat = dp.FlatParams(dp.trainable_parameters(dn, pn), grad_sinks=sinks)
        if sinks:
            fired = {}
            for n, p in zip(flat.names, flat.tensors):
                p.register_post_accumulate_grad_hook(lambda _p, n=n: fired.__setitem__(n, fired.get(n, 0) + 1))
        tr = MonodepthTrainer(dn, pn, cfg, gpu_device)
        for rep in range(2):                       # two backwards without zeroing: gradients must add up
            tr._noise = noise
            _, losses = tr.process_batch(dict(sample))
            losses["loss"].backward()
            gradsink.join()                        # sunk weight gradients run on side streams
            if rep == 0:
                first = flat.grads.clone()
        arenas.append((first, flat.grads.clone(), flat))
    arenas.append(fired)


def test_posenet_pairs_in_one_pass_equal_two_calls(gpu_device):
    """PoseNet.forward(x, pairs=2) -- both frame pairs as one batch of 2B with per-pair BatchNorm statistics -- against
    the reference's two calls (vo/learner_new.py:113-114): outputs and BatchNorm running statistics (two momentum updates
    in call order) must agree tightly; the parameter gradients of BOTH forms are judged against an fp64 run of the oracle
    with the fp32 oracle as the yardstick.  (Round 1 compared the two GPU forms with each other at 5e-2.  The cause of the
    spread is not summation order as such: rounding differences flip single ReLU branches, one flip moves a gradient
    tensor by ~1/sqrt(elements of that activation) -- 7e-3 at layer 4 of this 96x128 case -- and the 24-sample BatchNorms
    spread it over the channel; the reference's own fp32 CPU arithmetic shows the same steps against fp64,
    tools/grad_truth.py / profiles/r02_grad_truth_pose_pairs_96.txt.)"""
    from deep_visual_slam_amd.posenet_single import PoseNet
    from oracle import networks as ON
    torch.manual_seed(21)
    a = PoseNet(18, pretrained=False, num_input_images=2)
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    a = a.to(gpu_device).train()
    b = PoseNet(18, pretrained=False, num_input_images=2).to(gpu_device).train()
    b.load_state_dict(a.state_dict())
    B, H, W = 3, 96, 128
    x1c, x2c = torch.rand(B, 6, H, W), torch.rand(B, 6, H, W)
    wc = torch.randn(2 * B, 1, 1, 6)
    x1, x2, w = x1c.to(gpu_device), x2c.to(gpu_device), wc.to(gpu_device)
    aa1, t1 = a(x1)
    aa2, t2 = a(x2)
    (torch.cat([torch.cat([aa1, t1], -1), torch.cat([aa2, t2], -1)]) * w).sum().backward()
    aab, tb = b(torch.cat([x1, x2]), pairs=2)
    (torch.cat([aab, tb], -1) * w).sum().backward()
    assert rel(aab, torch.cat([aa1, aa2])) < 1e-5 and rel(tb, torch.cat([t1, t2])) < 1e-5
    for (n, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        if "num_batches_tracked" in n:
            assert int(ba) == int(bb) == 2, n
        elif ".fc." not in n:
            assert rel(bb, ba) < 1e-5, n
    # the truth and the yardstick
    grads = {}
    for dtype in (torch.float64, torch.float32):
        s = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point() and ".fc." not in k and "running" not in k)
             for k, v in sd.items()}
        o1, o2 = ON.posenet(x1c.to(dtype), s, train=True), ON.posenet(x2c.to(dtype), s, train=True)
        (torch.cat([torch.cat(o1, -1), torch.cat(o2, -1)]) * wc.to(dtype)).sum().backward()
        grads[dtype] = {k: v.grad for k, v in s.items() if v.requires_grad}
    worst_cpu = max(rel(grads[torch.float32][k], grads[torch.float64][k]) for k in grads[torch.float64])
    flip = 1.0 / (B * 3 * 4 * 256) ** 0.5              # one branch flip in the smallest ReLU map (pose decoder, 3x4)
    for net, tag in ((a, "two calls"), (b, "pairs=2")):
        worst = max((rel(p.grad, grads[torch.float64][n]), n) for n, p in net.named_parameters() if p.grad is not None)
        print("%s: worst gradient error vs fp64 %.2e at %s (fp32 CPU oracle: %.2e)" % (tag, worst[0], worst[1], worst_cpu))
        assert worst[0] <= 3.0 * worst_cpu + 2.0 * flip, (tag, worst, worst_cpu)


@pytest.mark.parametrize("name", ["layers_level_skip", "layers_level_noskip"])
def test_decoder_level_matches_reference_goldens(gpu_device, name):
    """One decoder level through the drop-in modules (fused upsample + concat gather, ELU / sigmoid epilogues, heads on
    the vector ALUs) against vectors produced by the reference's own model/layers.py (tests/golden/make_golden_layers.py)."""
    import os
    from deep_visual_slam_amd.layers import Conv3x3, ConvBlock
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    T = lambda k: torch.from_numpy(g[k]).to(gpu_device)
    c_mid, c_in = g["blk0.conv.conv.weight"].shape[:2]
    c_out, c_cat = g["blk1.conv.conv.weight"].shape[:2]
    blk0, blk1, head = ConvBlock(c_in, c_mid), ConvBlock(c_cat, c_out), Conv3x3(c_out, 1)
    for mod, pre in ((blk0, "blk0."), (blk1, "blk1."), (head, "head.")):
        mod.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)})
        mod.to(gpu_device).to(memory_format=torch.channels_last)
    x = T("x").requires_grad_(True)
    skip = T("skip").requires_grad_(True) if "skip" in g.files else None
    feat = blk1(blk0(x), skip=skip, upsample=True)
    disp = head(feat, act="sigmoid")
    assert rel(feat, T("feat")) < 2e-5 and rel(disp, T("disp")) < 2e-5
    ins = [x] + ([skip] if skip is not None else [])
    params = [("blk0.", blk0), ("blk1.", blk1), ("head.", head)]
    plist = [(pre + k, p) for pre, m in params for k, p in m.named_parameters()]
    grads = torch.autograd.grad([feat, disp], ins + [p for _, p in plist], [T("cot_feat"), T("cot_disp")])
    assert rel(grads[0], T("d_x")) < 1e-4
    if skip is not None:
        assert rel(grads[1], T("d_skip")) < 1e-4
    for (k, _), gr in zip(plist, grads[len(ins):]):
        assert rel(gr, T("d_" + k)) < 1e-4, k
