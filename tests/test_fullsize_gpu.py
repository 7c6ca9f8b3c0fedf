"""BASELINE-size (480x640) parity of the HIP path -- the shapes bench.py actually runs.

At full size the library picks other code than at the 96x128 of the small tests: 128x128 LDS-DMA tiles, the
row-ring kernels with 64/128-pixel segments on the 240x320 / 480x640 maps, split-K thresholds, the persistent
stem loops over 76 800 pixels per image.  Three layers of checks (VERDICT r1 "next round" item 1):

  (a) every Appendix-C convolution shape with its real fusion mode -- forward, data gradient, weight
      and bias gradient -- against torch's fp32 composition of the reference modules on the same device;
  (b) DepthNet / PoseNet forward and ALL weight gradients against oracle/networks.py on the CPU;
  (c) whole training steps -- BASELINE.json configs[1] (batch 4, single-scale loss) and configs[2] (batch 12,
      4 scales) -- `process_batch` + backward against oracle networks + oracle loss chain with the tie-break noise
      injected: the five loss scalars (rel 2e-4) and every parameter gradient (rel-L2 per tensor).

Every comparison also lands in gpurun_out/fullsize_report.txt (worst tensors per test) so the tolerances below
are backed by numbers, not by hope.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last
H, W = 480, 640
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def report(line):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "fullsize_report.txt"), "a") as f:
        f.write(line + "\n")
    print(line)


def relmax(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ref_act(y, act):
    return {None: lambda v: v, "elu": F.elu, "relu": F.relu, "sigmoid": torch.sigmoid}[act](y)


# ------------------------------------------------------------------------------------------------ (a)
# name, Cin, Cout, k, stride, pad, reflect, Hin, Win, act, bias, stats (BatchNorm statistics epilogue)
ENC = [
    ("l1_64_64", 64, 64, 3, 1, 1, False, 120, 160, None, False, True),
    ("l2_64_128_s2", 64, 128, 3, 2, 1, False, 120, 160, None, False, True),
    ("l2_ds_1x1_s2", 64, 128, 1, 2, 0, False, 120, 160, None, False, True),
    ("l2_128_128", 128, 128, 3, 1, 1, False, 60, 80, None, False, True),
    ("l3_128_256_s2", 128, 256, 3, 2, 1, False, 60, 80, None, False, True),
    ("l3_ds_1x1_s2", 128, 256, 1, 2, 0, False, 60, 80, None, False, True),
    ("l3_256_256", 256, 256, 3, 1, 1, False, 30, 40, None, False, True),
    ("l4_256_512_s2", 256, 512, 3, 2, 1, False, 30, 40, None, False, True),
    ("l4_ds_1x1_s2", 256, 512, 1, 2, 0, False, 30, 40, None, False, True),
    ("l4_512_512", 512, 512, 3, 1, 1, False, 15, 20, None, False, True),
    # PoseNet decoder @15x20 (model/posenet_single.py:157-172)
    ("pose_squeeze", 512, 256, 1, 1, 0, False, 15, 20, "relu", True, False),
    ("pose_3x3", 256, 256, 3, 1, 1, False, 15, 20, "relu", True, False),
]
DEC_PLAIN = [
    # upconv(i, 0): reflection pad + ELU (model/depthnet.py:43-47)
    ("up4_0", 512, 256, 15, 20), ("up3_0", 256, 128, 30, 40), ("up2_0", 128, 64, 60, 80),
    ("up1_0", 64, 32, 120, 160), ("up0_0", 32, 16, 240, 320),
]
DEC_CAT = [
    # upconv(i, 1): upsample(x) ; cat skip (model/depthnet.py:79-85): (C1 coarse, C2 skip, Cout, Hout, Wout)
    ("up4_1", 256, 256, 256, 30, 40), ("up3_1", 128, 128, 128, 60, 80), ("up2_1", 64, 64, 64, 120, 160),
    ("up1_1", 32, 64, 32, 240, 320),
]
HEADS = [("disp0", 16, 480, 640), ("disp1", 32, 240, 320), ("disp2", 64, 120, 160), ("disp3", 128, 60, 80)]


def _check_grads(tag, got, want, names, tol=1e-4):
    for a, r, nm in zip(got, want, names):
        assert a.shape == r.shape, (tag, nm)
        e = relmax(a, r)
        report("conv %-22s %-4s relmax %.2e" % (tag, nm, e))
        assert e < tol, (tag, nm, e)


@pytest.mark.parametrize("B", [2, 12, 24])
@pytest.mark.parametrize("case", ENC, ids=[c[0] for c in ENC])
def test_encoder_conv_shapes(gpu_device, case, B):
    """Encoder / pose-decoder convolutions at their 480x640 sizes: batch 2, the bench's 12, and PoseNet's 24 = 2 x 12
    with per-pair statistics (stat_groups = 2)."""
    from deep_visual_slam_amd import conv as DC
    name, ci, co, k, s, p, refl, Hi, Wi, act, has_b, stats = case
    groups = 2 if B == 24 else 1
    torch.manual_seed(0)
    x = torch.randn(B, ci, Hi, Wi, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, ci, k, k, device=gpu_device) * (2.0 / (ci * k * k)) ** 0.5).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True) if has_b else None
    y_ref = ref_act(F.conv2d(x, w, b, s, p), act)
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [x, w] + ([b] if has_b else []), cot)
    out = DC.conv2d(x, w, b, s, p, 0, act, want_stats=groups if stats else 0)
    y, st = out if stats else (out, None)
    assert y.shape == y_ref.shape
    e = relmax(y, y_ref)
    report("conv %-22s B=%-2d y    relmax %.2e" % (name, B, e))
    assert e < 2e-5
    if stats:
        yr = y_ref.detach().view(groups, B // groups, co, *y_ref.shape[2:])
        st = st.view(groups, 2, co)
        assert relmax(st[:, 0], yr.sum((1, 3, 4))) < 1e-4
        assert relmax(st[:, 1], (yr ** 2).sum((1, 3, 4))) < 1e-4
    g = torch.autograd.grad(y, [x, w] + ([b] if has_b else []), cot)
    _check_grads("%s B=%d" % (name, B), g, g_ref, ("dx", "dw", "db"))


@pytest.mark.parametrize("B", [2, 12, 24])
@pytest.mark.parametrize("cin", [3, 6])
def test_stem_full_size(gpu_device, cin, B):
    """conv1 7x7 s2 from the planar 480x640 image with (x - 0.45) / 0.225 folded in (resnet_encoder.py:102-103) and the
    BatchNorm statistics epilogue."""
    from deep_visual_slam_amd import conv as DC
    groups = 2 if B == 24 else 1
    torch.manual_seed(2)
    x = torch.rand(B, cin, H, W, device=gpu_device)
    w = (torch.randn(64, cin, 7, 7, device=gpu_device) * 0.05).requires_grad_(True)
    y_ref = F.conv2d((x - 0.45) / 0.225, w, None, 2, 3)
    cot = torch.randn_like(y_ref)
    (gw_ref,) = torch.autograd.grad(y_ref, [w], cot)
    sc = torch.full((cin,), 1 / 0.225, device=gpu_device)
    sh = torch.full((cin,), -0.45 / 0.225, device=gpu_device)
    y, st = DC.conv2d(x, w, None, 2, 3, 0, None, planar_norm=(sc, sh), want_stats=groups)
    e = relmax(y, y_ref)
    report("stem cin=%d B=%-2d y relmax %.2e" % (cin, B, e))
    assert e < 2e-5
    yr = y_ref.detach().view(groups, B // groups, 64, 240, 320)
    st = st.view(groups, 2, 64)
    assert relmax(st[:, 0], yr.sum((1, 3, 4))) < 1e-4 and relmax(st[:, 1], (yr ** 2).sum((1, 3, 4))) < 1e-4
    (gw,) = torch.autograd.grad(y, [w], cot)
    _check_grads("stem%d B=%d" % (cin, B), [gw], [gw_ref], ("dw",))


@pytest.mark.parametrize("B", [2, 12])
@pytest.mark.parametrize("case", DEC_PLAIN, ids=[c[0] for c in DEC_PLAIN])
def test_decoder_plain_conv_shapes(gpu_device, case, B):
    from deep_visual_slam_amd import conv as DC
    name, ci, co, Hi, Wi = case
    torch.manual_seed(1)
    x = torch.randn(B, ci, Hi, Wi, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, ci, 3, 3, device=gpu_device) * (2.0 / (ci * 9)) ** 0.5).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True)
    y_ref = F.elu(F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [x, w, b], cot)
    y = DC.conv2d(x, w, b, 1, 0, 1, "elu")
    e = relmax(y, y_ref)
    report("conv %-22s B=%-2d y    relmax %.2e" % (name, B, e))
    assert e < 2e-5
    g = torch.autograd.grad(y, [x, w, b], cot)
    _check_grads("%s B=%d" % (name, B), g, g_ref, ("dx", "dw", "db"))


@pytest.mark.parametrize("B", [2, 12])
@pytest.mark.parametrize("case", DEC_CAT, ids=[c[0] for c in DEC_CAT])
def test_decoder_upsample_concat_shapes(gpu_device, case, B):
    from deep_visual_slam_amd import conv as DC
    name, c1, c2, co, Ho, Wo = case
    torch.manual_seed(1)
    xa = torch.randn(B, c1, Ho // 2, Wo // 2, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    xb = torch.randn(B, c2, Ho, Wo, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, c1 + c2, 3, 3, device=gpu_device) * (2.0 / ((c1 + c2) * 9)) ** 0.5).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True)
    cat = torch.cat([F.interpolate(xa, scale_factor=2, mode="nearest"), xb], 1)
    y_ref = F.elu(F.conv2d(F.pad(cat, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [xa, xb, w, b], cot)
    del cat
    y = DC.conv2d(xa, w, b, 1, 0, 1, "elu", x2=xb)
    e = relmax(y, y_ref)
    report("conv %-22s B=%-2d y    relmax %.2e" % (name, B, e))
    assert e < 2e-5
    g = torch.autograd.grad(y, [xa, xb, w, b], cot)
    _check_grads("%s B=%d" % (name, B), g, g_ref, ("dxa", "dxb", "dw", "db"))


@pytest.mark.parametrize("B", [2, 12])
def test_decoder_up0_1_full_resolution(gpu_device, B):
    """upconv(0, 1): upsample-only 16 -> 16 at 480x640, the widest map of the network."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(2)
    xa = torch.randn(B, 16, H // 2, W // 2, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(16, 16, 3, 3, device=gpu_device) * 0.12).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(16, device=gpu_device) * 0.1).requires_grad_(True)
    up = F.interpolate(xa, scale_factor=2, mode="nearest")
    y_ref = F.elu(F.conv2d(F.pad(up, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [xa, w, b], cot)
    del up
    y = DC.conv2d(xa, w, b, 1, 0, 1, "elu", upsample=True)
    e = relmax(y, y_ref)
    report("conv up0_1 B=%-2d y relmax %.2e" % (B, e))
    assert e < 2e-5
    g = torch.autograd.grad(y, [xa, w, b], cot)
    _check_grads("up0_1 B=%d" % B, g, g_ref, ("dxa", "dw", "db"))


@pytest.mark.parametrize("B", [2, 12])
@pytest.mark.parametrize("case", HEADS, ids=[c[0] for c in HEADS])
def test_disparity_heads_full_size(gpu_device, case, B):
    from deep_visual_slam_amd import nn_ops
    name, ci, Hi, Wi = case
    torch.manual_seed(4)
    x = torch.randn(B, ci, Hi, Wi, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(1, ci, 3, 3, device=gpu_device) * 0.1).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(1, device=gpu_device) * 0.1).requires_grad_(True)
    y_ref = torch.sigmoid(F.conv2d(F.pad(x, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [x, w, b], cot)
    y = nn_ops.conv2d(x, w, b, 1, 0, reflect_pad=1, act="sigmoid")
    assert relmax(y, y_ref) < 2e-5
    g = torch.autograd.grad(y, [x, w, b], cot)
    _check_grads("%s B=%d" % (name, B), g, g_ref, ("dx", "dw", "db"))


# ------------------------------------------------------------------------------------------------ (b)
# Gradient yardstick.  The backward pass through ~20 ReLUs / the maxpool is discontinuous: an element whose pre-activation
# is within rounding of zero takes the other branch in another fp32 implementation, and ONE flipped element moves a
# gradient tensor by ~1/sqrt(N) of its norm (N = elements of that activation: 1.3e-3 at layer 3), which the training-mode
# BatchNorms then spread over the whole channel.  tools/grad_truth.py (profiles/r02_grad_truth_*.txt) shows it: against
# an fp64 run of the oracle, the reference's OWN fp32 CPU arithmetic is off by 3-5e-3 on the encoder tensors, exactly
# like the HIP path, while tensors before the first flip agree to 1e-6 on both.  So the gradients are judged against
# the fp64 oracle with the fp32 oracle as the yardstick: the HIP path must be as close to the truth as the reference's
# arithmetic is -- per tensor within 3x the worst fp32-CPU tensor, in the median within 2x the fp32-CPU median.
def _fresh_nets(dev, seed=0):
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.posenet_single import PoseNet
    torch.manual_seed(seed)
    dn, pn = DepthNet(18, pretrained=False), PoseNet(18, pretrained=False, num_input_images=2)
    sd_d = {k: v.clone() for k, v in dn.state_dict().items()}
    sd_p = {k: v.clone() for k, v in pn.state_dict().items()}
    return dn.to(dev).train(), pn.to(dev).train(), sd_d, sd_p


def _grad_sd(sd, dtype=torch.float32):
    return {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(
        v.is_floating_point() and ".fc." not in k and "running" not in k) for k, v in sd.items()}


def _median(v):
    v = sorted(v)
    return v[len(v) // 2]


def _judge_grads(tag, module, sd32, sd64, n_min_relu, worst_n=3):
    """n_min_relu: elements of the smallest ReLU output on the path (the unit of one branch flip: 1/sqrt(n))."""
    # judged per LAYER (weight and bias gradients concatenated): a bias gradient is one number per channel -- for the
    # 1-channel disparity heads a single, heavily cancelling sum over every pixel -- whose rounding error is only
    # meaningful against the scale of its layer's gradient, not against itself
    groups = {}
    for n, p in module.named_parameters():
        g64 = sd64[n].grad
        if ".fc." in n or g64 is None:
            assert p.grad is None and sd32[n].grad is None, n        # unused tensors: no gradient on either side
            continue
        assert p.grad is not None, n
        assert p.grad.shape == g64.shape, n
        groups.setdefault(n.rsplit(".", 1)[0], []).append((p.grad.detach().double().cpu().reshape(-1),
                                                           sd32[n].grad.double().reshape(-1), g64.reshape(-1)))
    rows = []
    for n, parts in groups.items():
        g, c, t = (torch.cat([q[i] for q in parts]) for i in range(3))
        den = float(t.norm()) + 1e-300
        rows.append((float((g - t).norm()) / den, float((c - t).norm()) / den, n))
    worst_cpu = max(r[1] for r in rows)
    med_gpu, med_cpu = _median([r[0] for r in rows]), _median([r[1] for r in rows])
    for e, c, n in sorted(rows, reverse=True)[:worst_n]:
        report("%s grad vs f64: gpu %.2e cpu32 %.2e  %s" % (tag, e, c, n))
    report("%s grad vs f64 over %d layers: worst gpu %.2e cpu32 %.2e, median gpu %.2e cpu32 %.2e"
           % (tag, len(rows), max(r[0] for r in rows), worst_cpu, med_gpu, med_cpu))
    # per tensor: within 3x the reference arithmetic's own worst tensor, plus two branch flips of the smallest ReLU map (the
    # two implementations do not flip the same elements); in the median: within 2x the reference's median
    flip = 1.0 / n_min_relu ** 0.5
    bad = [(e, n) for e, c, n in rows if not e <= 3.0 * worst_cpu + 2.0 * flip]
    assert not bad, (bad[:8], worst_cpu, flip)
    assert med_gpu <= 2.0 * med_cpu + flip, (med_gpu, med_cpu)
    return rows


def _oracle_threads():
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))


def test_depthnet_full_size_forward_and_all_weight_gradients(gpu_device):
    """DepthNet at 480x640, batch 2, training-mode BatchNorm: four disparity maps against the fp32 oracle, all 14.3 M
    weight gradients against the fp64 oracle with the fp32 oracle as yardstick (model/depthnet.py:64-90)."""
    from oracle import networks as ON
    _oracle_threads()
    dn, _, sd_d, _ = _fresh_nets(gpu_device)
    torch.manual_seed(3)
    x = torch.rand(2, 3, H, W)
    sd32, sd64 = _grad_sd(sd_d), _grad_sd(sd_d, torch.float64)
    upd = {}
    ref = ON.depthnet(x, sd32, train=True, update=upd)
    ref64 = ON.depthnet(x.double(), sd64, train=True)
    out = dn(x.to(gpu_device))
    cots = [torch.randn(ref[("disp", s)].shape) / ref[("disp", s)][0].numel() ** 0.5 for s in range(4)]
    for s in range(4):
        e = rel(out[("disp", s)], ref[("disp", s)])
        report("depthnet 480x640 disp%d rel-L2 %.2e (fp32 oracle vs fp64: %.2e)" % (s, e, rel(ref[("disp", s)], ref64[("disp", s)])))
        assert e < 2e-5                                  # ~40 layers deep, measured 6e-8 .. 7e-7
    sum((ref[("disp", s)] * cots[s]).sum() for s in range(4)).backward()
    sum((ref64[("disp", s)] * cots[s].double()).sum() for s in range(4)).backward()
    dn.zero_grad(set_to_none=True)
    sum((out[("disp", s)] * cots[s].to(gpu_device)).sum() for s in range(4)).backward()
    torch.cuda.synchronize()
    rows = _judge_grads("depthnet 480x640 B=2", dn, sd32, sd64, 2 * 15 * 20 * 512)
    # the decoder's own gradients sit in front of every BatchNorm / ReLU of the backward pass: no flips, tight agreement
    for e, c, n in rows:
        if n.startswith("decoder."):
            assert e < 5e-5, (n, e)
    new = dn.state_dict()
    for k in ("encoder.encoder.bn1.running_mean", "encoder.encoder.layer4.1.bn2.running_var",
              "encoder.encoder.layer3.0.downsample.1.running_var"):
        assert rel(new[k], upd[k]) < 1e-4, k


def test_posenet_full_size_forward_and_all_weight_gradients(gpu_device):
    from oracle import networks as ON
    _oracle_threads()
    _, pn, _, sd_p = _fresh_nets(gpu_device)
    torch.manual_seed(4)
    x = torch.rand(2, 6, H, W)
    sd32, sd64 = _grad_sd(sd_p), _grad_sd(sd_p, torch.float64)
    aa_r, t_r = ON.posenet(x, sd32, train=True)
    aa_d, t_d = ON.posenet(x.double(), sd64, train=True)
    aa, t = pn(x.to(gpu_device))
    report("posenet 480x640 axisangle rel-L2 %.2e translation %.2e" % (rel(aa, aa_r), rel(t, t_r)))
    assert rel(aa, aa_r) < 2e-5 and rel(t, t_r) < 2e-5
    cot = torch.randn(2, 1, 1, 3)
    ((aa_r + t_r) * cot).sum().backward()
    ((aa_d + t_d) * cot.double()).sum().backward()
    pn.zero_grad(set_to_none=True)
    ((aa + t) * cot.to(gpu_device)).sum().backward()
    torch.cuda.synchronize()
    _judge_grads("posenet 480x640 B=2", pn, sd32, sd64, 2 * 15 * 20 * 256)      # smallest ReLU map: the pose decoder's


# ------------------------------------------------------------------------------------------------ (c)
def _oracle_step(sample, sd_d, sd_p, noise, num_scales, dtype):
    from oracle import loss_chain as OL, networks as ON
    cast = lambda t: t.to(dtype) if t.is_floating_point() else t
    smp = {k: cast(v) for k, v in sample.items()}
    sdd, sdp = _grad_sd(sd_d, dtype), _grad_sd(sd_p, dtype)
    tgt, left, right = smp[("target_image", 0)], smp[("source_left", 0)], smp[("source_right", 0)]
    disp = ON.depthnet(tgt, sdd, train=True)
    aa_l, t_l = ON.posenet(torch.cat([left, tgt], 1), sdp, train=True)
    aa_r, t_r = ON.posenet(torch.cat([tgt, right], 1), sdp, train=True)
    _, losses = OL.loss_chain(smp, [disp[("disp", s)] for s in range(num_scales)], (aa_l, t_l, aa_r, t_r),
                              [cast(n) for n in noise], num_scales=num_scales, dtype=dtype)
    losses["loss"].backward()
    return sdd, sdp, disp, (aa_l, t_l, aa_r, t_r), {k: float(v) for k, v in losses.items()}


def _full_step(gpu_device, B, num_scales):
    from deep_visual_slam_amd import gradsink, synth
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    _oracle_threads()
    dn, pn, sd_d, sd_p = _fresh_nets(gpu_device)
    sample = synth.parity_sample(B, H, W)
    g = torch.Generator().manual_seed(7)
    noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(num_scales)]
    d32, p32, disp, (aa_l, t_l, aa_r, t_r), ref = _oracle_step(sample, sd_d, sd_p, noise, num_scales, torch.float32)
    d64, p64, _, _, ref64 = _oracle_step(sample, sd_d, sd_p, noise, num_scales, torch.float64)
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                         ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
    tr = MonodepthTrainer(dn, pn, cfg, gpu_device)
    tr.num_scales = num_scales
    tr._noise = torch.stack(noise).to(gpu_device)
    outputs, losses = tr.process_batch(dict(sample))
    keys = ["loss"] + ["loss/%d" % s for s in range(num_scales)]
    assert sorted(losses) == sorted(keys)
    for k in keys:
        e = abs(float(losses[k]) - ref[k]) / abs(ref[k])
        report("step B=%d S=%d %-7s gpu %.8f oracle %.8f rel %.2e (fp32 oracle vs fp64: %.2e)"
               % (B, num_scales, k, float(losses[k]), ref[k], e, abs(ref[k] - ref64[k]) / abs(ref64[k])))
        assert e < 2e-4, k
    for s in range(num_scales):
        assert rel(outputs[("disp", s)], disp[("disp", s)]) < 2e-5
    for f, (aa, t) in ((-1, (aa_l, t_l)), (1, (aa_r, t_r))):
        assert rel(outputs[("axisangle", 0, f)], aa) < 2e-5 and rel(outputs[("translation", 0, f)], t) < 2e-5
    dn.zero_grad(set_to_none=True)
    pn.zero_grad(set_to_none=True)
    losses["loss"].backward()
    gradsink.join()
    torch.cuda.synchronize()
    tag = "step B=%d S=%d" % (B, num_scales)
    _judge_grads(tag + " depth", dn, d32, d64, B * 15 * 20 * 512)
    _judge_grads(tag + " pose ", pn, p32, p64, B * 15 * 20 * 256)
    # lazily materialised outputs at full size keep the reference's schema
    assert outputs[("color", -1, 0)].shape == (B, 3, H, W) and outputs[("sample", 1, 0)].shape == (B, H, W, 2)


def test_full_step_config1_batch4_single_scale(gpu_device):
    """BASELINE.json configs[1]: 3-frame 640x480 snippet, batch 4, single-scale loss (trainer.num_scales = 1)."""
    _full_step(gpu_device, 4, 1)


def test_full_step_config2_batch12_four_scales(gpu_device):
    """BASELINE.json configs[2] (= configs[3] per GPU): full 4-scale photometric + smoothness loss, batch 12."""
    _full_step(gpu_device, 12, 4)
