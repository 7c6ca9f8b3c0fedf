"""The opt-in bf16 mode of the implicit-GEMM convolutions (dvs_set_precision(1); include/dvslam.h) -- the counterpart of the
reference's `use_amp` branch (vo/train.py:44,177-185).  NOT the parity mode: its own tolerance table.

What the mode computes is exactly specified -- every operand of the product rounded to bf16 (nearest even), exact products,
fp32 accumulation -- so the kernels are checked against THAT, evaluated by torch in fp64 on bf16-rounded operands:

    forward        y  = conv(bf16(x), bf16(w))                       2e-5 of the tensor max (fp32 summation order)
    data gradient  dx = conv_transpose(bf16(dy), bf16(w))            1e-4
    weight grad    dw = correlate(bf16(dy), bf16(x))                 1e-4

and against the fp32 result, which bounds what the mode costs in accuracy (operand rounding 2^-9 relative per factor, averaging
down over K): 1e-2 of the tensor max for single layers, 3e-2 on the loss of a whole training step.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def relmax(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def r16(t):
    return t.detach().to(torch.bfloat16).to(torch.float64)


@pytest.fixture
def bf16_mode():
    from deep_visual_slam_amd import _lib
    _lib.set_precision("bf16")
    yield
    _lib.set_precision("fp32")


CASES = [
    # name, B, Cin, Cout, k, stride, pad, reflect, H, W
    ("3x3_s1_64", 2, 64, 64, 3, 1, 1, False, 24, 40),
    ("3x3_s1_128", 2, 128, 128, 3, 1, 1, False, 15, 20),
    ("3x3_s2", 2, 64, 128, 3, 2, 1, False, 24, 40),
    ("1x1_s2", 2, 64, 128, 1, 2, 0, False, 24, 40),
    ("3x3_deep", 3, 256, 512, 3, 1, 1, False, 9, 13),
    ("1x1_gemm", 2, 512, 256, 1, 1, 0, False, 7, 9),
    ("refl_wide", 2, 128, 64, 3, 1, 1, True, 12, 20),
    ("refl_96", 2, 64, 96, 3, 1, 1, True, 12, 20),
    ("odd_tail", 1, 20, 36, 3, 1, 1, False, 11, 17),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_bf16_matches_its_specification(gpu_device, bf16_mode, case):
    from deep_visual_slam_amd import conv as DC
    name, B, ci, co, k, s, p, refl, H, W = case
    torch.manual_seed(0)
    x = torch.randn(B, ci, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, ci, k, k, device=gpu_device) * (2.0 / (ci * k * k)) ** 0.5).contiguous(memory_format=CL).requires_grad_(True)
    y = DC.conv2d(x, w, None, s, p, 1 if refl else 0, None)
    cot = torch.randn_like(y)
    dx, dw = torch.autograd.grad(y, [x, w], cot)

    def ref(xv, wv, cv):
        xv = xv.clone().requires_grad_(True)
        wv = wv.clone().requires_grad_(True)
        xx = F.pad(xv, (p,) * 4, mode="reflect") if refl else xv
        yr = F.conv2d(xx, wv, None, s, 0 if refl else p)
        return yr, xv, wv, cv

    # the specification: bf16-rounded operands, exact arithmetic (fp64)
    yr, xv, wv, _ = ref(r16(x), r16(w), None)
    assert relmax(y, yr) < 2e-5, relmax(y, yr)
    (dx_r,) = torch.autograd.grad(yr, [xv], r16(cot), retain_graph=True)        # dgrad: bf16(dy) x bf16(w)
    # (ReflectionPad2d: the data-gradient gather adds the mirrored taps' dy values in fp32 BEFORE the bf16 rounding -- one product
    # of bf16(dy_a + dy_b) instead of two -- so border pixels differ from the two-product specification by one operand rounding)
    assert relmax(dx, dx_r) < (5e-3 if refl else 1e-4), relmax(dx, dx_r)
    (dw_r,) = torch.autograd.grad(yr, [wv], r16(cot))                           # wgrad: bf16(dy) x bf16(x)
    assert relmax(dw, dw_r) < 1e-4, relmax(dw, dw_r)
    # the cost against fp32
    yf, xf, wf, _ = ref(x.detach().double(), w.detach().double(), None)
    gxf, gwf = torch.autograd.grad(yf, [xf, wf], cot.double())
    assert relmax(y, yf) < 1e-2 and relmax(dx, gxf) < 1e-2 and relmax(dw, gwf) < 1e-2


def test_decoder_block_bf16(gpu_device, bf16_mode):
    """upsample ; cat ; ReflectionPad ; 3x3 ; ELU (model/depthnet.py:79-88) in the bf16 mode against fp32 torch."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(1)
    B, c1, c2, co, H, W = 2, 256, 256, 128, 10, 14
    xa = torch.randn(B, c1, H // 2, W // 2, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    xb = torch.randn(B, c2, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, c1 + c2, 3, 3, device=gpu_device) * 0.03).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True)
    cat = torch.cat([F.interpolate(xa, scale_factor=2, mode="nearest"), xb], 1)
    y_ref = F.elu(F.conv2d(F.pad(cat, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [xa, xb, w, b], cot)
    y = DC.conv2d(xa, w, b, 1, 0, 1, "elu", x2=xb)
    assert relmax(y, y_ref) < 1e-2
    g = torch.autograd.grad(y, [xa, xb, w, b], cot)
    for a, r, nm in zip(g, g_ref, ("dxa", "dxb", "dw", "db")):
        assert a.shape == r.shape and relmax(a, r) < 1e-2, (nm, relmax(a, r))


THIN = [
    # c1, c2 (None: no skip; 0: upsample only), co, H, W (of the output), act
    (64, None, 32, 24, 40, "elu"),       # upconv_1_0
    (32, 64, 32, 24, 80, "elu"),         # upconv_1_1: cat(up(32), 64)
    (32, None, 16, 17, 45, "elu"),       # upconv_0_0 (odd sizes)
    (16, 0, 16, 36, 66, "elu"),          # upconv_0_1: upsample only
    (48, None, 32, 9, 33, None),         # 16-channel chunks
]


@pytest.mark.parametrize("c1,c2,co,H,W,act", THIN, ids=["%s_%s_%s" % (c[0], c[1], c[2]) for c in THIN])
def test_thin_decoder_layers_bf16(gpu_device, bf16_mode, c1, c2, co, H, W, act):
    """The decoder's 32- / 16-channel levels (model/depth_decoder.py:52-62) on the thin patch kernel: forward against the bf16
    specification, gradients against fp32 torch."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(5)
    B = 2
    up = c2 is not None
    xa = torch.randn(B, c1, H // 2 if up else H, W // 2 if up else W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    xb = (torch.randn(B, c2, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True) if c2 else None)
    ci = c1 + (c2 or 0)
    w = (torch.randn(co, ci, 3, 3, device=gpu_device) * (2.0 / (ci * 9)) ** 0.5).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True)

    def ref(xa_, xb_, w_, b_):
        t = F.interpolate(xa_, scale_factor=2, mode="nearest") if up else xa_
        if xb_ is not None:
            t = torch.cat([t, xb_], 1)
        y_ = F.conv2d(F.pad(t, (1,) * 4, mode="reflect"), w_, b_)
        return F.elu(y_) if act == "elu" else y_

    if not up:
        y = DC.conv2d(xa, w, b, 1, 1, 1, act)
    elif c2:
        y = DC.conv2d(xa, w, b, 1, 0, 1, act, x2=xb)
    else:
        y = DC.conv2d(xa, w, b, 1, 0, 1, act, upsample=True)
    y_spec = ref(r16(xa), r16(xb) if xb is not None else None, r16(w), b.detach().double())
    assert y.shape == y_spec.shape and relmax(y, y_spec) < 2e-5, relmax(y, y_spec)
    y_ref = ref(xa, xb, w, b)
    cot = torch.randn_like(y_ref)
    ins = [xa] + ([xb] if xb is not None else []) + [w, b]
    g_ref = torch.autograd.grad(y_ref, ins, cot)
    g = torch.autograd.grad(y, ins, cot)
    for a, r_, nm in zip(g, g_ref, ("dxa", "dxb", "dw", "db") if xb is not None else ("dxa", "dw", "db")):
        assert a.shape == r_.shape and relmax(a, r_) < 1e-2, (nm, relmax(a, r_))


def test_bf16_statistics_epilogue(gpu_device, bf16_mode):
    """The BatchNorm statistics the convolution's epilogue takes are sums of the values it STORES (fp32), also in the bf16 mode."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(3)
    x = torch.randn(2, 64, 20, 24, device=gpu_device).contiguous(memory_format=CL)
    w = (torch.randn(128, 64, 3, 3, device=gpu_device) * 0.05).contiguous(memory_format=CL)
    stats = torch.zeros(2, 128, device=gpu_device)
    y = DC.conv2d_forward(x, w, None, 1, 1, False, None, stats=stats)
    assert relmax(stats[0], y.double().sum((0, 2, 3))) < 1e-5
    assert relmax(stats[1], (y.double() ** 2).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("B,ci,co,H,W,groups,slots", [(2, 64, 64, 24, 40, 1, 1), (4, 64, 128, 15, 20, 2, 16), (3, 128, 64, 9, 13, 1, 4),
                                                     (2, 512, 512, 15, 20, 2, 16), (1, 256, 256, 30, 40, 0, 1)])
def test_patch_kernel_forward_stats_residual(gpu_device, bf16_mode, B, ci, co, H, W, groups, slots):
    """csrc/conv_p16.hip directly: forward with the BatchNorm statistics ([slots][G][2][C], G = halves of the batch), the data
    gradient form with a residual, image sizes that are not multiples of the 8 x 16 patch."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(4)
    x = torch.randn(B, ci, H, W, device=gpu_device).contiguous(memory_format=CL)
    w = (torch.randn(co, ci, 3, 3, device=gpu_device) * (2.0 / (ci * 9)) ** 0.5).contiguous(memory_format=CL)
    stats = torch.zeros((slots, groups, 2, co) if slots > 1 else ((groups, 2, co) if groups == 2 else (2, co)), device=gpu_device) if groups else None
    y = DC.conv3x3_p16(x, w, stats, groups, stat_slots=slots)
    yr = F.conv2d(r16(x), r16(w), None, 1, 1)
    assert relmax(y, yr) < 2e-5, relmax(y, yr)
    if groups:
        st = stats.double().sum(0) if slots > 1 else stats.double()
        st = st.reshape(groups, 2, co)
        for g in range(groups):
            yy = y.double()[g * B // groups:(g + 1) * B // groups]
            assert relmax(st[g, 0], yy.sum((0, 2, 3))) < 1e-5
            assert relmax(st[g, 1], (yy ** 2).sum((0, 2, 3))) < 1e-5
    dy = torch.randn(B, co, H, W, device=gpu_device).contiguous(memory_format=CL)
    res = torch.randn(B, ci, H, W, device=gpu_device).contiguous(memory_format=CL)
    dx = DC.conv3x3_p16(dy, w, flip=True, residual=res)
    dxr = F.conv_transpose2d(r16(dy), r16(w), None, 1, 1) + res.double()
    assert relmax(dx, dxr) < 1e-4, relmax(dx, dxr)
    # weight gradient: bf16(dy) x bf16(x), into a zero-filled tensor and, a second time, on top of it (it ADDS)
    dw = DC.conv3x3_p16_wgrad(x, dy, (co, ci, 3, 3))
    wv = r16(w).requires_grad_(True)
    (dwr,) = torch.autograd.grad(F.conv2d(r16(x), wv, None, 1, 1), [wv], r16(dy))
    assert dw.shape == dwr.shape and relmax(dw, dwr) < 1e-4, relmax(dw, dwr)
    DC.conv3x3_p16_wgrad(x, dy, (co, ci, 3, 3), dw_out=dw)
    assert relmax(dw, 2 * dwr) < 1e-4


def test_training_step_bf16_against_fp32(gpu_device):
    """One whole training step (both networks, loss chain, backward) in the two modes from the same weights: the loss within 3e-2
    relative, every network gradient within 0.2 of the fp32 gradient's norm in total (cosine > 0.98)."""
    import sys
    sys.path.insert(0, ".")
    import bench
    from deep_visual_slam_amd import _lib
    out = {}
    for mode in ("fp32", "bf16"):
        _lib.set_precision(mode)
        try:
            torch.manual_seed(0)
            trainer, flat, sync, opt, sample = bench.build_gpu(2, 4, gpu_device, 0)
            _, losses = trainer.process_batch(sample)
            losses["loss"].backward()
            sync.finish()
            torch.cuda.synchronize()
            out[mode] = (float(losses["loss"]), flat.grads.detach().clone())
            del trainer, flat, sync, opt, sample
        finally:
            _lib.set_precision("fp32")
    (l32, g32), (l16, g16) = out["fp32"], out["bf16"]
    assert abs(l16 - l32) / abs(l32) < 3e-2, (l16, l32)
    cos = float((g16.double() @ g32.double()) / (g16.double().norm() * g32.double().norm()))
    assert cos > 0.98, cos


def test_trainer_opt_in_follows_autocast(gpu_device):
    """Train.amp_bf16: the mode is armed while process_batch runs under torch.autocast (the use_amp branch of vo/train.py:177-185),
    stays for the backward the caller runs after leaving autocast, and is dropped by the next call outside autocast."""
    import sys
    sys.path.insert(0, ".")
    import bench
    from torch.amp import GradScaler, autocast
    from deep_visual_slam_amd import _lib, dp, synth
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    try:
        torch.manual_seed(0)
        dn = DepthNet(18, pretrained=False).to(gpu_device).train()
        pn = PoseNet(18, pretrained=False, num_input_images=2).to(gpu_device).train()
        cfg = bench.train_config(2, 4)
        cfg["Train"]["amp_bf16"] = True
        flat = dp.FlatParams(dp.trainable_parameters(dn, pn))
        opt = dp.FusedAdam(flat, lr=1e-4, params=list(dn.parameters()) + list(pn.parameters()))
        tr = MonodepthTrainer(dn, pn, cfg, gpu_device)
        sample = synth.throughput_sample(2, bench.H, bench.W, rank=0, device=gpu_device)
        scaler = GradScaler()
        assert _lib.precision() == "fp32"
        with torch.no_grad():
            _, l32 = tr.process_batch(dict(sample))            # not under autocast: fp32
        assert _lib.precision() == "fp32"
        tr._step = 0
        opt.zero_grad(set_to_none=True)
        with autocast(device_type="cuda"):
            _, l16 = tr.process_batch(dict(sample))
        assert _lib.precision() == "bf16"
        scaler.scale(l16["loss"]).backward()                   # outside autocast, still the mode of this step
        assert _lib.precision() == "bf16"
        scaler.step(opt)
        scaler.update()
        a, b = float(l32["loss"]), float(l16["loss"])
        assert a != b and abs(a - b) / abs(a) < 3e-2, (a, b)
        assert all(torch.isfinite(p.grad).all() for p in dn.parameters() if p.grad is not None)
        with torch.no_grad():
            tr.process_batch(dict(sample))
        assert _lib.precision() == "fp32"
    finally:
        _lib.set_precision("fp32")


def test_weight_pack_follows_the_fused_optimiser(gpu_device, bf16_mode):
    """dp.FusedAdam updates the weights through a raw pointer (no torch version bump): the bf16 weight packs of the patch kernels
    must not outlive that step."""
    from deep_visual_slam_amd import conv as DC, dp
    torch.manual_seed(6)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(gpu_device).to(memory_format=CL)
    flat = dp.FlatParams([("w", conv.weight)])
    opt = dp.FusedAdam(flat, lr=0.05)
    w = conv.weight
    x = torch.randn(2, 64, 16, 24, device=gpu_device).contiguous(memory_format=CL)
    y0 = DC.conv3x3_p16(x, w)
    assert relmax(y0, F.conv2d(r16(x), r16(w), None, 1, 1)) < 2e-5
    w.grad.copy_(torch.randn_like(w))
    opt.step()
    torch.cuda.synchronize()
    y1 = DC.conv3x3_p16(x, w)
    y1_ref = F.conv2d(r16(x), r16(w), None, 1, 1)
    assert relmax(y1, y1_ref) < 2e-5, relmax(y1, y1_ref)
    assert relmax(y1, y0) > 1e-2              # the step did move the weights


@pytest.mark.parametrize("cin,B,H,W", [(3, 2, 38, 50), (6, 4, 64, 132), (6, 2, 37, 259)])
def test_stem_forward_bf16(gpu_device, bf16_mode, cin, B, H, W):
    """conv1 (7x7, stride 2, pad 3, planar image with the input normalisation folded in; model/resnet_encoder.py:102-103,141) on the
    bf16 stem kernel: operands = bf16(normalised image), bf16(weights); BatchNorm statistics from the fp32 results, two batch groups."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(7)
    x = torch.rand(B, cin, H, W, device=gpu_device)
    w = (torch.randn(64, cin, 7, 7, device=gpu_device) * 0.05)
    sc = torch.full((cin,), 1 / 0.225, device=gpu_device)
    sh = torch.full((cin,), -0.45 / 0.225, device=gpu_device)
    stats = torch.zeros(2, 2, 64, device=gpu_device)
    y = DC.conv2d_forward(x, w, None, 2, 3, False, None, in_scale=sc, in_shift=sh, nchw_planar=True, stats=stats, stat_groups=2)
    xn = x * sc[None, :, None, None] + sh[None, :, None, None]
    y_spec = F.conv2d(r16(xn), r16(w), None, 2, 3)
    # (the kernel normalises with one fma, torch with a multiply and an add: a last-bit difference in fp32 that flips the bf16
    # rounding of a few image operands -- 2^-9 of one product each)
    assert y.shape == y_spec.shape and relmax(y, y_spec) < 2e-4, relmax(y, y_spec)
    for g in range(2):
        yy = y.double()[g * B // 2:(g + 1) * B // 2]
        assert relmax(stats[g, 0], yy.sum((0, 2, 3))) < 1e-5
        assert relmax(stats[g, 1], (yy ** 2).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("cin,B,H,W", [(3, 2, 38, 50), (6, 2, 64, 132), (6, 3, 37, 259)])
def test_stem_weight_gradient_bf16(gpu_device, bf16_mode, cin, B, H, W):
    """Weight gradient of conv1 on the bf16 stem kernel: bf16(dY) x bf16(normalised image), fp32 accumulation."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(8)
    x = torch.rand(B, cin, H, W, device=gpu_device)
    w = (torch.randn(64, cin, 7, 7, device=gpu_device) * 0.05).requires_grad_(True)
    sc = torch.full((cin,), 1 / 0.225, device=gpu_device)
    sh = torch.full((cin,), -0.45 / 0.225, device=gpu_device)
    y = DC.conv2d(x, w, None, 2, 3, 0, None, planar_norm=(sc, sh))
    cot = torch.randn_like(y)
    (gw,) = torch.autograd.grad(y, [w], cot)
    xn = x * sc[None, :, None, None] + sh[None, :, None, None]
    wv = r16(w).requires_grad_(True)
    (gw_spec,) = torch.autograd.grad(F.conv2d(r16(xn), wv, None, 2, 3), [wv], r16(cot))
    assert gw.shape == gw_spec.shape and relmax(gw, gw_spec) < 2e-4, relmax(gw, gw_spec)


@pytest.mark.parametrize("B,N,heads", [(1, 1370, 6), (2, 49, 6), (3, 33, 2)])
def test_attention_forward_bf16(gpu_device, B, N, heads):
    """Inference attention in the mode: both products on bf16 MFMA (scaled q, k, the probabilities and v rounded to bf16, softmax and
    accumulation fp32).  Checked against softmax(q k^T / 8) v in fp32 at 1e-2 of the output max, and it must differ from the fp32
    kernel (the mode is really taken).  The training forward (log-sum-exp kept) stays fp32 -- test_dav2_gpu covers it."""
    from deep_visual_slam_amd import _lib
    from deep_visual_slam_amd.depth_anything_v2 import attention
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3, heads, 64, generator=g) * 1.5
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    ref = ((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * N, heads * 64)
    dev = qkv.reshape(B * N, -1).to(gpu_device)
    y32 = attention(dev, B, N, heads, 64).clone()
    try:
        _lib.set_precision("bf16")
        y16 = attention(dev, B, N, heads, 64)
    finally:
        _lib.set_precision("fp32")
    assert relmax(y16.cpu(), ref) < 1e-2, relmax(y16.cpu(), ref)
    assert relmax(y16, y32) > 0.0


def test_depth_anything_forward_bf16_against_fp32(gpu_device):
    """configs[4] in the mode: the token GEMMs, the attention products and the DPT head's convolutions take bf16 operands (softmax,
    LayerNorm, GELU stay fp32); the depth map of seeded random weights stays within 5e-3 of the fp32 one."""
    from deep_visual_slam_amd import _lib
    from deep_visual_slam_amd.depth_anything_v2 import DepthAnythingV2
    torch.manual_seed(0)
    net = DepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384]).to(gpu_device).eval()
    x = torch.rand(1, 3, 518, 518, device=gpu_device)
    try:
        with torch.no_grad():
            y32 = net(x).float().clone()
            _lib.set_precision("bf16")
            y16 = net(x).float()
    finally:
        _lib.set_precision("fp32")
    assert torch.isfinite(y16).all()
    d = relmax(y16, y32)
    assert 0.0 < d < 5e-3, d
