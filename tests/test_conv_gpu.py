"""GPU parity of the hand-written implicit-GEMM convolution (forward, data gradient, weight/bias
gradient, fused pad / upsample+concat / activation / input normalisation) against torch's own fp32
composition of the reference modules (F.pad reflect, F.interpolate nearest, torch.cat, F.conv2d, F.elu).

Tolerance: exact-fp32 MFMA vs MIOpen fp32 -- both sum K <= 4608 products in a different order:
relative 2e-5 of the tensor's max on values, 1e-4 on gradients (sums over up to 1e5 pixels).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def relmax(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def ref_act(y, act):
    return {None: lambda v: v, "elu": F.elu, "relu": F.relu, "sigmoid": torch.sigmoid}[act](y)


CASES = [
    # name, B, Cin, Cout, k, stride, pad, reflect, H, W, act, bias
    ("3x3_s1", 2, 64, 64, 3, 1, 1, False, 24, 40, None, False),
    ("3x3_s2", 2, 64, 128, 3, 2, 1, False, 24, 40, None, False),
    ("1x1_s2", 2, 64, 128, 1, 2, 0, False, 24, 40, None, False),
    ("3x3_deep", 3, 256, 256, 3, 1, 1, False, 9, 13, None, False),
    ("refl_elu", 2, 32, 16, 3, 1, 1, True, 20, 36, "elu", True),
    ("refl_elu_c16", 1, 16, 16, 3, 1, 1, True, 33, 47, "elu", True),
    ("relu_bias_1x1", 2, 512, 256, 1, 1, 0, False, 7, 9, "relu", True),
    ("relu_bias_3x3", 2, 256, 256, 3, 1, 1, False, 7, 9, "relu", True),
    ("odd_tail", 1, 20, 36, 3, 1, 1, False, 11, 17, None, True),
    # decoder layers with 16 / 32 output channels at segment-aligned widths: the row-ring weight-gradient kernel
    ("thin32_c64", 2, 64, 32, 3, 1, 1, True, 13, 64, "elu", True),
    ("thin16_c32", 2, 32, 16, 3, 1, 1, True, 21, 128, "elu", True),
    ("thin16_c16_nobias", 1, 16, 16, 3, 1, 1, True, 9, 256, None, False),
    # wide decoder layers: pre-activation pass + padded-domain data gradient + reflection fold + LDS-DMA weight gradient
    ("refl_elu_wide", 2, 128, 64, 3, 1, 1, True, 12, 20, "elu", True),
    ("refl_elu_wide_h3", 1, 64, 64, 3, 1, 1, True, 3, 5, "elu", True),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fwd_bwd(gpu_device, case):
    from deep_visual_slam_amd import conv as DC
    name, B, ci, co, k, s, p, refl, H, W, act, has_b = case
    torch.manual_seed(0)
    x = torch.randn(B, ci, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, ci, k, k, device=gpu_device) * (2.0 / (ci * k * k)) ** 0.5).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True) if has_b else None
    xx = F.pad(x, (p,) * 4, mode="reflect") if refl else x
    y_ref = ref_act(F.conv2d(xx, w, b, s, 0 if refl else p), act)
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [x, w] + ([b] if has_b else []), cot)
    y = DC.conv2d(x, w, b, s, p, 1 if refl else 0, act)
    assert y.shape == y_ref.shape and y.is_contiguous(memory_format=CL)
    assert relmax(y, y_ref) < 2e-5
    g = torch.autograd.grad(y, [x, w] + ([b] if has_b else []), cot)
    for a, r, nm in zip(g, g_ref, ("dx", "dw", "db")):
        assert a.shape == r.shape, nm
        assert relmax(a, r) < 1e-4, (nm, relmax(a, r))


@pytest.mark.parametrize("c1,c2,co,H,W", [(32, 64, 32, 24, 40), (256, 256, 128, 10, 14), (32, 64, 32, 26, 96), (32, 64, 32, 20, 128)])
def test_upsample_concat_conv(gpu_device, c1, c2, co, H, W):
    """upsample(x) ; cat([x, skip]) ; ConvBlock (model/depthnet.py:79-88) as one call."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(1)
    B = 2
    xa = torch.randn(B, c1, H // 2, W // 2, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    xb = torch.randn(B, c2, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, c1 + c2, 3, 3, device=gpu_device) * 0.03).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True)
    cat = torch.cat([F.interpolate(xa, scale_factor=2, mode="nearest"), xb], 1)
    y_ref = F.elu(F.conv2d(F.pad(cat, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [xa, xb, w, b], cot)
    y = DC.conv2d(xa, w, b, 1, 0, 1, "elu", x2=xb)
    assert relmax(y, y_ref) < 2e-5
    g = torch.autograd.grad(y, [xa, xb, w, b], cot)
    for a, r, nm in zip(g, g_ref, ("dxa", "dxb", "dw", "db")):
        assert a.shape == r.shape and relmax(a, r) < 1e-4, (nm, relmax(a, r))


@pytest.mark.parametrize("c,co,H,W", [(16, 16, 20, 128), (32, 32, 10, 24)])
def test_upsample_only_conv(gpu_device, c, co, H, W):
    """upsample(x) ; ConvBlock with nothing concatenated (decoder level 0, model/depthnet.py:79-88)."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(2)
    B = 2
    xa = torch.randn(B, c, H // 2, W // 2, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, c, 3, 3, device=gpu_device) * 0.08).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(co, device=gpu_device) * 0.1).requires_grad_(True)
    up = F.interpolate(xa, scale_factor=2, mode="nearest")
    y_ref = F.elu(F.conv2d(F.pad(up, (1,) * 4, mode="reflect"), w, b))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [xa, w, b], cot)
    y = DC.conv2d(xa, w, b, 1, 0, 1, "elu", upsample=True)
    assert relmax(y, y_ref) < 2e-5
    g = torch.autograd.grad(y, [xa, w, b], cot)
    for a, r, nm in zip(g, g_ref, ("dxa", "dw", "db")):
        assert a.shape == r.shape and relmax(a, r) < 1e-4, (nm, relmax(a, r))


@pytest.mark.parametrize("cin", [3, 6])
def test_conv1_planar_with_input_normalisation(gpu_device, cin):
    """(x - 0.45) / 0.225 -> conv 7x7 stride 2 pad 3 from the planar NCHW image (resnet_encoder.py:102-103)."""
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(2)
    x = torch.rand(2, cin, 38, 50, device=gpu_device)
    w = (torch.randn(64, cin, 7, 7, device=gpu_device) * 0.05).requires_grad_(True)
    y_ref = F.conv2d((x - 0.45) / 0.225, w, None, 2, 3)
    cot = torch.randn_like(y_ref)
    (gw_ref,) = torch.autograd.grad(y_ref, [w], cot)
    sc = torch.full((cin,), 1 / 0.225, device=gpu_device)
    sh = torch.full((cin,), -0.45 / 0.225, device=gpu_device)
    y = DC.conv2d(x, w, None, 2, 3, 0, None, planar_norm=(sc, sh))
    assert relmax(y, y_ref) < 2e-5
    (gw,) = torch.autograd.grad(y, [w], cot)
    assert gw.shape == gw_ref.shape and relmax(gw, gw_ref) < 1e-4


def test_bn_fold_and_stats_epilogue(gpu_device):
    from deep_visual_slam_amd import conv as DC
    torch.manual_seed(3)
    x = torch.randn(2, 64, 20, 24, device=gpu_device).contiguous(memory_format=CL)
    w = (torch.randn(64, 64, 3, 3, device=gpu_device) * 0.05).contiguous(memory_format=CL)
    sc, sh = torch.rand(64, device=gpu_device) + 0.5, torch.randn(64, device=gpu_device) * 0.2
    stats = torch.zeros(2, 64, device=gpu_device)
    y = DC.conv2d_forward(x, w, None, 1, 1, False, None, in_scale=sc, in_shift=sh, in_relu=True, stats=stats)
    y_ref = F.conv2d(F.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), w, None, 1, 1)
    assert relmax(y, y_ref) < 2e-5
    assert relmax(stats[0], y_ref.sum((0, 2, 3))) < 1e-4
    assert relmax(stats[1], (y_ref ** 2).sum((0, 2, 3))) < 1e-4


def test_unsupported_shapes_are_reported(gpu_device):
    from deep_visual_slam_amd import conv as DC
    x = torch.zeros(1, 16, 8, 8, device=gpu_device)
    assert not DC.supported(x, torch.zeros(1, 16, 3, 3))       # 1-channel disparity head
    assert not DC.supported(x, torch.zeros(6, 256, 1, 1))      # 6-channel pose head
    assert DC.supported(x, torch.zeros(16, 16, 3, 3))


@pytest.mark.parametrize("cin,cout,k,refl,act,H,W", [(16, 1, 3, True, "sigmoid", 40, 56), (128, 1, 3, True, "sigmoid", 9, 13),
                                                    (256, 6, 1, False, None, 7, 9), (32, 1, 3, True, "sigmoid", 3, 4),
                                                    # the input-indexed weight gradient: zero padding, no activation, widths that are not
                                                    # a multiple of a wave's columns, one-column waves (Cin = 256), runs longer than H
                                                    (16, 1, 3, False, None, 37, 45), (64, 1, 3, True, None, 33, 18),
                                                    (256, 1, 3, True, "sigmoid", 10, 7), (32, 1, 3, False, "sigmoid", 2, 2),
                                                    (16, 1, 3, True, "sigmoid", 70, 33)])
def test_head_conv(gpu_device, cin, cout, k, refl, act, H, W):
    """Disparity heads (reflect-pad 3x3 -> 1 channel -> sigmoid) and PoseNet's 1x1 -> 6 channels."""
    from deep_visual_slam_amd import nn_ops
    torch.manual_seed(4)
    B = 2
    x = torch.randn(B, cin, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, device=gpu_device) * 0.1).contiguous(memory_format=CL).requires_grad_(True)
    b = (torch.randn(cout, device=gpu_device) * 0.1).requires_grad_(True)
    p = (k - 1) // 2
    xx = F.pad(x, (p,) * 4, mode="reflect") if refl else x
    y_ref = ref_act(F.conv2d(xx, w, b, 1, 0 if refl else p), act)
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [x, w, b], cot)
    y = nn_ops.conv2d(x, w, b, 1, 0 if refl else p, reflect_pad=p if refl else 0, act=act)
    assert y.shape == y_ref.shape and relmax(y, y_ref) < 2e-5
    g = torch.autograd.grad(y, [x, w, b], cot)
    for a, r, nm in zip(g, g_ref, ("dx", "dw", "db")):
        assert a.shape == r.shape and relmax(a, r) < 1e-4, (nm, relmax(a, r))


# ---- fan-in gradients added inside the data-gradient kernels (dvs_conv2d_dgrad_res / _head_bwd_res / maxpool _bwd_res)
RES_CASES = [
    # name, B, Cin, Cout, k, stride, pad, H, W      (the encoder's consumers of a block input / feature map)
    ("3x3_s2", 2, 64, 128, 3, 2, 1, 24, 40),        # conv1 of a downsample block
    ("1x1_s2", 2, 64, 128, 1, 2, 0, 24, 40),        # its downsample branch
    ("3x3_s2_deep_splitk", 2, 256, 512, 3, 2, 1, 8, 12),
    ("3x3_s1_direct", 1, 20, 36, 3, 1, 1, 11, 17),  # a stride-1 layer the Winograd kernel does not take (small K split path)
    ("1x1_s2_odd", 1, 64, 128, 1, 2, 0, 15, 21),
]


@pytest.mark.parametrize("case", RES_CASES, ids=[c[0] for c in RES_CASES])
def test_dgrad_with_residual(gpu_device, case):
    """dx = conv data gradient + residual, in one kernel, against torch's data gradient plus the same tensor."""
    from deep_visual_slam_amd import conv as DC
    name, B, ci, co, k, s, p, H, W = case
    torch.manual_seed(3)
    x = torch.randn(B, ci, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w = (torch.randn(co, ci, k, k, device=gpu_device) * (2.0 / (ci * k * k)) ** 0.5).contiguous(memory_format=CL)
    y_ref = F.conv2d(x, w, None, s, p)
    cot = torch.randn_like(y_ref).contiguous(memory_format=CL)
    res = torch.randn_like(x).contiguous(memory_format=CL)
    (dx_ref,) = torch.autograd.grad(y_ref, [x], cot)
    dx = DC.conv2d_dgrad(cot, w, tuple(x.shape), s, p, False, residual=res)
    assert relmax(dx, dx_ref + res) < 1e-4
    dx0 = DC.conv2d_dgrad(cot, w, tuple(x.shape), s, p, False)
    assert relmax(dx0, dx_ref) < 1e-4
    with pytest.raises(Exception):
        DC.conv2d_dgrad(cot, w, tuple(x.shape), s, p, False, residual=res[:, :4])


def test_passthrough_nodes_add_fan_in_gradients(gpu_device):
    """A tensor with three consumers wired as conv1 -> downsample conv -> extra consumer through passthrough outputs, a head
    with a passthrough, and the pool with a passthrough: same gradients as plain autograd accumulation."""
    from deep_visual_slam_amd import conv as DC, nn_ops
    torch.manual_seed(4)
    B, H, W = 2, 16, 24
    x = torch.randn(B, 64, H, W, device=gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    w1 = (torch.randn(128, 64, 3, 3, device=gpu_device) * 0.05).contiguous(memory_format=CL).requires_grad_(True)
    wd = (torch.randn(128, 64, 1, 1, device=gpu_device) * 0.1).contiguous(memory_format=CL).requires_grad_(True)
    wh = (torch.randn(1, 64, 3, 3, device=gpu_device) * 0.05).contiguous(memory_format=CL).requires_grad_(True)
    bh = torch.zeros(1, device=gpu_device, requires_grad=True)

    def run(passthrough):
        if passthrough:
            y1, xa = DC.conv2d(x, w1, None, 2, 1, passthrough=True)
            yd, xb = DC.conv2d(xa, wd, None, 2, 0, passthrough=True)
            yh, xc = DC.head_conv2d(xb, wh, bh, 0, 1, "sigmoid", passthrough=True)
            yp, xe = nn_ops._MaxPool3x3s2.apply(xc, True)
            extra = xe
        else:
            y1 = DC.conv2d(x, w1, None, 2, 1)
            yd = DC.conv2d(x, wd, None, 2, 0)
            yh = DC.head_conv2d(x, wh, bh, 0, 1, "sigmoid")
            yp = nn_ops._MaxPool3x3s2.apply(x)
            extra = x
        loss = (y1 * c1).sum() + (yd * cd).sum() + (yh * ch).sum() + (yp * cp).sum() + (extra * ce).sum()
        return torch.autograd.grad(loss, [x, w1, wd, wh, bh])

    c1 = torch.randn(B, 128, H // 2, W // 2, device=gpu_device).contiguous(memory_format=CL)
    cd = torch.randn_like(c1)
    ch = torch.randn(B, 1, H, W, device=gpu_device).contiguous(memory_format=CL)
    cp = torch.randn(B, 64, H // 2, W // 2, device=gpu_device).contiguous(memory_format=CL)
    ce = torch.randn_like(x)
    g_ref = run(False)
    g = run(True)
    for a, r, nm in zip(g, g_ref, ("dx", "dw1", "dwd", "dwh", "dbh")):
        assert relmax(a, r) < 2e-5, (nm, relmax(a, r))
    # an unused head output (single-scale training): the passthrough gradient still arrives
    yh, xc = DC.head_conv2d(x, wh, bh, 0, 1, "sigmoid", passthrough=True)
    (gx,) = torch.autograd.grad((xc * ce).sum(), [x])
    assert relmax(gx, ce) < 1e-6
