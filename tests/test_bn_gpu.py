"""Fused training-mode BatchNorm (+ residual, + ReLU) vs nn.BatchNorm2d + add + relu (forward values,
running statistics, input / residual / affine gradients)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("C,H,W,mode", [(64, 24, 40, "plain"), (128, 12, 20, "residual"), (256, 7, 9, "res_bn"),
                                         (512, 5, 6, "residual"), (64, 17, 23, "norelu")])
def test_conv_bn_act(gpu_device, C, H, W, mode):
    from deep_visual_slam_amd import nn_ops
    torch.manual_seed(0)
    B = 3
    dev = gpu_device
    cin = C // 2 if mode == "res_bn" else C
    x = torch.randn(B, cin, H * (2 if mode == "res_bn" else 1), W * (2 if mode == "res_bn" else 1), device=dev)
    x = x.contiguous(memory_format=CL).requires_grad_(True)
    stride = 2 if mode == "res_bn" else 1
    w = (torch.randn(C, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C).to(dev).train()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.3, 0.3)
    bn_ref = torch.nn.BatchNorm2d(C).to(dev).train(); bn_ref.load_state_dict(bn.state_dict())
    relu = mode != "norelu"
    res_in = res = res_ref = None
    if mode == "residual":
        res_in = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=CL).requires_grad_(True)
    elif mode == "res_bn":
        wd = (torch.randn(C, cin, 1, 1, device=dev) * 0.2).requires_grad_(True)
        bnd = torch.nn.BatchNorm2d(C).to(dev).train(); bnd.weight.data.uniform_(0.5, 1.5)
        bnd_ref = torch.nn.BatchNorm2d(C).to(dev).train(); bnd_ref.load_state_dict(bnd.state_dict())
        res = (wd, bnd, 2)
        res_in = x
    # reference composition
    y = bn_ref(F.conv2d(x, w, None, stride, 1))
    if mode == "residual":
        y = y + res_in
    elif mode == "res_bn":
        y = y + bnd_ref(F.conv2d(x, wd, None, 2, 0))
    z_ref = F.relu(y) if relu else y
    z = nn_ops.conv_bn_act(x, w, bn, stride, 1, relu=relu, residual=res_in, res=res)
    assert rel(z, z_ref) < 5e-5
    assert rel(bn.running_mean, bn_ref.running_mean) < 1e-4 and rel(bn.running_var, bn_ref.running_var) < 1e-4
    assert int(bn.num_batches_tracked) == 1
    cot = torch.randn_like(z_ref)
    ins = [x, w, bn.weight, bn.bias] + ([res_in] if mode == "residual" else []) + ([wd, bnd.weight, bnd.bias] if mode == "res_bn" else [])
    ins_ref = [x, w, bn_ref.weight, bn_ref.bias] + ([res_in] if mode == "residual" else []) + ([wd, bnd_ref.weight, bnd_ref.bias] if mode == "res_bn" else [])
    g = torch.autograd.grad(z, ins, cot)
    g_ref = torch.autograd.grad(z_ref, ins_ref, cot)
    for a, r, i in zip(g, g_ref, range(len(g))):
        assert rel(a, r) < 5e-4, (i, rel(a, r))


@pytest.mark.parametrize("shape", [(2, 64, 48, 64), (1, 16, 7, 9), (3, 8, 10, 5)])
def test_maxpool_matches_torch_including_ties(gpu_device, shape):
    """dvs_maxpool3x3s2_*: values bit-exact, and -- with half the inputs clamped to 0 as after the stem's ReLU --
    gradients routed to the same (first) maximum as torch's kernel."""
    import torch.nn.functional as F
    from deep_visual_slam_amd import nn_ops
    torch.manual_seed(4)
    x = torch.relu(torch.randn(*shape, device=gpu_device)).contiguous(memory_format=torch.channels_last)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y1 = nn_ops.max_pool_3x3_s2(x1)
    y2 = F.max_pool2d(x2, 3, 2, 1)
    assert y1.shape == y2.shape and torch.equal(y1, y2)
    g = torch.randn_like(y2)
    y1.backward(g)
    y2.backward(g)
    # torch scatters with atomics (any order), this gathers in a fixed order: same terms, last-bit differences;
    # a different tie rule would move whole gradient values between pixels
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-6, atol=1e-6)
    assert torch.equal(x1.grad != 0, x2.grad != 0)


def test_relu_mask_recomputed_from_y_equals_the_mask_from_z(gpu_device):
    """Residual-free BatchNorm+ReLU (bn1 of a BasicBlock, the stem): the backward recomputes z > 0 as y*scale + shift > 0 with
    the forward kernel's own expression (dvs_bn_bwd_*_ymask) -- every gradient must be identical, bit for bit, to the path
    that reads the saved z, also with two statistics groups."""
    import torch.nn as nn
    from deep_visual_slam_amd import bn as DB
    CL = torch.channels_last
    torch.manual_seed(4)
    for groups, (B, C, H, W) in ((1, (4, 64, 24, 40)), (2, (6, 128, 12, 20)), (1, (3, 512, 3, 5))):
        y = torch.randn(B, C, H, W, device=gpu_device).contiguous(memory_format=CL)
        cot = torch.randn_like(y)
        gam, bet = torch.rand(C, device=gpu_device) + 0.5, torch.rand(C, device=gpu_device) * 0.6 - 0.3
        out = []
        for ymask in (True, False):
            old, DB._YMASK = DB._YMASK, ymask
            try:
                m = nn.BatchNorm2d(C).to(gpu_device).train()
                with torch.no_grad():
                    m.weight.copy_(gam)
                    m.bias.copy_(bet)
                yy = y.clone().requires_grad_(True)
                st = DB.channel_stats(yy.detach(), groups)
                z = DB.bn_act(yy, m, st, relu=True, groups=groups)
                z.backward(cot)
                out.append((z.detach().clone(), yy.grad.clone(), m.weight.grad.clone(), m.bias.grad.clone()))
            finally:
                DB._YMASK = old
        for a, b in zip(*out):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,C,H,W,groups,need_z", [(4, 64, 24, 32, 1, True), (4, 64, 24, 32, 2, False), (2, 64, 15, 21, 1, True),
                                                     (6, 16, 9, 10, 2, True), (2, 64, 240, 320, 1, False)])
def test_stem_tail_fused_equals_bn_relu_then_maxpool(gpu_device, B, C, H, W, groups, need_z):
    """relu(bn1(y)) -> MaxPool2d(3, 2, 1) in one pass (bn.bn_relu_pool: z not materialised unless asked for, the pool gradient
    gathered inside the BatchNorm backward passes) against the separate operators (bn.bn_act + nn_ops.max_pool_3x3_s2) and,
    through them, torch: outputs, running statistics, dy, d gamma, d beta; with a second consumer of z when need_z."""
    import torch.nn as nn
    from deep_visual_slam_amd import bn as DB, nn_ops
    torch.manual_seed(11)
    y0 = (torch.randn(B, C, H, W, device=gpu_device) * 1.5 + 0.3).contiguous(memory_format=torch.channels_last)
    cot_p = torch.randn(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=gpu_device).contiguous(memory_format=torch.channels_last)
    cot_z = torch.randn_like(y0)

    def stats_of(y):
        parts = y.chunk(groups, 0)
        st = torch.stack([torch.stack([p.sum((0, 2, 3)), (p * p).sum((0, 2, 3))]) for p in parts])
        return st if groups > 1 else st[0]

    def run(fused):
        m = nn.BatchNorm2d(C).to(gpu_device).train()
        with torch.no_grad():
            m.weight.copy_(torch.linspace(0.5, 1.5, C)); m.bias.copy_(torch.linspace(-0.4, 0.4, C))
        y = y0.clone().requires_grad_(True)
        st = stats_of(y.detach())
        if fused:
            z, p = DB.bn_relu_pool(y, m, st, groups=groups, need_z=need_z)
        else:
            z = DB.bn_act(y, m, st, True, groups=groups)
            p = nn_ops.max_pool_3x3_s2(z)
        loss = (p * cot_p).sum()
        if need_z:
            loss = loss + (z * cot_z).sum()
        g = torch.autograd.grad(loss, [y, m.weight, m.bias])
        return p.detach(), (z.detach() if need_z else None), g, m.running_mean.clone(), m.running_var.clone()

    p1, z1, g1, rm1, rv1 = run(True)
    p0, z0, g0, rm0, rv0 = run(False)
    assert rel(p1, p0) < 1e-6          # (scale / shift come from two kernels whose fused multiply-adds may round differently)
    if need_z:
        assert rel(z1, z0) < 1e-6
    assert torch.allclose(rm1, rm0) and torch.allclose(rv1, rv0)
    for a, r, nm in zip(g1, g0, ("dy", "dgamma", "dbeta")):
        err = float((a - r).abs().max() / (r.abs().max() + 1e-30))
        assert err < 2e-5, (nm, err)
