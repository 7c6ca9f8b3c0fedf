"""Fused arena Adam (dvs_adam_step) vs torch.optim.Adam on the same parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch(gpu_device):
    from deep_visual_slam_amd import dp
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).to(gpu_device)
    m2 = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5)).to(gpu_device)
    m2.load_state_dict(m1.state_dict())
    ref = torch.optim.Adam(m1.parameters(), lr=1e-3)
    flat = dp.FlatParams(dp.trainable_parameters(m2))
    opt = dp.FusedAdam(flat, lr=1e-3)
    for step in range(5):
        x = torch.randn(8, 37, device=gpu_device)
        ref.zero_grad()
        m1(x).pow(2).mean().backward()
        ref.step()
        m2(x).pow(2).mean().backward()
        opt.step(grad_scale=1.0, zero_grad=True)
        assert float(flat.grads.abs().max()) == 0.0
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(p1, p2, atol=1e-6, rtol=1e-5)


def test_grad_scale_equals_prescaled_gradient(gpu_device):
    from deep_visual_slam_amd import dp
    torch.manual_seed(1)
    w1 = torch.nn.Parameter(torch.randn(1001, device=gpu_device))
    w2 = torch.nn.Parameter(w1.detach().clone())
    f1, f2 = dp.FlatParams([("w", w1)]), dp.FlatParams([("w", w2)])
    o1, o2 = dp.FusedAdam(f1, lr=1e-2), dp.FusedAdam(f2, lr=1e-2)
    g = torch.randn(1001, device=gpu_device)
    f1.grads[:1001].copy_(g * 0.125)
    f2.grads[:1001].copy_(g)
    o1.step(grad_scale=1.0)
    o2.step(grad_scale=0.125)
    assert torch.allclose(w1, w2, atol=1e-7)


def test_prepacked_dgrad_weights_follow_the_optimiser(gpu_device):
    """dp.FusedAdam repacks the data-gradient operand of every arena convolution after each step (one launch); a pack is
    used only while the weight's version counter is the one it was packed at."""
    import torch.nn.functional as F
    from deep_visual_slam_amd import conv as DC, dp
    torch.manual_seed(2)
    net = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 3, padding=1, bias=False), torch.nn.Conv2d(64, 32, 3, padding=1, bias=False))
    net = net.to(gpu_device).to(memory_format=torch.channels_last)
    flat = dp.FlatParams(dp.trainable_parameters(net))
    opt = dp.FusedAdam(flat, lr=1e-2)
    x = torch.randn(2, 32, 12, 16, device=gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def packed_matches(w):
        ent = DC._prepacked.get(w.data_ptr())
        assert ent is not None and ent[1] == w._version
        co, ci, kh, kw = w.shape
        ref = w.detach().permute(1, 2, 3, 0).contiguous().reshape(-1)       # [Cin][kh][kw][Cout]
        return torch.equal(ent[0], ref)

    for step in range(3):
        w0, w1 = net[0].weight, net[1].weight
        assert packed_matches(w0) and packed_matches(w1)
        y = DC.conv2d(DC.conv2d(x, w0, None, 1, 1), w1, None, 1, 1)
        y_ref = F.conv2d(F.conv2d(x, w0.detach(), None, 1, 1), w1.detach(), None, 1, 1)
        cot = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, [x], cot, retain_graph=True)
        (gx_ref,) = torch.autograd.grad(y_ref, [x], cot)
        assert float((gx - gx_ref).abs().max() / gx_ref.abs().max()) < 1e-4      # the dgrad used the (current) packs
        (y * cot).sum().backward()
        opt.step()
    # an in-place torch update invalidates the packs (version counter): conv2d_dgrad packs on the fly again
    with torch.no_grad():
        net[1].weight.mul_(0.5)
    ent = DC._prepacked[net[1].weight.data_ptr()]
    assert ent[1] != net[1].weight._version
    y = DC.conv2d(x, net[0].weight, None, 1, 1).detach().requires_grad_(True)
    cot = torch.randn(2, 32, 12, 16, device=gpu_device)
    (gy,) = torch.autograd.grad(DC.conv2d(y, net[1].weight, None, 1, 1), [y], cot)
    (gy_ref,) = torch.autograd.grad(F.conv2d(y, net[1].weight.detach(), None, 1, 1), [y], cot)
    assert float((gy - gy_ref).abs().max() / gy_ref.abs().max()) < 1e-4
