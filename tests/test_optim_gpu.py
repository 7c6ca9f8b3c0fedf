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
