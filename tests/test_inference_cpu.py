"""Host logic of the inference path that needs no GPU: the BatchNorm fold (values, cache invalidation) and the
disparity-head selection switch."""
import torch
import torch.nn.functional as F


def test_folded_bn_equals_eval_batchnorm():
    from deep_visual_slam_amd import nn_ops
    torch.manual_seed(0)
    w = torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(8).eval()
    bn.running_mean.copy_(torch.randn(8) * 0.3)
    bn.running_var.copy_(torch.rand(8) + 0.5)
    bn.weight.data.copy_(torch.rand(8) + 0.5)
    bn.bias.data.copy_(torch.randn(8) * 0.1)
    x = torch.randn(2, 4, 9, 11)
    ref = bn(F.conv2d(x, w, None, 1, 1))
    w_f, b_f = nn_ops.folded_bn(w, bn)
    assert w_f.is_contiguous(memory_format=torch.channels_last) and not w_f.requires_grad
    assert torch.allclose(F.conv2d(x, w_f, b_f, 1, 1), ref, atol=1e-5, rtol=1e-5)
    # cached until one of the five tensors changes in place
    assert nn_ops.folded_bn(w, bn)[0] is w_f
    bn.running_var.mul_(2.0)
    w_g, b_g = nn_ops.folded_bn(w, bn)
    assert w_g is not w_f
    assert torch.allclose(F.conv2d(x, w_g, b_g, 1, 1), bn(F.conv2d(x, w, None, 1, 1)), atol=1e-5, rtol=1e-5)
    w.mul_(0.5)
    assert nn_ops.folded_bn(w, bn)[0] is not w_g


def test_inference_mode_switch():
    from deep_visual_slam_amd import nn_ops
    bn = torch.nn.BatchNorm2d(4)
    assert not nn_ops.inference_mode(bn)                 # training
    bn.eval()
    assert not nn_ops.inference_mode(bn)                 # eval but differentiable: keep autograd
    with torch.no_grad():
        assert nn_ops.inference_mode(bn)
        assert not nn_ops.inference_mode(torch.nn.BatchNorm2d(4, track_running_stats=False).eval())


def test_depthnet_scale_selection_is_inference_only():
    from deep_visual_slam_amd.depthnet import DepthNet
    dn = DepthNet(18, pretrained=False)
    dn.inference_scales = (0,)
    assert all(dn._wanted(s) for s in range(4))          # training: every head, as the reference
    dn.eval()
    assert all(dn._wanted(s) for s in range(4))          # eval with autograd on: unchanged
    with torch.no_grad():
        assert [dn._wanted(s) for s in range(4)] == [True, False, False, False]
        dn.inference_scales = None
        assert all(dn._wanted(s) for s in range(4))
