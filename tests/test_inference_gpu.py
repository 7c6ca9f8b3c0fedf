"""Inference path (eval() + no_grad: BatchNorm folded into the convolutions, identity + ReLU in the conv epilogue,
optional HIP-graph replay) against the CPU oracle's eval-mode networks.  SURVEY.md section 8(f) rank 2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _randomise_bn(net, seed):
    """Non-trivial running statistics and affine parameters, as a trained checkpoint has."""
    g = torch.Generator().manual_seed(seed)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 1.5 + 0.25)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


@pytest.fixture(scope="module")
def eval_nets(gpu_device):
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.posenet_single import PoseNet
    torch.manual_seed(0)
    dn, pn = DepthNet(18, pretrained=False), PoseNet(18, pretrained=False, num_input_images=2)
    _randomise_bn(dn, 1)
    _randomise_bn(pn, 2)
    sd_d = {k: v.clone() for k, v in dn.state_dict().items()}
    sd_p = {k: v.clone() for k, v in pn.state_dict().items()}
    return dn.to(gpu_device).eval(), pn.to(gpu_device).eval(), sd_d, sd_p


def test_eval_networks_match_oracle(gpu_device, eval_nets):
    from oracle import networks as ON
    dn, pn, sd_d, sd_p = eval_nets
    torch.manual_seed(5)
    x = torch.rand(1, 3, 96, 128)
    x6 = torch.rand(1, 6, 96, 128)
    ref = ON.depthnet(x, sd_d, train=False)
    aa_r, t_r = ON.posenet(x6, sd_p, train=False)
    with torch.no_grad():
        out = dn(x.to(gpu_device))
        aa, t = pn(x6.to(gpu_device))
    for s in range(4):
        assert out[("disp", s)].shape == ref[("disp", s)].shape
        assert rel(out[("disp", s)], ref[("disp", s)]) < 2e-4
    assert rel(aa, aa_r) < 2e-4 and rel(t, t_r) < 2e-4
    # buffers are untouched in eval mode
    assert int(dn.state_dict()["encoder.encoder.bn1.num_batches_tracked"]) == 0
    assert torch.equal(dn.state_dict()["encoder.encoder.bn1.running_mean"].cpu(), sd_d["encoder.encoder.bn1.running_mean"])


def test_fold_follows_weight_updates(gpu_device, eval_nets):
    """The cached fold is refreshed when a weight or a running statistic changes in place (load_state_dict, optimiser)."""
    from oracle import networks as ON
    dn, _, sd_d, _ = eval_nets
    x = torch.rand(1, 3, 64, 96)
    with torch.no_grad():
        before = dn(x.to(gpu_device))[("disp", 0)].clone()
        sd2 = {k: v.clone() for k, v in sd_d.items()}
        sd2["encoder.encoder.layer1.0.bn1.running_var"] *= 3.0
        sd2["encoder.encoder.layer2.0.conv1.weight"] *= 0.5
        dn.load_state_dict(sd2)
        after = dn(x.to(gpu_device))[("disp", 0)]
    ref = ON.depthnet(x, sd2, train=False)[("disp", 0)]
    assert rel(after, ref) < 2e-4 and rel(before, ref) > 1e-3
    dn.load_state_dict(sd_d)


def test_graph_replay_and_scale_selection(gpu_device, eval_nets):
    from deep_visual_slam_amd import inference
    dn, pn, _, _ = eval_nets
    inference.prepare(dn, pn, scales=(0,))
    try:
        torch.manual_seed(6)
        x = torch.rand(1, 3, 96, 128, device=gpu_device)
        x6 = torch.rand(1, 6, 96, 128, device=gpu_device)
        with torch.no_grad():
            eager = dn(x)
            assert list(eager) == [("disp", 0)]
            d0 = eager[("disp", 0)].clone()
            aa0, t0 = [v.clone() for v in pn(x6)]
        gd, gp = inference.Graphed(dn, torch.zeros_like(x)), inference.Graphed(pn, torch.zeros_like(x6))
        for _ in range(2):                       # a replay depends on the copied-in input only (the graphs were captured
            out = gd(x)                          # on zeros); split-K layers sum with atomics, so not bit-for-bit
            aa, t = gp(x6)
            assert rel(out[("disp", 0)], d0) < 1e-5
            assert rel(aa, aa0) < 1e-5 and rel(t, t0) < 1e-5
        with pytest.raises(Exception):
            gd(torch.zeros(1, 3, 64, 64, device=gpu_device))
    finally:
        dn.inference_scales = None


def test_eval_with_grad_keeps_autograd(gpu_device, eval_nets):
    """eval() without no_grad (frozen-BatchNorm fine-tuning) stays differentiable: the fold is an inference-only path."""
    dn, _, _, _ = eval_nets
    x = torch.rand(1, 3, 64, 96, device=gpu_device)
    out = dn(x)[("disp", 0)]
    assert out.requires_grad
    out.mean().backward()
    assert dn.encoder.encoder.layer1[0].conv1.weight.grad is not None
    dn.zero_grad()


def test_frame_predictor_two_streams(gpu_device, eval_nets):
    """PoseNet and DepthNet side by side (eager and as one graph with a fork / join) equal the sequential calls."""
    from deep_visual_slam_amd import inference
    from deep_visual_slam_amd.layers import disp_to_depth, transformation_from_parameters
    dn, pn, _, _ = eval_nets
    inference.prepare(dn, pn, scales=(0,))
    try:
        torch.manual_seed(8)
        x = torch.rand(1, 3, 96, 128, device=gpu_device)
        x6 = torch.rand(1, 6, 96, 128, device=gpu_device)
        with torch.no_grad():
            aa, t = pn(x6)
            T0 = transformation_from_parameters(aa[:, 0], t[:, 0], invert=False).clone()
            d0 = disp_to_depth(dn(x)[("disp", 0)], 0.1, 10.0)[1].clone()
        for graph in (False, True):
            fp = inference.FramePredictor(dn, pn, torch.zeros_like(x), torch.zeros_like(x6), graph=graph)
            for _ in range(2):
                T, depth, disp = fp(x, x6)
                torch.cuda.synchronize()
                assert rel(T, T0) < 1e-5 and rel(depth, d0) < 1e-5 and disp.shape == (1, 1, 96, 128)
    finally:
        dn.inference_scales = None
