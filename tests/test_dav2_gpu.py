"""GPU parity of the Depth-Anything-V2 ViT-S path (a14 / BASELINE.json configs[4]; csrc/vit.hip + the implicit-GEMM engine
through deep_visual_slam_amd.depth_anything_v2) against reference-generated goldens (encoder) and the CPU oracle (whole net).

fp32 tolerances: 12 transformer blocks deep with O(1) activations -> tokens and depth rel-L2 2e-5 (measured 3e-7 .. 8e-7,
profiles/r02_dav2_gpu_tests.log); the
kernels themselves (attention, LayerNorm, resize, deconv scatter) are checked against torch at 2e-5 of the tensor max."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def relmax(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,N,heads", [(1, 1370, 6), (2, 49, 6), (3, 33, 2), (1, 129, 12)])
def test_attention_kernel(gpu_device, B, N, heads):
    """softmax(q k^T / 8) v for d = 64 (attention.py:49-62): N = 1370 (518x518), ragged key / query tiles, few tokens."""
    from deep_visual_slam_amd.depth_anything_v2 import attention
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn(B, N, 3, heads, 64, generator=g) * 1.5
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    ref = ((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B * N, heads * 64)
    out = attention(qkv.reshape(B * N, -1).to(gpu_device), B, N, heads, 64)
    assert out.shape == ref.shape and relmax(out, ref) < 2e-5


@pytest.mark.parametrize("M,C", [(1370, 384), (7, 768), (300, 1024), (5, 64)])
def test_layernorm_kernel(gpu_device, M, C):
    from deep_visual_slam_amd.depth_anything_v2 import layernorm
    g = torch.Generator().manual_seed(C)
    x, w, b = torch.randn(M, C, generator=g) * 2 + 0.5, torch.randn(C, generator=g), torch.randn(C, generator=g)
    out = layernorm(x.to(gpu_device), w.to(gpu_device), b.to(gpu_device), 1e-6)
    assert relmax(out, F.layer_norm(x, (C,), w, b, 1e-6)) < 2e-5


def test_gemm_epilogues(gpu_device):
    """Token GEMMs on the implicit-GEMM engine: bias, GELU (erf form), residual; K = 384 / 1536 / padded 608."""
    from deep_visual_slam_amd.depth_anything_v2 import gemm
    g = torch.Generator().manual_seed(0)
    for M, K, N, act, res in ((1370, 384, 1152, None, False), (1370, 384, 1536, "gelu", False), (2740, 1536, 384, None, True),
                              (1369, 608, 384, None, False), (50, 384, 384, None, True)):
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g) * 0.1
        r = torch.randn(M, N, generator=g) if res else None
        ref = F.linear(x, w, b)
        ref = F.gelu(ref) if act == "gelu" else ref
        ref = ref + r if res else ref
        w4 = w.view(N, K, 1, 1).contiguous(memory_format=torch.channels_last).to(gpu_device)
        out = gemm(x.to(gpu_device), w4, b.to(gpu_device), act=act, residual=r.to(gpu_device) if res else None)
        assert out.shape == ref.shape and relmax(out, ref) < 2e-5, (M, K, N, act, res, relmax(out, ref))


def test_resize_and_deconv_kernels(gpu_device):
    from deep_visual_slam_amd.depth_anything_v2 import conv_transpose_s, resize_bilinear_ac
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 19, 19, generator=g)
    for size in ((37, 37), (38, 38), (25, 60), (19, 19)):
        out = resize_bilinear_ac(x.to(gpu_device).contiguous(memory_format=torch.channels_last), *size)
        assert relmax(out, F.interpolate(x, size, mode="bilinear", align_corners=True)) < 2e-6
    for k, ci, co in ((4, 48, 48), (2, 96, 96)):
        xin, w, b = torch.randn(2, ci, 7, 9, generator=g), torch.randn(ci, co, k, k, generator=g) * 0.1, torch.randn(co, generator=g)
        ref = F.conv_transpose2d(xin, w, b, stride=k)
        wg = w.permute(2, 3, 1, 0).reshape(k * k * co, ci, 1, 1).contiguous(memory_format=torch.channels_last).to(gpu_device)
        out = conv_transpose_s(xin.to(gpu_device).contiguous(memory_format=torch.channels_last), wg, b.repeat(k * k).to(gpu_device), k, co)
        assert out.shape == ref.shape and relmax(out, ref) < 2e-5


@pytest.fixture(scope="module")
def dav2(gpu_device):
    from deep_visual_slam_amd.depth_anything_v2 import DepthAnythingV2
    from oracle.depth_anything import seeded_weights
    net = DepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384]).eval()
    enc = seeded_weights(net.pretrained.state_dict(), seed=0)
    head = seeded_weights(net.depth_head.state_dict(), seed=4)
    net.pretrained.load_state_dict(enc)
    net.depth_head.load_state_dict(head)
    sd = {"pretrained." + k: v for k, v in enc.items()}
    sd.update({"depth_head." + k: v for k, v in head.items()})
    assert sorted(sd) == sorted(net.state_dict())
    return net.to(gpu_device), sd


def test_dinov2_encoder_vs_reference_goldens(gpu_device, dav2):
    """get_intermediate_layers(x, [2,5,8,11], return_class_token=True) against the reference's own outputs: the small
    non-square image in full, 518x518 (N = 1370, the BASELINE size) by strided sample + checksums."""
    net, _ = dav2
    rec = load_golden("dav2_dinov2_vits.npz")
    with torch.no_grad():
        outs = net.pretrained.get_intermediate_layers(torch.from_numpy(rec["small/x"]).to(gpu_device), [2, 5, 8, 11], return_class_token=True)
    for i, (tok, cls) in enumerate(outs):
        e = rel(tok, rec["small/tok%d" % i])
        print("dinov2 84x112 tap %d rel-L2 %.2e" % (i, e))
        assert tok.shape == rec["small/tok%d" % i].shape and e < 2e-5 and rel(cls, rec["small/cls%d" % i]) < 2e-5
    g = torch.Generator().manual_seed(1)
    torch.randn(2, 3, 84, 112, generator=g)
    x = torch.randn(1, 3, 518, 518, generator=g)
    with torch.no_grad():
        outs = net.pretrained.get_intermediate_layers(x.to(gpu_device), [2, 5, 8, 11], return_class_token=True)
    for i, (tok, cls) in enumerate(outs):
        assert tok.shape == (1, 1369, 384) and cls.shape == (1, 384)
        e = rel(tok[0, ::37, ::7], rec["full/tok%d#sample" % i])
        print("dinov2 518x518 tap %d sample rel-L2 %.2e" % (i, e))
        assert e < 2e-5 and rel(cls, rec["full/cls%d" % i]) < 2e-5
        t = tok.double().cpu().numpy()
        assert abs((t * t).sum() - rec["full/tok%d#sq" % i]) < 2e-4 * rec["full/tok%d#sq" % i]


@pytest.mark.parametrize("B,H,W", [(1, 518, 518), (2, 84, 112)])
def test_depth_anything_forward_vs_oracle(gpu_device, dav2, B, H, W):
    """DepthAnythingV2.forward (dpt.py:192-199) -> depth [B,H,W] against the CPU oracle; the disp adapter's contract."""
    from oracle import depth_anything as OD
    net, sd = dav2
    g = torch.Generator().manual_seed(H)
    x = torch.randn(B, 3, H, W, generator=g)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = OD.depth_anything_v2(x, sd)
        out = net(x.to(gpu_device))
        disp = net.disp_outputs(x.to(gpu_device), scales=(0, 1))
    assert out.shape == (B, H, W)
    e = rel(out, ref)
    print("depth_anything_v2 %dx%d rel-L2 %.2e max-rel %.2e" % (H, W, e, relmax(out, ref)))
    assert e < 2e-5                               # measured 3e-7 .. 5e-7
    assert disp[("disp", 0)].shape == (B, 1, H, W) and disp[("disp", 1)].shape == (B, 1, H // 2, W // 2)
    assert float(disp[("disp", 0)].min()) >= 0.0 and float(disp[("disp", 0)].max()) <= 1.0


def test_checkpoint_keys(gpu_device, dav2):
    net, sd = dav2
    keys = set(net.state_dict())
    for k in ("pretrained.blocks.11.ls2.gamma", "pretrained.pos_embed", "pretrained.mask_token", "depth_head.projects.3.bias",
              "depth_head.resize_layers.0.weight", "depth_head.scratch.refinenet4.resConfUnit1.conv1.weight",
              "depth_head.scratch.output_conv2.2.weight", "depth_head.scratch.layer1_rn.weight"):
        assert k in keys, k


# ------------------------------------------------------------------------------------------------ training path
@pytest.mark.parametrize("B,N,heads", [(1, 1370, 6), (2, 49, 6), (2, 100, 2)])
def test_attention_backward_kernels(gpu_device, B, N, heads):
    """dvs_attention_bwd (dQ kernel + dK/dV kernel, both recomputing the scores in the forward's register layout) against
    torch autograd of softmax(q k^T / 8) v."""
    from deep_visual_slam_amd.depth_anything_v2 import _AttentionF
    g = torch.Generator().manual_seed(N + 1)
    qkv = (torch.randn(B, N, 3, heads, 64, generator=g) * 1.2).requires_grad_(True)
    cot = torch.randn(B * N, heads * 64, generator=g)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    ref = (((q * 0.125) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(B * N, heads * 64)
    (ref * cot).sum().backward()
    x = qkv.detach().reshape(B * N, -1).to(gpu_device).requires_grad_(True)
    out = _AttentionF.apply(x, B, N, heads, 64)
    assert relmax(out, ref) < 2e-5
    (out * cot.to(gpu_device)).sum().backward()
    got, want = x.grad.reshape(B, N, 3, heads, 64).cpu(), qkv.grad
    for i, name in enumerate(("dq", "dk", "dv")):
        e = relmax(got[:, :, i], want[:, :, i])
        assert e < 5e-5, (name, e)


def test_layernorm_act_resize_shuffle_backward(gpu_device):
    from deep_visual_slam_amd.depth_anything_v2 import _ActF, _LayerNormF, _ResizeF, _ShuffleF
    g = torch.Generator().manual_seed(5)
    CL = torch.channels_last
    # LayerNorm
    x, w, b = (torch.randn(700, 384, generator=g) * 2 + 0.3).requires_grad_(True), torch.randn(384, generator=g).requires_grad_(True), torch.randn(384, generator=g).requires_grad_(True)
    cot = torch.randn(700, 384, generator=g)
    (F.layer_norm(x, (384,), w, b, 1e-6) * cot).sum().backward()
    xg, wg, bg = (t.detach().to(gpu_device).requires_grad_(True) for t in (x, w, b))
    (_LayerNormF.apply(xg, wg, bg, 1e-6) * cot.to(gpu_device)).sum().backward()
    assert relmax(xg.grad, x.grad) < 5e-5 and relmax(wg.grad, w.grad) < 5e-5 and relmax(bg.grad, b.grad) < 5e-5
    # GELU / ReLU
    for act, fn in (("gelu", F.gelu), ("relu", F.relu)):
        x = torch.randn(33, 1536, generator=g, requires_grad=True)
        cot = torch.randn(33, 1536, generator=g)
        (fn(x) * cot).sum().backward()
        xg = x.detach().to(gpu_device).requires_grad_(True)
        y = _ActF.apply(xg, act)
        assert relmax(y, fn(x.detach())) < 2e-6
        (y * cot.to(gpu_device)).sum().backward()
        assert relmax(xg.grad, x.grad) < 5e-6, act
    # bilinear resize (align_corners=True) and the deconv scatter
    x = torch.randn(2, 32, 9, 12, generator=g, requires_grad=True)
    cot = torch.randn(2, 32, 19, 25, generator=g)
    (F.interpolate(x, (19, 25), mode="bilinear", align_corners=True) * cot).sum().backward()
    xg = x.detach().to(gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    (_ResizeF.apply(xg, 19, 25) * cot.to(gpu_device)).sum().backward()
    assert relmax(xg.grad, x.grad) < 2e-5
    gsrc = torch.randn(2, 4 * 8, 5, 6, generator=g)
    gg = gsrc.to(gpu_device).contiguous(memory_format=CL).requires_grad_(True)
    y = _ShuffleF.apply(gg, 2, 8)
    cot = torch.randn(2, 8, 10, 12, generator=g).to(gpu_device)
    (y * cot).sum().backward()
    # the scatter is a permutation: its backward is the inverse gather of the cotangent
    ref = cot.cpu().reshape(2, 8, 5, 2, 6, 2).permute(0, 3, 5, 1, 2, 4).reshape(2, 32, 5, 6)
    assert torch.equal(gg.grad.cpu(), ref)


def test_depth_anything_gradients_vs_oracle(gpu_device):
    """Whole DepthAnythingV2 (encoder + DPT head) with autograd on: depth and every parameter gradient against the CPU
    oracle's autograd, 84x112, batch 2 (bicubic pos-embed path included)."""
    from deep_visual_slam_amd.depth_anything_v2 import DepthAnythingV2
    from oracle import depth_anything as OD
    from oracle.depth_anything import seeded_weights
    net = DepthAnythingV2(encoder="vits", features=64, out_channels=[48, 96, 192, 384])
    enc, head = seeded_weights(net.pretrained.state_dict(), seed=0), seeded_weights(net.depth_head.state_dict(), seed=4)
    net.pretrained.load_state_dict(enc)
    net.depth_head.load_state_dict(head)
    net = net.to(gpu_device).train()
    sd = {"pretrained." + k: v.clone().requires_grad_(True) for k, v in enc.items()}
    sd.update({"depth_head." + k: v.clone().requires_grad_(True) for k, v in head.items()})
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 3, 84, 112, generator=g)
    cot = torch.randn(2, 84, 112, generator=g) / 50.0
    torch.set_num_threads(16)
    ref = OD.depth_anything_v2(x, sd)
    (ref * cot).sum().backward()
    out = net(x.to(gpu_device))
    assert out.requires_grad and rel(out, ref) < 2e-5
    (out * cot.to(gpu_device)).sum().backward()
    torch.cuda.synchronize()
    rows = []
    for n, p in net.named_parameters():
        rg = sd[n].grad
        if rg is None or float(rg.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n       # mask_token: unused on both sides
            continue
        assert p.grad is not None, n
        rows.append((rel(p.grad, rg), n))
    rows.sort(reverse=True)
    print("DA-V2 gradients vs oracle: worst", rows[:3], "median %.2e over %d tensors" % (rows[len(rows) // 2][0], len(rows)))
    # ReLU branches in the DPT head only (8 ResidualConvUnits + output conv): smooth elsewhere (GELU, softmax, LayerNorm)
    assert rows[0][0] < 2e-3 and rows[len(rows) // 2][0] < 1e-4, rows[:5]


def test_encoder_swap_trains_in_the_vo_trainer(gpu_device):
    """DepthAnythingDispNet in place of DepthNet inside MonodepthTrainer (BASELINE configs[4]: "encoder swap"): the adapter's
    disparity pyramid against the oracle composition, then one full VO training step (fused loss chain + backward through the
    DPT head, the 12 transformer blocks and PoseNet) with finite gradients on every trained tensor and a loss that goes down
    over a few Adam steps."""
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.depth_anything_v2 import DepthAnythingDispNet
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    from oracle import depth_anything as OD
    B, H, W = 2, 96, 128
    torch.manual_seed(0)
    dn = DepthAnythingDispNet().to(gpu_device).train()
    pn = PoseNet(18, pretrained=False, num_input_images=2).to(gpu_device).train()
    sample = synth.parity_sample(B, H, W)
    x = sample[("target_image", 0)]
    # adapter vs oracle: resize to 84x126, ImageNet normalisation, DA-V2, sigmoid map resized to the pyramid
    sd = {k: v.detach().cpu() for k, v in dn.net.state_dict().items()}
    with torch.no_grad():
        xin = F.interpolate(x, (84, 126), mode="bilinear", align_corners=True)
        xin = (xin - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        d_ref = (OD.depth_anything_v2(xin, sd) / 20.0).unsqueeze(1)
        out = dn(x.to(gpu_device))
    for s in range(4):
        ref = F.interpolate(d_ref, (H >> s, W >> s), mode="bilinear", align_corners=True)
        assert out[("disp", s)].shape == ref.shape and rel(out[("disp", s)], ref) < 2e-5, s
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                         ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
    tr = MonodepthTrainer(dn, pn, cfg, gpu_device)
    opt = torch.optim.Adam(list(dn.parameters()) + list(pn.parameters()), lr=1e-4)
    seen = []
    for it in range(4):
        opt.zero_grad(set_to_none=True)
        outputs, losses = tr.process_batch(dict(sample))
        losses["loss"].backward()
        if it == 0:
            missing = [n for n, p in dn.named_parameters() if p.grad is None and "mask_token" not in n]
            assert not missing, missing[:5]
            assert all(torch.isfinite(p.grad).all() for p in dn.parameters() if p.grad is not None)
            assert float(dict(dn.named_parameters())["net.pretrained.blocks.0.attn.qkv.weight"].grad.abs().max()) > 0
        opt.step()
        seen.append(float(losses["loss"]))
    assert all(np.isfinite(seen)) and seen[-1] < seen[0], seen
