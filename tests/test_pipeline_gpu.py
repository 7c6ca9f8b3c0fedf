"""GPU parity of the input side (csrc/preprocess.hip, input_pipeline.py) and of the MonoVO network
adapter (8(f) rank 4: slam_network.Networks) against the CPU oracle restatements."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_u8_to_f32_planar_is_to_tensor(gpu_device):
    """transforms.ToTensor (vo/dataset/common.py:77): exact -- a byte times 1/255 in fp32."""
    from deep_visual_slam_amd.input_pipeline import u8_to_f32_planar
    from oracle import input_pipeline as OI
    rng = np.random.default_rng(0)
    for n, h, w in ((3, 480, 640), (1, 6, 10), (5, 33, 28)):
        u8 = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
        ref = OI.to_tensor(u8)
        got = u8_to_f32_planar(torch.from_numpy(u8).to(gpu_device))
        assert got.shape == ref.shape
        assert float((got.cpu() - ref).abs().max()) <= 6e-8          # x * (1/255) vs x / 255: at most one ulp
        bgr = u8_to_f32_planar(torch.from_numpy(u8).to(gpu_device), bgr=True)
        assert torch.equal(bgr.cpu(), got.cpu().flip(1))


def test_pil_bilinear_resize_is_bit_exact(gpu_device):
    """img.resize((W, H), Image.BILINEAR) of the reference's loader (vo/dataset/common.py:38-44) on the GPU, against outputs
    of Pillow itself (tests/golden/pil_resize_bilinear.npz): shrinking (antialiased), enlarging, one axis only -- every byte."""
    from conftest import load_golden
    from deep_visual_slam_amd.input_pipeline import make_sample, resize_u8
    rec = load_golden("pil_resize_bilinear.npz")
    i = 0
    while "in%d" % i in rec:
        a, b = rec["in%d" % i], rec["out%d" % i]
        x = torch.from_numpy(np.stack([a, a[::-1].copy()])).to(gpu_device)          # a batch of two
        got = resize_u8(x, b.shape[0], b.shape[1]).cpu().numpy()
        assert got.shape == (2,) + b.shape and np.array_equal(got[0], b), i
        i += 1
    assert i == 6
    # through make_sample: frames arrive at another size than the training resolution
    a, b = rec["in4"], rec["out4"]                                                   # 120x160 -> 48x64
    frames = torch.from_numpy(np.stack([a, a, a])[None]).to(gpu_device)
    K = np.eye(4, dtype=np.float32)[None]
    sample = make_sample(frames, K, image_size=(48, 64))
    ref = torch.from_numpy(b).permute(2, 0, 1).float() / 255.0
    assert sample[("target_image", 0)].shape == (1, 3, 48, 64)
    assert float((sample[("target_image", 0)][0].cpu() - ref).abs().max()) <= 6e-8


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_color_jitter_matches_oracle(gpu_device, seed):
    """ColorJitter(0.3, 0.3, 0.3, 0.2) on the three frames of each sample (common.py:31-37,79-81): every permutation
    seen over the seeds; un-jittered samples stay bit-identical."""
    from deep_visual_slam_amd.input_pipeline import JitterParams, color_jitter_
    from oracle import input_pipeline as OI
    rng = np.random.default_rng(seed)
    B, F, H, W = 4, 3, 40, 56
    img = torch.rand(B * F, 3, H, W, generator=torch.Generator().manual_seed(seed))
    img[0, :, :4, :4] = 0.5                       # gray pixels: the maxc == minc branch of the hue conversion
    img[1, :, 0, 0] = torch.tensor([1.0, 0.0, 0.0])
    jp = JitterParams(B, rng)
    jp.apply[:] = [True, True, False, True]
    got = color_jitter_(img.clone().to(gpu_device), jp.records(F)).cpu()
    for b in range(B):
        for f in range(F):
            i = b * F + f
            ref = OI.color_jitter(img[i], jp.order[b], jp.factor[b]) if jp.apply[b] else img[i]
            if not jp.apply[b]:
                assert torch.equal(got[i], ref)
                continue
            err = (got[i] - ref).abs()
            # hue: floor(h * 6) sits on a discontinuity for a handful of pixels (h within rounding of k/6); everywhere
            # else fp32 agreement
            assert float((err > 2e-5).float().mean()) < 2e-3, (b, f, float(err.max()))
            assert float(err.median()) < 1e-6


def test_prefetcher_yields_reference_schema(gpu_device):
    from deep_visual_slam_amd.input_pipeline import Prefetcher, intrinsics_pyramid
    from deep_visual_slam_amd import synth
    from oracle import input_pipeline as OI
    rng = np.random.default_rng(5)
    B, H, W = 2, 48, 64
    K0 = synth.intrinsics(B, H, W)[("K", 0)].numpy()
    batches = [{"frames": rng.integers(0, 256, size=(B, 3, H, W, 3), dtype=np.uint8), "K": K0} for _ in range(3)]
    out = list(Prefetcher(batches, gpu_device, augment=False))
    assert len(out) == 3
    ref_int = synth.intrinsics(B, H, W)
    for batch, sample in zip(batches, out):
        ref = OI.to_tensor(batch["frames"].reshape(B * 3, H, W, 3)).view(B, 3, 3, H, W)
        for j, key in enumerate((("source_left", 0), ("target_image", 0), ("source_right", 0))):
            assert sample[key].shape == (B, 3, H, W) and sample[key].is_cuda
            assert float((sample[key].cpu() - ref[:, j]).abs().max()) <= 6e-8
        for s in range(4):
            assert torch.allclose(sample[("K", s)].cpu(), ref_int[("K", s)], atol=1e-6)
            assert torch.allclose(sample[("inv_K", s)].cpu(), ref_int[("inv_K", s)], atol=1e-7)
    # with augmentation: same shapes, values in [0, 1], reproducible for a fixed seed
    a = list(Prefetcher(batches, gpu_device, augment=True, seed=3))
    b = list(Prefetcher(batches, gpu_device, augment=True, seed=3))
    for x, y in zip(a, b):
        t = x[("target_image", 0)]
        assert float(t.min()) >= 0.0 and float(t.max()) <= 1.0
        assert torch.equal(t, y[("target_image", 0)])


def test_prefetched_sample_trains(gpu_device):
    """A prefetched sample goes straight into MonodepthTrainer.process_batch (the consumer side of the pipeline)."""
    from deep_visual_slam_amd import synth
    from deep_visual_slam_amd.depthnet import DepthNet
    from deep_visual_slam_amd.input_pipeline import Prefetcher
    from deep_visual_slam_amd.learner_new import MonodepthTrainer
    from deep_visual_slam_amd.posenet_single import PoseNet
    rng = np.random.default_rng(6)
    B, H, W = 2, 96, 128
    tex = (synth.parity_sample(B, H, W)[("target_image", 0)].permute(0, 2, 3, 1).numpy() * 255).astype(np.uint8)
    frames = np.stack([np.roll(tex, 2, 2), tex, np.roll(tex, -2, 2)], 1)
    K0 = synth.intrinsics(B, H, W)[("K", 0)].numpy()
    torch.manual_seed(0)
    dn = DepthNet(18, pretrained=False).to(gpu_device).train()
    pn = PoseNet(18, pretrained=False, num_input_images=2).to(gpu_device).train()
    cfg = {"Train": dict(num_source=1, batch_size=B, img_h=H, img_w=W, smoothness_ratio=0.001, auto_mask=True,
                         ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
    tr = MonodepthTrainer(dn, pn, cfg, gpu_device)
    for sample in Prefetcher([{"frames": frames, "K": K0}] * 2, gpu_device, augment=True, seed=1):
        _, losses = tr.process_batch(sample)
        losses["loss"].backward()
        assert torch.isfinite(losses["loss"])


def test_monovo_networks_adapter(gpu_device):
    """slam/network.py contract: depth(frame) -> [H,W] numpy in [0.1, 10]; pose(img1, img2, depth) -> 4x4 numpy, against
    the oracle's eval-mode networks on the same (BGR uint8) frames."""
    from deep_visual_slam_amd.slam_network import Networks
    from oracle import input_pipeline as OI, loss_chain as OL, networks as ON
    H, W = 96, 128
    torch.manual_seed(4)
    nn_ = Networks(image_shape=(H, W), device=gpu_device)
    rng = np.random.default_rng(2)
    f1 = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    f2 = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    sd_d = {k: v.detach().cpu() for k, v in nn_.depth_net.state_dict().items()}
    sd_p = {k: v.detach().cpu() for k, v in nn_.pose_net.state_dict().items()}
    x1, x2 = (OI.to_tensor(f[None, :, :, ::-1].copy()) for f in (f1, f2))      # BGR -> RGB
    with torch.no_grad():
        disp = ON.depthnet(x1, sd_d, train=False)[("disp", 0)]
        depth_ref = OL.disp_to_depth(disp, 0.1, 10.0)[1][0, 0].clamp(0.1, 10.0).numpy()
        aa, t = ON.posenet(torch.cat([x1, x2], 1), sd_p, train=False)
        T_ref = OL.transformation_from_parameters(aa[:, 0], t[:, 0], invert=True)[0].numpy()
    for rep in range(2):                                   # second call replays the captured graphs
        depth = nn_.depth(f1)
        T = nn_.pose(f1, f2, depth=depth)
        assert depth.shape == (H, W) and depth.dtype == np.float32 and depth.min() >= 0.1 and depth.max() <= 10.0
        assert np.abs(depth - depth_ref).max() <= 2e-4 * np.abs(depth_ref).max()
        assert T.shape == (4, 4) and np.abs(T - T_ref).max() < 1e-5
