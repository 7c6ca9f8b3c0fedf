"""GPU parity of the fused loss chain (C-ABI dvs_chain_fwd/bwd, dvs_pose_to_mat_*) against
(1) golden vectors produced by the reference itself and (2) the CPU oracle on other seeded inputs.

Tolerances (fp32): loss scalars rel 1e-5; colours abs 2e-5 (a 1-ulp change of a pixel coordinate
~300 moves the bilinear sample by ~3e-5 * image gradient); gradients rel 2e-3 on all but <=0.2% of
the elements (min/argmin, the SSIM clamp and floor() are discontinuous: rounding flips a few pixels).
"""
import numpy as np
import pytest
import torch

from conftest import golden_chain_inputs, load_golden

pytestmark = pytest.mark.gpu


def close(a, b, atol, rtol):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def close_frac(a, b, atol, rtol, frac=2e-3, l2=2e-3):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    bad = np.abs(a - b) > atol + rtol * np.abs(b)
    # one flipped full-resolution pixel touches 4 elements of a low-resolution gradient map
    assert bad.sum() <= max(frac * bad.size, 8), "mismatch fraction %.2e (%d elements)" % (bad.mean(), bad.sum())
    num = np.linalg.norm(np.where(bad, 0, a - b).astype(np.float64))
    den = np.linalg.norm(b.astype(np.float64)) + 1e-30
    assert num / den < l2, "rel L2 %.3e" % (num / den)


def close_pose(a, b, npix):
    """Pose gradients sum over every pixel, argmin/clamp-flipped ones included.  One flipped pixel moves a
    component by up to ~|grad I| * f / (B*H*W) ~ 10 / npix, so: 1% of the component + 1% of the largest
    component + the weight of one flipped pixel."""
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    close(a, b, atol=max(1e-4, 1e-2 * float(np.abs(b).max()), 10.0 / npix), rtol=1e-2)


def run_chain(dev, sample, disps, poses, noise, ns, materialize=True, auto_mask=True):
    from deep_visual_slam_amd import ops
    d_disps = [d.to(dev).requires_grad_(True) for d in disps[:ns]]
    d_poses = [p.to(dev).requires_grad_(True) for p in poses]
    T_l = ops.pose_to_mat(d_poses[0][:, 0], d_poses[1][:, 0], invert=True)
    T_r = ops.pose_to_mat(d_poses[2][:, 0], d_poses[3][:, 0], invert=False)
    nz = torch.stack(noise).to(dev) if noise is not None else None
    losses, sel, extras = ops.loss_chain(
        sample[("target_image", 0)].to(dev), sample[("source_left", 0)].to(dev),
        sample[("source_right", 0)].to(dev), sample[("K", 0)].to(dev), sample[("inv_K", 0)].to(dev),
        T_l, T_r, d_disps, noise=nz, materialize=materialize, auto_mask=auto_mask)
    total = losses.mean()
    total.backward()
    torch.cuda.synchronize()
    return dict(losses=losses, total=total, sel=sel, extras=extras, T=(T_l, T_r),
                d_disp=[d.grad for d in d_disps], d_pose=[p.grad for p in d_poses])


@pytest.mark.parametrize("name", ["chain_b2_48x64.npz", "chain_b2_96x128.npz", "chain_b1_48x64_s1.npz", "chain_b2_48x64_oob.npz"])
def test_chain_vs_reference_golden(gpu_device, name):
    rec = load_golden(name)
    sample, disps, poses, noise, ns = golden_chain_inputs(rec)
    if name.endswith("_oob.npz"):
        # the out-of-image case: >= 10 % of the samples are clamped at each of the four borders (border padding and the
        # zero coordinate gradient of F.grid_sample, vo/learner_new.py:165-170, through the FUSED chain)
        cm, cp = rec["meta/clamped_m1"], rec["meta/clamped_p1"]
        assert cm[0] >= 0.1 and cm[2] >= 0.1 and cp[1] >= 0.1 and cp[3] >= 0.1
    out = run_chain(gpu_device, sample, disps, poses, noise, ns)
    close(out["total"], rec["loss"], 1e-7, 1e-5)
    close(out["T"][0], rec["out/T_m1"], 2e-6, 2e-5)
    close(out["T"][1], rec["out/T_p1"], 2e-6, 2e-5)
    for s in range(ns):
        close(out["losses"][s], rec["loss/%d" % s], 1e-7, 1e-5)
        sel = ((out["sel"].cpu().numpy() >> (2 * s)) & 3) > 1
        assert (sel[:, None] != rec["out/identity_selection%d" % s].astype(bool)).mean() < 1e-3
        close_frac(out["d_disp"][s], rec["grad/disp%d" % s], atol=2e-8, rtol=2e-3)
        if "out/depth%d" % s in rec:
            e = out["extras"][s]
            close(e["depth"], rec["out/depth%d" % s], 2e-6, 2e-5)
            close(e["disp_up"], rec["out/disp_up%d" % s], 2e-6, 2e-5)
            for f, nm in ((0, "m1"), (1, "p1")):
                close(e["color"][f], rec["out/color_%s_%d" % (nm, s)], 2e-5, 2e-5)
                close(e["grid"][f], rec["out/sample_%s_%d" % (nm, s)], 1e-5, 1e-5)
    for i, n in enumerate(("aa_left", "t_left", "aa_right", "t_right")):
        close_pose(out["d_pose"][i], rec["grad/" + n], sample[("target_image", 0)][:, 0].numel())


@pytest.mark.parametrize("B,H,W,ns,auto_mask", [(1, 48, 64, 4, True), (3, 80, 200, 4, True), (2, 64, 96, 2, False),
                                                 (1, 50, 70, 1, True)])
def test_chain_vs_oracle(gpu_device, B, H, W, ns, auto_mask):
    """Other shapes (ragged tiles: sizes that are not multiples of the 64x16 tile), fewer scales and
    auto_mask off, against the CPU oracle."""
    from deep_visual_slam_amd import synth
    from oracle import loss_chain as O
    sample = synth.parity_sample(B, H, W, seed=77)
    disps = [d for d in synth.parity_disps(B, H, W, seed=3)]
    if H % 8 or W % 8:   # decoder pyramids only exist for sizes divisible by 2^s: use a single scale
        disps = disps[:1]
    disps = disps[:ns]
    ns = len(disps)
    poses = synth.parity_poses(B, seed=5)
    g = torch.Generator().manual_seed(7)
    noise = [torch.randn(B, 2, H, W, generator=g) for _ in range(ns)]
    _, ref_losses, ref_grads = O.loss_chain_with_grads(sample, disps, poses, noise if auto_mask else None,
                                                       num_scales=ns, auto_mask=auto_mask)
    out = run_chain(gpu_device, sample, disps, poses, noise if auto_mask else None, ns, materialize=False,
                    auto_mask=auto_mask)
    close(out["total"], ref_losses["loss"], 1e-7, 1e-5)
    for s in range(ns):
        close(out["losses"][s], ref_losses["loss/%d" % s], 1e-7, 1e-5)
        close_frac(out["d_disp"][s], ref_grads["disp"][s], atol=2e-8, rtol=2e-3)
    for i in range(4):
        close_pose(out["d_pose"][i], ref_grads["pose"][i], B * H * W)


def test_chain_full_resolution_checksums(gpu_device):
    """480x640 (BASELINE.json's size): checksums of the reference's outputs; inputs are regenerated
    from the same seeded synth the generator used."""
    from deep_visual_slam_amd import synth
    rec = load_golden("chain_b1_480x640_sums.npz")
    B, H, W = 1, 480, 640
    sample, disps, poses = synth.parity_sample(B, H, W), synth.parity_disps(B, H, W), synth.parity_poses(B)
    torch.manual_seed(7)
    noise = [torch.randn(B, 2, H, W) for _ in range(4)]
    out = run_chain(gpu_device, sample, disps, poses, noise, 4)
    close(out["total"], rec["loss"], 1e-7, 1e-5)
    for s in range(4):
        close(out["losses"][s], rec["loss/%d" % s], 1e-7, 1e-5)
        g = out["d_disp"][s].double().cpu().numpy()
        assert abs(np.abs(g).sum() - rec["grad/disp%d#abs" % s]) < 5e-3 * rec["grad/disp%d#abs" % s]
        e = out["extras"][s]
        for f, nm in ((0, "m1"), (1, "p1")):
            c = e["color"][f].double().cpu().numpy()
            assert abs(c.sum() - rec["out/color_%s_%d#sum" % (nm, s)]) < 1e-5 * abs(rec["out/color_%s_%d#sum" % (nm, s)])
            assert abs((c * c).sum() - rec["out/color_%s_%d#sq" % (nm, s)]) < 1e-5 * rec["out/color_%s_%d#sq" % (nm, s)]
    for i, n in enumerate(("aa_left", "t_left", "aa_right", "t_right")):
        close_pose(out["d_pose"][i], rec["grad/" + n], sample[("target_image", 0)][:, 0].numel())


def test_philox_noise_statistics(gpu_device):
    """Without injected noise the kernel draws its own tie-break noise; the loss must stay within the
    noise scale (1e-5) of the noise-free value and be reproducible for a fixed seed."""
    from deep_visual_slam_amd import ops, synth
    B, H, W = 2, 96, 128
    sample, disps, poses = synth.parity_sample(B, H, W), synth.parity_disps(B, H, W), synth.parity_poses(B)
    dev = gpu_device
    args = [sample[k].to(dev) for k in (("target_image", 0), ("source_left", 0), ("source_right", 0), ("K", 0), ("inv_K", 0))]
    T_l = ops.pose_to_mat(poses[0][:, 0].to(dev), poses[1][:, 0].to(dev), True)
    T_r = ops.pose_to_mat(poses[2][:, 0].to(dev), poses[3][:, 0].to(dev), False)
    dd = [d.to(dev) for d in disps]
    zero = torch.zeros(4, B, 2, H, W, device=dev)
    l0, _, _ = ops.loss_chain(*args, T_l, T_r, dd, noise=zero)
    l1, s1, _ = ops.loss_chain(*args, T_l, T_r, dd, seed=123)
    l2, s2, _ = ops.loss_chain(*args, T_l, T_r, dd, seed=123)
    l3, s3, _ = ops.loss_chain(*args, T_l, T_r, dd, seed=124)
    assert torch.equal(l1, l2) and torch.equal(s1, s2)
    assert not torch.equal(s1, s3)
    assert float((l1 - l0).abs().max()) < 5e-5


def test_backward_by_scale_on_two_streams_equals_one_launch(gpu_device):
    """The loss-chain backward launched by scale (scale 0 on the current stream, the coarser scales and the pose
    reduction on a second stream, consumers waiting for their events) gives the gradients of the single launch."""
    from deep_visual_slam_amd import conv as DC, ops, synth
    B, H, W = 2, 96, 128
    sample = synth.parity_sample(B, H, W)
    tgt, left, right = [sample[k].to(gpu_device) for k in (("target_image", 0), ("source_left", 0), ("source_right", 0))]
    K, inv_K = sample[("K", 0)].to(gpu_device), sample[("inv_K", 0)].to(gpu_device)
    torch.manual_seed(3)
    feats = [torch.randn(B, 16, H >> s, W >> s, device=gpu_device).contiguous(memory_format=torch.channels_last) * 0.3
             for s in range(4)]
    ws = [(torch.randn(1, 16, 3, 3, device=gpu_device) * 0.2).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    bs = [torch.zeros(1, device=gpu_device) for _ in range(4)]
    aa = torch.randn(2, B, 3, device=gpu_device) * 0.01
    tt = torch.randn(2, B, 3, device=gpu_device) * 0.02
    noise = torch.randn(4, B, 2, H, W, device=gpu_device)

    def run(split):
        old = ops._CHAIN_SPLIT
        ops._CHAIN_SPLIT = split
        aux = torch.cuda.Stream(device=gpu_device) if split else None
        try:
            f = [t.clone().requires_grad_(True) for t in feats]
            a, t = aa.clone().requires_grad_(True), tt.clone().requires_grad_(True)
            disps = [DC.head_conv2d(f[s], ws[s], bs[s], 0, 1, "sigmoid") for s in range(4)]
            T_l, T_r = ops.pose_to_mat(a[0], t[0], True), ops.pose_to_mat(a[1], t[1], False)
            losses, _, _ = ops.loss_chain(tgt, left, right, K, inv_K, T_l, T_r, disps, noise=noise, aux_stream=aux)
            assert ops._LossChain is not None
            (losses * torch.tensor([1.0, 0.5, 0.25, 0.125], device=gpu_device)).sum().backward()
            torch.cuda.synchronize()
            return [x.grad.clone() for x in f] + [a.grad.clone(), t.grad.clone()]
        finally:
            ops._CHAIN_SPLIT = old

    from deep_visual_slam_amd import gradsink
    seen, orig = [], gradsink.StreamSet.set_pending
    gradsink.StreamSet.set_pending = lambda self, t, ev: (seen.append(t.data_ptr()), orig(self, t, ev))[1]
    try:
        one, two = run(False), run(True)
    finally:
        gradsink.StreamSet.set_pending = orig
    assert len(seen) == 5                                 # d disp_1..3 and the two d T were handed out with events
    for g1, g2 in zip(one, two):
        assert torch.isfinite(g2).all()
        assert float((g1 - g2).abs().max()) <= 1e-6 * float(g1.abs().max()) + 1e-12
