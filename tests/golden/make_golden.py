"""Generate golden vectors by IMPORTING the reference loss chain in the dev container.

Run (dev container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports vo/learner_func.py and vo/learner_new.py from /root/reference (read-only, nothing is
copied), drives ``MonodepthTrainer.process_batch`` with stand-in networks that return fixed
disparity pyramids / pose vectors, and stores inputs + expected outputs as small .npz fixtures.
The auto-mask tie-break noise (vo/learner_new.py:226-229) is captured by re-seeding
``torch.manual_seed(7)`` and drawing the same four ``torch.randn`` tensors.
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path[:0] = ["/root/reference", "/root/reference/vo"]

import learner_func as RF  # noqa: E402  (reference)
import learner_new as RL  # noqa: E402  (reference)
from deep_visual_slam_amd import synth  # noqa: E402

NOISE_SEED = 7


class StubDepth(nn.Module):
    def __init__(self, disps):
        super().__init__()
        self.disps = nn.ParameterList([nn.Parameter(d.clone()) for d in disps])

    def forward(self, x):
        return {("disp", s): self.disps[s] * 1.0 for s in range(len(self.disps))}


class StubPose(nn.Module):
    def __init__(self, poses):
        super().__init__()
        self.p = nn.ParameterList([nn.Parameter(p.clone()) for p in poses])
        self.calls = 0

    def forward(self, x):
        i = self.calls % 2
        self.calls += 1
        return self.p[2 * i] * 1.0, self.p[2 * i + 1] * 1.0


def config(b, h, w):
    return {"Train": dict(num_source=1, batch_size=b, img_h=h, img_w=w, smoothness_ratio=0.001,
                          auto_mask=True, ssim_ratio=0.85, min_depth=0.1, max_depth=10.0,
                          use_compile=False)}


def np32(t):
    return t.detach().cpu().numpy()


def oob_poses(b):
    """Motions large enough that a good part of every warp leaves the image: frame -1 samples beyond the right / bottom
    border, frame +1 beyond the left / top one (border clamp + zero coordinate gradient of F.grid_sample,
    vo/learner_new.py:165-170)."""
    rng = np.random.RandomState(23)
    out = []
    for sign in (1.0, 1.0):       # frame -1 is built with invert=True, so equal parameters move the two warps apart
        aa = np.tile(np.array([[[[0.05, -0.08, 0.03]]]]), (b, 1, 1, 1)) * sign + rng.uniform(-0.01, 0.01, size=(b, 1, 1, 3))
        tt = np.tile(np.array([[[[0.08, 0.05, 0.02]]]]), (b, 1, 1, 1)) * sign + rng.uniform(-0.01, 0.01, size=(b, 1, 1, 3))
        out += [torch.from_numpy(aa).float(), torch.from_numpy(tt).float()]
    return out


def run_chain(b, h, w, num_scales=4, dump_outputs=True, poses=None):
    sample = synth.parity_sample(b, h, w)
    disps = synth.parity_disps(b, h, w)
    poses = synth.parity_poses(b) if poses is None else poses
    dn, pn = StubDepth(disps), StubPose(poses)
    tr = RL.MonodepthTrainer(dn, pn, config(b, h, w), torch.device("cpu"))
    tr.num_scales = num_scales
    torch.manual_seed(NOISE_SEED)
    outputs, losses = tr.process_batch(dict(sample))
    losses["loss"].backward()
    torch.manual_seed(NOISE_SEED)
    noise = [torch.randn(b, 2, h, w) for _ in range(num_scales)]

    rec = {}
    for k in (("source_left", 0), ("target_image", 0), ("source_right", 0), ("K", 0), ("inv_K", 0)):
        rec["in/%s" % k[0]] = np32(sample[k])
    for s in range(4):
        rec["in/disp%d" % s] = np32(disps[s])
    for i, n in enumerate(("aa_left", "t_left", "aa_right", "t_right")):
        rec["in/%s" % n] = np32(poses[i])
    for s in range(num_scales):
        rec["in/noise%d" % s] = np32(noise[s])
        rec["loss/%d" % s] = np32(losses["loss/%d" % s])
        rec["grad/disp%d" % s] = np32(dn.disps[s].grad)
        rec["out/identity_selection%d" % s] = np32(outputs["identity_selection/%d" % s]).astype(np.uint8)
        if dump_outputs:
            rec["out/depth%d" % s] = np32(outputs[("depth", s)])
            rec["out/disp_up%d" % s] = np32(outputs[("disp_up", s)])
            for f, nm in ((-1, "m1"), (1, "p1")):
                rec["out/color_%s_%d" % (nm, s)] = np32(outputs[("color", f, s)])
                rec["out/sample_%s_%d" % (nm, s)] = np32(outputs[("sample", f, s)].contiguous())
    rec["loss"] = np32(losses["loss"])
    for i, n in enumerate(("aa_left", "t_left", "aa_right", "t_right")):
        rec["grad/%s" % n] = np32(pn.p[i].grad)
    rec["out/T_m1"] = np32(outputs[("cam_T_cam", 0, -1)])
    rec["out/T_p1"] = np32(outputs[("cam_T_cam", 0, 1)])
    rec["meta/num_scales"] = np.array(num_scales)
    # fraction of scale-0 samples clamped at each border (left, right, top, bottom), per frame
    for f, nm in ((-1, "m1"), (1, "p1")):
        g = outputs[("sample", f, 0)].detach()
        rec["meta/clamped_%s" % nm] = np.array([float((g[..., 0] < -1).float().mean()), float((g[..., 0] > 1).float().mean()),
                                                float((g[..., 1] < -1).float().mean()), float((g[..., 1] > 1).float().mean())])
    return rec


def run_operators(b, h, w):
    """Per-operator fixtures: forward values + input grads under a fixed cotangent."""
    g = torch.Generator().manual_seed(3)
    sample = synth.parity_sample(b, h, w)
    rec = {}
    # a4 pose -> matrix (both inverts), incl. a zero rotation (|v| = 0 edge case, eps path)
    aa = torch.randn(b + 1, 1, 3, generator=g) * 0.05
    aa[-1] = 0
    tt = torch.randn(b + 1, 1, 3, generator=g) * 0.1
    cot = torch.randn(b + 1, 4, 4, generator=g)
    for inv in (False, True):
        a, t = aa.clone().requires_grad_(True), tt.clone().requires_grad_(True)
        M = RF.transformation_from_parameters(a, t, invert=inv)
        (M * cot).sum().backward()
        tag = "inv" if inv else "fwd"
        rec["pose/%s/M" % tag] = np32(M)
        rec["pose/%s/d_aa" % tag] = np32(a.grad)
        rec["pose/%s/d_t" % tag] = np32(t.grad)
    rec["pose/aa"], rec["pose/t"], rec["pose/cot"] = np32(aa), np32(tt), np32(cot)

    # a5 upsample (each scale) + disp_to_depth
    disps = synth.parity_disps(b, h, w)
    for s in range(4):
        d = disps[s].clone().requires_grad_(True)
        up = F.interpolate(d, [h, w], mode="bilinear", align_corners=False)
        sc, depth = RF.disp_to_depth(up, 0.1, 10.0)
        cot_d = torch.randn(depth.shape, generator=g)
        (depth * cot_d).sum().backward()
        rec["up/%d/disp" % s] = np32(disps[s])
        rec["up/%d/disp_up" % s] = np32(up)
        rec["up/%d/depth" % s] = np32(depth)
        rec["up/%d/cot" % s] = np32(cot_d)
        rec["up/%d/d_disp" % s] = np32(d.grad)

    # a6-a8 backproject -> project -> grid_sample with grads to depth and T
    depth = (1.0 / (0.1 + 9.9 * disps[0])).clone().requires_grad_(True)
    T = RF.transformation_from_parameters(aa[:b] * 0.4, tt[:b] * 0.5).clone().requires_grad_(True)
    bp, pj = RF.BackprojectDepth(b, h, w), RF.Project3D(b, h, w)
    cam = bp(depth, sample[("inv_K", 0)])
    grid = pj(cam, sample[("K", 0)], T)
    src = sample[("source_right", 0)]
    color = F.grid_sample(src, grid, padding_mode="border", align_corners=True)
    cot_c = torch.randn(color.shape, generator=g)
    (color * cot_c).sum().backward()
    rec["warp/depth"], rec["warp/T"] = np32(depth), np32(T)
    rec["warp/K"], rec["warp/inv_K"] = np32(sample[("K", 0)]), np32(sample[("inv_K", 0)])
    rec["warp/src"], rec["warp/cam"] = np32(src), np32(cam)
    rec["warp/grid"], rec["warp/color"], rec["warp/cot"] = np32(grid.contiguous()), np32(color), np32(cot_c)
    rec["warp/d_depth"], rec["warp/d_T"] = np32(depth.grad), np32(T.grad)

    # a8 edge cases: grid that leaves the image on all four sides (border clamp, zero grad)
    gr = (torch.rand(b, h, w, 2, generator=g) * 2.6 - 1.3).requires_grad_(True)
    col = F.grid_sample(src, gr, padding_mode="border", align_corners=True)
    (col * cot_c).sum().backward()
    rec["gs/grid"], rec["gs/color"], rec["gs/d_grid"] = np32(gr), np32(col), np32(gr.grad)

    # a9 SSIM + reprojection loss, grads wrt pred
    tgt = sample[("target_image", 0)]
    pred = color.detach().clone().requires_grad_(True)
    ss = RF.SSIM()(pred, tgt)
    cot_s = torch.randn(ss.shape, generator=g)
    (ss * cot_s).sum().backward()
    rec["ssim/pred"], rec["ssim/target"] = np32(pred), np32(tgt)
    rec["ssim/out"], rec["ssim/cot"], rec["ssim/d_pred"] = np32(ss), np32(cot_s), np32(pred.grad)
    tr = RL.MonodepthTrainer(None, None, config(b, h, w), torch.device("cpu"))
    pred2 = color.detach().clone().requires_grad_(True)
    rl = tr._compute_reprojection_loss(pred2, tgt)
    cot_r = torch.randn(rl.shape, generator=g)
    (rl * cot_r).sum().backward()
    rec["reproj/out"], rec["reproj/cot"], rec["reproj/d_pred"] = np32(rl), np32(cot_r), np32(pred2.grad)

    # a11 smoothness on mean-normalised disparity (vo/learner_new.py:246-250)
    d = F.interpolate(disps[1], [h, w], mode="bilinear", align_corners=False).clone().requires_grad_(True)
    mean_disp = torch.clamp(d.mean(2, True).mean(3, True), min=0.001)
    sm = RF.get_smooth_loss(d / (mean_disp + 1e-7), tgt)
    sm.backward()
    rec["smooth/disp"], rec["smooth/img"] = np32(d), np32(tgt)
    rec["smooth/out"], rec["smooth/d_disp"] = np32(sm), np32(d.grad)
    return rec


def checksums(rec):
    """Size-independent summary for the full-resolution case (keeps the fixture small)."""
    out = {}
    for k, v in rec.items():
        if k.startswith("in/") and v.size > 4096:
            continue
        if v.size > 4096:
            v64 = v.astype(np.float64)
            out[k + "#sum"] = np.array(v64.sum())
            out[k + "#abs"] = np.array(np.abs(v64).sum())
            out[k + "#sq"] = np.array((v64 * v64).sum())
        else:
            out[k] = v
    return out


def main():
    torch.set_num_threads(8)
    if "--oob" in sys.argv:        # only the out-of-image chain fixture (added in round 2; the others are unchanged)
        rec = run_chain(2, 48, 64, poses=oob_poses(2))
        print("clamped fractions (left, right, top, bottom): frame -1", rec["meta/clamped_m1"], "frame +1", rec["meta/clamped_p1"])
        np.savez_compressed(os.path.join(HERE, "chain_b2_48x64_oob.npz"), **rec)
        return
    np.savez_compressed(os.path.join(HERE, "chain_b2_48x64.npz"), **run_chain(2, 48, 64))
    np.savez_compressed(os.path.join(HERE, "chain_b2_96x128.npz"), **run_chain(2, 96, 128, dump_outputs=False))
    np.savez_compressed(os.path.join(HERE, "chain_b1_48x64_s1.npz"), **run_chain(1, 48, 64, num_scales=1))
    np.savez_compressed(os.path.join(HERE, "ops_b2_48x64.npz"), **run_operators(2, 48, 64))
    # full-resolution case: inputs are regenerated from synth (seeded), only checksums are stored
    full = run_chain(1, 480, 640)
    np.savez_compressed(os.path.join(HERE, "chain_b1_480x640_sums.npz"), **checksums(full))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
