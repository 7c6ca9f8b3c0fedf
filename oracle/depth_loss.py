"""ORACLE (test infrastructure only) -- CPU restatement of the supervised depth learner's losses.

Only ``tests/`` may import this.  Restates depth/depth_learner.py:32-117 (disp_to_depth, get_smooth_loss, silog_loss,
multi_scale_loss) in plain torch-CPU arithmetic, dtype-parametric, with the bilinear upsample written out
(oracle/loss_chain.upsample_bilinear: F.interpolate bilinear / align_corners=False semantics) instead of calling
F.interpolate.  PINNED: tests/golden/make_golden_depth.py imported the reference's depth/depth_learner.py in the dev
container (it needs torch only) and stored inputs, the three losses and the gradients w.r.t. the four disparity maps
(tests/golden/depth_learner_*.npz, checked by tests/test_oracle_golden.py).
"""
import torch

from .loss_chain import upsample_bilinear

ALPHAS = (1.0, 0.5, 0.25, 0.125)


def disp_to_depth(disp, min_depth, max_depth):
    """depth_learner.py:32-38."""
    min_disp, max_disp = 1.0 / max_depth, 1.0 / min_depth
    return 1.0 / (min_disp + (max_disp - min_disp) * disp)


def smooth_loss(disp, img):
    """depth_learner.py:51-73."""
    mean = disp.mean(dim=[2, 3], keepdim=True).clamp(min=1e-7)
    nd = disp / mean
    ddx = torch.abs(nd[:, :, :, 1:] - nd[:, :, :, :-1])
    ddy = torch.abs(nd[:, :, 1:, :] - nd[:, :, :-1, :])
    idx = torch.abs(img[:, :, :, 1:] - img[:, :, :, :-1]).mean(1, keepdim=True)
    idy = torch.abs(img[:, :, 1:, :] - img[:, :, :-1, :]).mean(1, keepdim=True)
    return (ddx * torch.exp(-idx)).mean() + (ddy * torch.exp(-idy)).mean()


def silog_loss(pred, target, valid, variance_focus=0.85):
    """depth_learner.py:75-95."""
    pred = torch.clamp(pred, min=1e-6)
    d = torch.log(pred[valid]) - torch.log(target[valid])
    return torch.sqrt((d ** 2).mean() - variance_focus * d.mean() ** 2)


def multi_scale_loss(pred_depths, gt_depth, rgb, valid, silog_weight=1.0, smooth_weight=0.1):
    """depth_learner.py:97-117 -> (total, total_silog, total_smooth, per-scale silog list, per-scale smooth list)."""
    H, W = gt_depth.shape[-2:]
    t_sm = t_si = 0.0
    silogs, smooths = [], []
    for i, a in enumerate(ALPHAS[:len(pred_depths)]):
        up = upsample_bilinear(pred_depths[i], H, W)
        sm, si = smooth_loss(up, rgb), silog_loss(up, gt_depth, valid)
        smooths.append(sm)
        silogs.append(si)
        t_sm = t_sm + a * sm
        t_si = t_si + a * si
    return silog_weight * t_si + smooth_weight * t_sm, t_si, t_sm, silogs, smooths
