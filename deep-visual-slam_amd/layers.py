"""Operator surface of the reference's model/layers.py (== vo/learner_func.py), backed by the C-ABI.

Names, signatures and return shapes follow model/layers.py:16-268 so callers import unchanged.
"""
import numpy as np
import torch
import torch.nn as nn

from . import nn_ops, ops


def disp_to_depth(disp, min_depth, max_depth):
    """model/layers.py:16-25."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    depth = 1 / scaled_disp
    return scaled_disp, depth


def transformation_from_parameters(axisangle, translation, invert=False):
    """model/layers.py:28-45: [B,1,3],[B,1,3] -> [B,4,4] (HIP kernel dvs_pose_to_mat_fwd/bwd)."""
    return ops.pose_to_mat(axisangle, translation, invert)


def get_translation_matrix(translation_vector):
    """model/layers.py:48-61."""
    t = translation_vector.contiguous().view(-1, 1, 3)
    return ops.pose_to_mat(torch.zeros_like(t), t, False)


def rot_from_axisangle(vec):
    """model/layers.py:64-103."""
    return ops.pose_to_mat(vec, torch.zeros_like(vec), False)


class Conv3x3(nn.Module):
    """Layer to pad and convolve input (model/layers.py:121-136)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = use_refl
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x, act=None, skip=None, upsample=False, passthrough=False):
        """act / skip / upsample are fusion hooks of this implementation (one launch for upsample + concat + pad + conv
        + activation); plain `conv(x)` is the reference call.  passthrough: (y, x') -- see nn_ops.conv2d."""
        pad = dict(reflect_pad=1) if self.use_refl else dict(padding=1)
        return nn_ops.conv2d(x, self.conv.weight, self.conv.bias, 1, act=act, x2=skip, upsample=upsample,
                             passthrough=passthrough, **pad)


class ConvBlock(nn.Module):
    """Layer to perform a convolution followed by ELU (model/layers.py:106-118)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x, skip=None, upsample=False):
        """ELU(conv(x)); with `skip`/`upsample` the input is cat([upsample(x), skip], 1) / upsample(x)
        (model/depthnet.py:79-85) gathered inside the convolution."""
        return self.conv(x, act="elu", skip=skip, upsample=upsample)


class BackprojectDepth(nn.Module):
    """Layer to transform a depth image into a point cloud (model/layers.py:139-168)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        return ops.backproject(depth.view(-1, 1, self.height, self.width), inv_K)


class Project3D(nn.Module):
    """Layer which projects 3D points into a camera with intrinsics K and at position T
    (model/layers.py:171-193)."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        return ops.project(points, K, T, self.height, self.width, self.eps)


def get_smooth_loss(disp, img):
    """Edge-aware smoothness of a disparity image (model/layers.py:202-215)."""
    return ops.smooth_loss(disp, img)


class SSIM(nn.Module):
    """Layer to compute the SSIM loss between a pair of images (model/layers.py:218-248)."""

    def __init__(self):
        super().__init__()
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, x, y):
        return ops.ssim(x, y)


def upsample(x):
    """Upsample input tensor by a factor of 2 (model/layers.py:196-199)."""
    return nn_ops.upsample_nearest2x(x)


def compute_depth_errors(gt, pred):
    """model/layers.py:251-268 (evaluation metric; plain tensor arithmetic)."""
    thresh = torch.max((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
