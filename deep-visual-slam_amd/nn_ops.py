"""Network primitives used by ResnetEncoder / DepthNet / PoseNet.

Every primitive is GPU-only.  The convolution engine is selected by DVS_CONV_BACKEND:
  "hip"    hand-written gfx950 implicit-GEMM kernels of libdvslam_hip.so (default where a kernel
           exists for the shape);
  "miopen" PyTorch-ROCm's library convolution -- bring-up/A-B baseline only, never the CPU.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib


def _require_gpu(x, who):
    if not x.is_cuda:
        raise _lib.DvsError("%s: GPU tensors only (got %s); this package has no CPU path" % (who, x.device))
    _lib.lib()  # the HIP library must be present even when a library conv is selected


def conv_backend():
    return os.environ.get("DVS_CONV_BACKEND", "hip")


def conv2d(x, weight, bias=None, stride=1, padding=0, reflect_pad=0):
    """Convolution with optional ReflectionPad2d(reflect_pad) in front (model/layers.py:121-136)."""
    _require_gpu(x, "conv2d")
    from . import conv as _conv
    if conv_backend() == "hip" and _conv.supported(x, weight, stride, padding, reflect_pad):
        return _conv.conv2d(x, weight, bias, stride, padding, reflect_pad)
    if reflect_pad:
        x = F.pad(x, (reflect_pad,) * 4, mode="reflect")
    return F.conv2d(x, weight, bias, stride, padding)


def batch_norm(x, bn, relu=False, residual=None):
    """nn.BatchNorm2d forward (batch statistics + running-stat update in training mode), optionally
    followed by `+ residual` and ReLU: the BasicBlock tail."""
    _require_gpu(x, "batch_norm")
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                     bn.training or not bn.track_running_stats, bn.momentum, bn.eps)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


def max_pool_3x3_s2(x):
    _require_gpu(x, "max_pool")
    return F.max_pool2d(x, 3, 2, 1)


def elu(x):
    return F.elu(x, inplace=True)


def upsample_nearest2x(x):
    """model/layers.py:196-199."""
    _require_gpu(x, "upsample")
    return F.interpolate(x, scale_factor=2, mode="nearest")
