"""Hand-written gfx950 convolution engine: autograd wrappers over dvs_conv2d_* of libdvslam_hip.so.

Tensors keep the reference's logical NCHW shapes but live in torch's channels_last memory format
(NHWC in HBM), weights likewise ([Cout][kh][kw][Cin] in HBM); nothing is transposed on the way in or
out of a kernel.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, ConvFusion, check, ptr

ACT = {None: 0, "none": 0, "relu": 1, "elu": 2, "sigmoid": 3}
CL = torch.channels_last


def _nhwc(t):
    """Contiguous-NHWC view requirement (a [B,1,H,W] or [B,C,1,1] tensor is both NCHW and NHWC)."""
    if t.dtype != torch.float32:
        raise _lib.DvsError("fp32 tensors only (got %s)" % t.dtype)
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def _desc(x_shape, w_shape, stride, pad, reflect):
    B, Cin, H, W = x_shape
    Cout, _, kh, kw = w_shape
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, Cout
    d.kh, d.kw, d.stride, d.pad, d.pad_mode = kh, kw, stride, pad, int(bool(reflect))
    return d


def out_hw(H, W, kh, kw, stride, pad):
    return (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1


def conv2d_forward(x, weight, bias=None, stride=1, pad=0, reflect=False, act=None, x2=None, in_scale=None,
                   in_shift=None, in_relu=False, nchw_planar=False, stats=None):
    """Raw forward launch (no autograd).  x: logical [B,Cin,H,W] (NHWC memory, or planar NCHW when
    nchw_planar); with x2 the logical input is cat([upsample2x(x), x2], 1)."""
    if not (x.is_cuda and weight.is_cuda):
        raise _lib.DvsError("conv2d: GPU tensors only; this package has no CPU path")
    if nchw_planar:
        # encoder conv1: planar image in, weights packed [Cout][Cin][kh][8] (kw padded to 8 with zeros)
        if not x.is_contiguous():
            x = x.contiguous()
        B, Cin, H, W = x.shape
        w = torch.nn.functional.pad(weight.contiguous(), (0, 8 - weight.shape[3]))
    else:
        w = _nhwc(weight)
        x = _nhwc(x)
        B, C1, H, W = x.shape
        Cin = C1
        if x2 is not None:
            x2 = _nhwc(x2)
            H, W = x2.shape[2], x2.shape[3]
            if (x.shape[2] * 2, x.shape[3] * 2) != (H, W):
                raise _lib.DvsError("upsample+concat fusion: skip tensor must be exactly 2x the coarse one")
            Cin = C1 + x2.shape[1]
    if weight.shape[1] != Cin:
        raise _lib.DvsError("weight expects %d input channels, got %d" % (weight.shape[1], Cin))
    d = _desc((B, Cin, H, W), weight.shape, stride, pad, reflect)
    Ho, Wo = out_hw(H, W, d.kh, d.kw, stride, pad)
    y = torch.empty((B, d.Cout, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    f = ConvFusion()
    if x2 is not None:
        f.x2, f.C1 = ptr_nhwc(x2), x.shape[1]
    if in_scale is not None:
        f.in_scale, f.in_shift, f.in_relu = ptr(in_scale), ptr(in_shift), int(bool(in_relu))
    f.nchw_planar = int(bool(nchw_planar))
    f.act = ACT[act]
    if stats is not None:
        f.stats = ptr(stats)
    check(_lib.lib().dvs_conv2d_fwd(x.data_ptr(), w.data_ptr(), ptr(bias), y.data_ptr(), C.byref(d), C.byref(f),
                                    _lib.stream()), "dvs_conv2d_fwd")
    return y


def ptr_nhwc(t):
    if not t.is_cuda:
        raise _lib.DvsError("GPU tensors only")
    return t.data_ptr()


def supported(x, weight, stride, padding, reflect_pad):
    """True when a hand-written kernel exists for this problem."""
    return False


def conv2d(x, weight, bias, stride, padding, reflect_pad):
    raise NotImplementedError
