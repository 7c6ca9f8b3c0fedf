"""Network primitives used by ResnetEncoder / DepthNet / PoseNet.

Every primitive is GPU-only and keeps activations in channels_last (NHWC) memory.  The convolution
engine is selected by DVS_CONV_BACKEND:
  "hip"    (default) hand-written gfx950 implicit-GEMM kernels of libdvslam_hip.so wherever they cover
           the shape (everything except the 1- and 6-channel heads);
  "miopen" PyTorch-ROCm's library convolution composed with eager pad / upsample / cat / activation --
           bring-up and A/B baseline only.  Neither choice ever runs on the CPU.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib
from . import conv as _conv

CL = torch.channels_last
_ACT = {None: lambda v: v, "relu": F.relu, "elu": F.elu, "sigmoid": torch.sigmoid}


def _require_gpu(x, who):
    if not x.is_cuda:
        raise _lib.DvsError("%s: GPU tensors only (got %s); this package has no CPU path" % (who, x.device))
    _lib.lib()  # the HIP library must be present even when a library conv is selected


_batch_groups = 1


class batch_groups:
    """Context: the batch handed to conv_bn_act holds `n` independent sub-batches back to back (PoseNet evaluates
    its two frame pairs as one batch of 2B); BatchNorm statistics, normalisation and running-stat updates are
    kept per sub-batch, so the result equals n separate forward calls (vo/learner_new.py:113-114)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global _batch_groups
        self.prev, _batch_groups = _batch_groups, self.n

    def __exit__(self, *exc):
        global _batch_groups
        _batch_groups = self.prev


def conv_backend():
    return os.environ.get("DVS_CONV_BACKEND", "hip")


def conv2d(x, weight, bias=None, stride=1, padding=0, reflect_pad=0, act=None, x2=None, upsample=False,
           planar_norm=None):
    """act(conv(pad(input), weight) + bias) where input is x, upsample2x(x) (upsample=True) or
    cat([upsample2x(x), x2], 1) (x2 given); reflect_pad=1 puts ReflectionPad2d(1) in front
    (model/layers.py:121-136); planar_norm=(scale, shift) is the encoder's conv1 on the raw planar image
    with (x - 0.45) / 0.225 folded in (model/resnet_encoder.py:102-103)."""
    _require_gpu(x, "conv2d")
    planar = planar_norm is not None
    if conv_backend() == "hip":
        if _conv.supported(x, weight, x2, planar, upsample):
            return _conv.conv2d(x, weight, bias, stride, padding, reflect_pad, act, x2=x2, upsample=upsample,
                                planar_norm=planar_norm)
        if _conv.head_supported(x, weight, stride, padding, reflect_pad, x2, upsample, planar):
            return _conv.head_conv2d(x, weight, bias, padding, reflect_pad, act)
    # library path (also serves the 1- and 6-channel heads)
    if planar:
        sc, sh = planar_norm
        x = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if upsample or x2 is not None:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if x2 is not None:
        x = torch.cat([x, x2], 1)
    if reflect_pad:
        x = F.pad(x, (reflect_pad,) * 4, mode="reflect")
    return _ACT[act](F.conv2d(x, weight, bias, stride, padding))


_fold_cache = {}


def folded_bn(weight, bn):
    """(w', b') with an eval-mode BatchNorm2d folded into the convolution in front of it:
    bn(conv(x, w)) = conv(x, w * s) + (beta - running_mean * s),  s = gamma / sqrt(running_var + eps).
    Cached per weight tensor and refreshed when any of the five tensors is modified in place (optimiser step,
    load_state_dict), so an inference loop pays for the fold once."""
    gamma = bn.weight if bn.weight is not None else torch.ones_like(bn.running_var)
    beta = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_var)
    key = id(weight)
    ver = (weight._version, gamma._version, beta._version, bn.running_mean._version, bn.running_var._version,
           weight.data_ptr(), bn.running_mean.data_ptr(), bn.eps)
    hit = _fold_cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    with torch.no_grad():
        s = gamma / torch.sqrt(bn.running_var + bn.eps)
        w_f = (weight * s.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
        b_f = (beta - bn.running_mean * s).contiguous()
    _fold_cache[key] = (ver, w_f, b_f)
    return w_f, b_f


def inference_mode(bn):
    """The fast path below applies when nothing will be differentiated and the BatchNorm uses its running statistics."""
    return (not bn.training) and bn.track_running_stats and bn.running_mean is not None and not torch.is_grad_enabled()


def conv_bn_act(x, weight, bn, stride=1, padding=0, relu=True, residual=None, res=None, planar_norm=None):
    """relu(bn(conv(x)) + residual') -- the conv -> BatchNorm2d -> (+identity) -> ReLU groups of torchvision's
    BasicBlock / stem (model/resnet_encoder.py:100-111).  `residual` is the identity tensor; `res` =
    (weight, bn, stride) describes the 1x1 downsample branch conv -> bn applied to `residual` instead.
    HIP path: the conv epilogue accumulates the batch statistics, BN + add + ReLU is one fused pass."""
    _require_gpu(x, "conv_bn_act")
    from . import bn as _bn
    planar = planar_norm is not None
    if (inference_mode(bn) and conv_backend() == "hip" and _conv.supported(x, weight, None, planar)
            and (res is None or (inference_mode(res[1]) and _conv.supported(residual, res[0])))):
        # inference (vo/predict.py:20-42,63-86: .eval() + torch.no_grad()): BatchNorm folded into the weights, bias +
        # identity + ReLU in the conv epilogue -- one kernel per conv, no normalisation passes
        sc, sh = planar_norm if planar else (None, None)
        if res is not None:
            wd, bd = folded_bn(res[0], res[1])
            residual = _conv.conv2d_forward(residual, wd, bd, res[2], 0)
        w_f, b_f = folded_bn(weight, bn)
        return _conv.conv2d_forward(x, w_f, b_f, stride, padding, act="relu" if relu else None, in_scale=sc, in_shift=sh,
                                    nchw_planar=planar, residual=residual)
    fused = (conv_backend() == "hip" and _conv.supported(x, weight, None, planar) and _bn.supported_c(weight.shape[0], bn)
             and (res is None or (_conv.supported(residual, res[0]) and _bn.supported_c(res[0].shape[0], res[1]))))
    G = _batch_groups
    if G != 1 and not fused:
        raise _lib.DvsError("batch_groups(%d) needs the fused HIP conv + BatchNorm path" % G)
    if fused:
        y, st = _conv.conv2d(x, weight, None, stride, padding, planar_norm=planar_norm, want_stats=G)
        if res is not None:
            yd, std = _conv.conv2d(residual, res[0], None, res[2], 0, want_stats=G)
            return _bn.bn_act(y, bn, st, relu, residual=yd, res_bn=res[1], res_stats=std, groups=G)
        return _bn.bn_act(y, bn, st, relu, residual=residual, groups=G)
    y = conv2d(x, weight, None, stride, padding, planar_norm=planar_norm)
    if res is not None:
        residual = batch_norm(conv2d(residual, res[0], None, res[2], 0), res[1])
    return batch_norm(y, bn, relu=relu, residual=residual)


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


class _MaxPool3x3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on NHWC tensors (dvs_maxpool3x3s2_*): byte argmax forward, gather backward."""

    @staticmethod
    def forward(ctx, x):
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        idx = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().dvs_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, C, _lib.stream()),
                   "dvs_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dy = dy if dy.is_contiguous(memory_format=torch.channels_last) else dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
        _lib.check(_lib.lib().dvs_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), B, H, W, C, _lib.stream()),
                   "dvs_maxpool3x3s2_bwd")
        return dx


def max_pool_3x3_s2(x):
    _require_gpu(x, "max_pool")
    if x.shape[1] % 4 == 0 and x.dtype == torch.float32:
        return _MaxPool3x3s2.apply(x)
    return F.max_pool2d(x, 3, 2, 1)


def upsample_nearest2x(x):
    """model/layers.py:196-199."""
    _require_gpu(x, "upsample")
    return F.interpolate(x, scale_factor=2, mode="nearest")
