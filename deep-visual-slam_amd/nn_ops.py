"""Network primitives used by ResnetEncoder / DepthNet / PoseNet.

Every primitive is GPU-only, keeps activations in channels_last (NHWC) memory and runs on the hand-written gfx950
kernels of libdvslam_hip.so.  There is no library (MIOpen / ATen) convolution, BatchNorm, pooling or padding path in
the product: a shape the kernels do not cover raises `DvsError` instead of silently leaving the MI355X-native path
(the A/B composition against PyTorch-ROCm's library ops lives in tools/miopen_compose.py, outside the package).
"""
import os

import torch

from . import _lib
from . import conv as _conv

CL = torch.channels_last


def _require_gpu(x, who):
    if not x.is_cuda:
        raise _lib.DvsError("%s: GPU tensors only (got %s); this package has no CPU path" % (who, x.device))
    _lib.lib()


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


def conv2d(x, weight, bias=None, stride=1, padding=0, reflect_pad=0, act=None, x2=None, upsample=False,
           planar_norm=None, passthrough=False):
    """act(conv(pad(input), weight) + bias) where input is x, upsample2x(x) (upsample=True) or
    cat([upsample2x(x), x2], 1) (x2 given); reflect_pad=1 puts ReflectionPad2d(1) in front
    (model/layers.py:121-136); planar_norm=(scale, shift) is the encoder's conv1 on the raw planar image
    with (x - 0.45) / 0.225 folded in (model/resnet_encoder.py:102-103).
    passthrough: returns (y, x') -- x' is x, on the training path as a second output of this convolution's autograd node, so
    that the gradient of x's NEXT consumer is added inside this convolution's data-gradient kernel."""
    _require_gpu(x, "conv2d")
    planar = planar_norm is not None
    if passthrough and not (_PASSTHROUGH and torch.is_grad_enabled() and x.requires_grad and not _lib.deterministic()
                            and x2 is None and not upsample and not planar):
        return conv2d(x, weight, bias, stride, padding, reflect_pad, act, x2, upsample, planar_norm), x
    if _conv.supported(x, weight, x2, planar, upsample):
        return _conv.conv2d(x, weight, bias, stride, padding, reflect_pad, act, x2=x2, upsample=upsample,
                            planar_norm=planar_norm, passthrough=passthrough)
    if _conv.head_supported(x, weight, stride, padding, reflect_pad, x2, upsample, planar):
        return _conv.head_conv2d(x, weight, bias, padding, reflect_pad, act, passthrough=passthrough)
    raise _lib.DvsError("conv2d: no gfx950 kernel for weight %s on input %s (stride %d, pad %d, reflect %d, concat %s, "
                        "upsample %s): channel counts must be multiples of 4 (or a 1/2/6/8-channel stride-1 head)"
                        % (tuple(weight.shape), tuple(x.shape), stride, padding, reflect_pad, x2 is not None, upsample))


# Folded-BatchNorm cache.  The fold depends on five tensors (conv weight, gamma, beta, running_mean, running_var); the
# repo's own writers update them through raw pointers (dvs_adam_step on the flat arena, dvs_bn_fwd / dvs_bn_finalize on
# the running statistics), which torch's `_version` counters never see -- so every such writer bumps `_generation`
# (dp.FusedAdam.step, bn.bn_act / _finalize_groups in training mode) and the cache key includes it.
_fold_cache = {}
_generation = 0


def bump_generation():
    """Called by every code path that modifies parameters or BatchNorm buffers behind torch's back."""
    global _generation
    _generation += 1


def generation():
    return _generation


def folded_bn(weight, bn):
    """(w', b') with an eval-mode BatchNorm2d folded into the convolution in front of it:
    bn(conv(x, w)) = conv(x, w * s) + (beta - running_mean * s),  s = gamma / sqrt(running_var + eps).
    Cached per weight tensor and refreshed when any of the five tensors is modified -- in place through torch
    (`_version`: torch optimisers, load_state_dict) or through the library's raw-pointer writers (`_generation`) --
    so an inference loop pays for the fold once."""
    gamma = bn.weight if bn.weight is not None else torch.ones_like(bn.running_var)
    beta = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_var)
    key = id(weight)
    ver = (weight._version, gamma._version, beta._version, bn.running_mean._version, bn.running_var._version,
           weight.data_ptr(), bn.running_mean.data_ptr(), bn.eps, _generation)
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


def _eval_affine(bn):
    """Eval-mode BatchNorm as the per-channel affine map it is: (scale, shift) as differentiable functions of gamma / beta
    (tiny [C] tensor arithmetic; the [M,C] passes are dvs_bn_* kernels, bn.affine_act)."""
    if bn.running_mean is None or not bn.track_running_stats:
        raise _lib.DvsError("BatchNorm without running statistics has no eval-mode form")
    inv = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight * inv if bn.weight is not None else inv
    shift = -bn.running_mean * scale
    if bn.bias is not None:
        shift = shift + bn.bias
    return scale, shift


_PASSTHROUGH = os.environ.get("DVS_SKIP_PASSTHROUGH", "1") != "0"


def passthrough_active(x, bn):
    """The training path on which a tensor with several consumers is handed from one consumer's autograd node to the next
    (x' = x as an extra output), so that their gradients are added inside the data-gradient kernels."""
    return (_PASSTHROUGH and torch.is_grad_enabled() and x.requires_grad and bn.training and not inference_mode(bn)
            and not _lib.deterministic())


def conv_bn_relu_with_identity(x, weight, bn, stride=1, padding=0):
    """(relu(bn(conv(x))), x') for the first half of a BasicBlock: x' is x, handed back as a second output of the convolution's
    autograd node.  The block takes its identity (or its 1x1 downsample branch) from x' (not x), so the skip path's gradient
    arrives in that node's backward and is added in the data-gradient kernel's epilogue -- otherwise autograd sums the two
    gradients of x with one more pass over the tensor (26 such passes, 0.64 ms, per VO step for the identity blocks alone).
    Outside the training path it is (conv_bn_act, x)."""
    from . import bn as _bn
    if not (passthrough_active(x, bn) and _conv.supported(x, weight, None, False) and _bn.supported_c(weight.shape[0], bn)):
        return conv_bn_act(x, weight, bn, stride, padding, relu=True), x
    _require_gpu(x, "conv_bn_act")
    G = _batch_groups
    y, st, xa = _conv.conv2d(x, weight, None, stride, padding, want_stats=G | _conv.STATS_SLOTTED, passthrough=True)
    return _bn.bn_act(y, bn, st, True, groups=G), xa


def conv_bn_act(x, weight, bn, stride=1, padding=0, relu=True, residual=None, res=None, planar_norm=None, res_passthrough=False):
    """relu(bn(conv(x)) + residual') -- the conv -> BatchNorm2d -> (+identity) -> ReLU groups of torchvision's
    BasicBlock / stem (model/resnet_encoder.py:100-111).  `residual` is the identity tensor; `res` =
    (weight, bn, stride) describes the 1x1 downsample branch conv -> bn applied to `residual` instead.
    Training: the conv epilogue accumulates the batch statistics, BN + add + ReLU is one fused pass.
    eval() + no_grad: BatchNorm folded into the weights, one kernel per conv.
    eval() with autograd (validation loss without no_grad, frozen-BN fine-tuning): BatchNorm is a per-channel affine
    map applied by the same BN kernels with fixed statistics; batch grouping is irrelevant there.
    res_passthrough (with `res`): returns (z, residual') -- `residual` handed on as a second output of the downsample
    convolution's autograd node (for one more consumer of the block input: DepthNet's skip connections)."""
    if res_passthrough:
        if res is not None and passthrough_active(residual, bn) and not _lib.deterministic():
            return _conv_bn_act_ds_passthrough(x, weight, bn, stride, padding, relu, residual, res)
        return conv_bn_act(x, weight, bn, stride, padding, relu, residual, res, planar_norm), residual
    _require_gpu(x, "conv_bn_act")
    from . import bn as _bn
    planar = planar_norm is not None
    if not _conv.supported(x, weight, None, planar) or (res is not None and not _conv.supported(residual, res[0])):
        raise _lib.DvsError("conv_bn_act: no gfx950 kernel for weight %s on input %s" % (tuple(weight.shape), tuple(x.shape)))
    if inference_mode(bn) and (res is None or inference_mode(res[1])):
        # inference (vo/predict.py:20-42,63-86: .eval() + torch.no_grad()): BatchNorm folded into the weights, bias +
        # identity + ReLU in the conv epilogue -- one kernel per conv, no normalisation passes
        sc, sh = planar_norm if planar else (None, None)
        if res is not None:
            wd, bd = folded_bn(res[0], res[1])
            residual = _conv.conv2d_forward(residual, wd, bd, res[2], 0)
        w_f, b_f = folded_bn(weight, bn)
        return _conv.conv2d_forward(x, w_f, b_f, stride, padding, act="relu" if relu else None, in_scale=sc, in_shift=sh,
                                    nchw_planar=planar, residual=residual)
    if not bn.training:
        # eval mode with autograd enabled: running statistics, differentiable w.r.t. x, conv weights, gamma, beta
        y = _conv.conv2d(x, weight, None, stride, padding, planar_norm=planar_norm)
        if res is not None:
            rs, rh = _eval_affine(res[1])
            residual = _bn.affine_act(_conv.conv2d(residual, res[0], None, res[2], 0), rs, rh)
        sc, sh = _eval_affine(bn)
        return _bn.affine_act(y, sc, sh, relu=relu, residual=residual)
    if not _bn.supported_c(weight.shape[0], bn) or (res is not None and not _bn.supported_c(res[0].shape[0], res[1])):
        raise _lib.DvsError("conv_bn_act: the fused BatchNorm kernels cover affine BatchNorm2d with C % 4 == 0 and C/4 "
                            "dividing 256 (got C = %d)" % weight.shape[0])
    G = _batch_groups
    if _lib.deterministic():
        # deterministic forward: no atomic statistics epilogue -- the batch statistics come from an ordered reduction pass
        y = _conv.conv2d(x, weight, None, stride, padding, planar_norm=planar_norm)
        st = _bn.channel_stats(y.detach(), G)
        if res is not None:
            yd = _conv.conv2d(residual, res[0], None, res[2], 0)
            return _bn.bn_act(y, bn, st, relu, residual=yd, res_bn=res[1], res_stats=_bn.channel_stats(yd.detach(), G), groups=G)
        return _bn.bn_act(y, bn, st, relu, residual=residual, groups=G)
    y, st = _conv.conv2d(x, weight, None, stride, padding, planar_norm=planar_norm, want_stats=G | _conv.STATS_SLOTTED)
    if res is not None:
        yd, std = _conv.conv2d(residual, res[0], None, res[2], 0, want_stats=G | _conv.STATS_SLOTTED)
        return _bn.bn_act(y, bn, st, relu, residual=yd, res_bn=res[1], res_stats=std, groups=G)
    return _bn.bn_act(y, bn, st, relu, residual=residual, groups=G)


def _conv_bn_act_ds_passthrough(x, weight, bn, stride, padding, relu, residual, res):
    """Training path of conv_bn_act with a downsample branch whose convolution also hands its input on (see conv_bn_act)."""
    _require_gpu(x, "conv_bn_act")
    from . import bn as _bn
    if (not _conv.supported(x, weight, None, False) or not _conv.supported(residual, res[0])
            or not _bn.supported_c(weight.shape[0], bn) or not _bn.supported_c(res[0].shape[0], res[1])):
        return conv_bn_act(x, weight, bn, stride, padding, relu, residual, res), residual
    G = _batch_groups
    # The 1x1 downsample convolution is a few dozen small workgroups (18 - 50 us that leave most of the chip idle) and depends on
    # the block input only: it runs on the network's side stream BESIDE conv2 -- and, because autograd replays a node on the stream of
    # its forward, its data gradient runs beside conv2's in the backward pass too (DVS_DS_STREAM=0: in line).
    from . import gradsink
    side = gradsink.active().side_stream(queue=False) if _DS_STREAM else None
    if side is None:
        y, st = _conv.conv2d(x, weight, None, stride, padding, want_stats=G | _conv.STATS_SLOTTED)
        yd, std, ra = _conv.conv2d(residual, res[0], None, res[2], 0, want_stats=G | _conv.STATS_SLOTTED, passthrough=True)
        return _bn.bn_act(y, bn, st, relu, residual=yd, res_bn=res[1], res_stats=std, groups=G), ra
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)                                  # the block input, the weights, the zero-filled statistics scratch
    with torch.cuda.stream(side):
        yd, std, ra = _conv.conv2d(residual, res[0], None, res[2], 0, want_stats=G | _conv.STATS_SLOTTED, passthrough=True)
    residual.record_stream(side)
    y, st = _conv.conv2d(x, weight, None, stride, padding, want_stats=G | _conv.STATS_SLOTTED)
    cur.wait_stream(side)
    for t in (yd, std):
        if isinstance(t, torch.Tensor):
            t.record_stream(cur)                           # allocated under the side stream, consumed here
    return _bn.bn_act(y, bn, st, relu, residual=yd, res_bn=res[1], res_stats=std, groups=G), ra


_DS_STREAM = os.environ.get("DVS_DS_STREAM", "0") == "1"      # measured: 24.35-24.43 ms/step with it against 24.18-24.23 without (profiles/r03_b_ds_stream_ab.txt): off
_STEM_TAIL = os.environ.get("DVS_STEM_TAIL", "1") != "0"


def stem_conv_bn_relu_pool(x, weight, bn, planar_norm, need_z=True):
    """(z, maxpool(z)) with z = relu(bn1(conv1((x - 0.45) / 0.225))) -- model/resnet_encoder.py:102-104.  Training: BatchNorm, ReLU
    and the pool are ONE pass over conv1's output (bn.bn_relu_pool) and z is only materialised when `need_z` (DepthNet's finest
    skip connection; PoseNet reads the last feature only, its z is returned as None).  Otherwise the separate operators."""
    _require_gpu(x, "stem")
    from . import bn as _bn
    if not (_STEM_TAIL and bn.training and not inference_mode(bn) and not _lib.deterministic()
            and _conv.supported(x, weight, None, True) and _bn.supported_c(weight.shape[0], bn)):
        z = conv_bn_act(x, weight, bn, 2, 3, relu=True, planar_norm=planar_norm)
        p, z = max_pool_3x3_s2(z, passthrough=True)
        return z, p
    G = _batch_groups
    y, st = _conv.conv2d(x, weight, None, 2, 3, planar_norm=planar_norm, want_stats=G | _conv.STATS_SLOTTED)
    return _bn.bn_relu_pool(y, bn, st, groups=G, need_z=need_z)


class _MaxPool3x3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on NHWC tensors (dvs_maxpool3x3s2_*): byte argmax forward, gather backward."""

    @staticmethod
    def forward(ctx, x, passthrough=False):
        x_in = x
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        idx = torch.empty((B, Ho, Wo, C), device=x.device, dtype=torch.uint8)
        _lib.check(_lib.lib().dvs_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, C, _lib.stream()),
                   "dvs_maxpool3x3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        if passthrough:       # x' = x as a second output: the gradient of its other consumer is added in the backward kernel
            ctx.set_materialize_grads(False)
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, dy, dxa=None):
        if dy is None:
            return dxa, None
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        cl = torch.channels_last
        dy = dy if dy.is_contiguous(memory_format=cl) else dy.contiguous(memory_format=cl)
        if dxa is not None:
            dxa = dxa if dxa.is_contiguous(memory_format=cl) else dxa.contiguous(memory_format=cl)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32, memory_format=cl)
        _lib.check(_lib.lib().dvs_maxpool3x3s2_bwd_res(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(),
                                                       dxa.data_ptr() if dxa is not None else None, B, H, W, C, _lib.stream()),
                   "dvs_maxpool3x3s2_bwd")
        return dx, None


def max_pool_3x3_s2(x, passthrough=False):
    """passthrough: returns (y, x') with x' = x as a second output of the pool's autograd node."""
    _require_gpu(x, "max_pool")
    if x.shape[1] % 4 or x.dtype != torch.float32:
        raise _lib.DvsError("max_pool_3x3_s2: fp32 tensors with a multiple of 4 channels only (got %s %s)" % (x.dtype, tuple(x.shape)))
    if passthrough and _PASSTHROUGH and torch.is_grad_enabled() and x.requires_grad and not _lib.deterministic():
        return _MaxPool3x3s2.apply(x, True)
    y = _MaxPool3x3s2.apply(x)
    return (y, x) if passthrough else y


class _Upsample2x(torch.autograd.Function):
    """Nearest-neighbour 2x upsampling of an NHWC tensor (dvs_upsample2x_fwd / _bwd; the backward is the 2x2 block sum)."""

    @staticmethod
    def forward(ctx, x):
        x = x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32, memory_format=CL)
        _lib.check(_lib.lib().dvs_upsample2x_fwd(x.data_ptr(), y.data_ptr(), B, H, W, C, _lib.stream()), "dvs_upsample2x_fwd")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        dy = dy if dy.is_contiguous(memory_format=CL) else dy.contiguous(memory_format=CL)
        dx = torch.empty((B, C, H, W), device=dy.device, dtype=torch.float32, memory_format=CL)
        _lib.check(_lib.lib().dvs_upsample2x_bwd(dy.data_ptr(), dx.data_ptr(), B, H, W, C, _lib.stream()), "dvs_upsample2x_bwd")
        return dx


def upsample_nearest2x(x):
    """model/layers.py:196-199 as a standalone operator (the decoder itself gathers the upsampled tensor inside its
    convolution and never materialises it)."""
    _require_gpu(x, "upsample")
    if x.shape[1] % 4 or x.dtype != torch.float32:
        raise _lib.DvsError("upsample: fp32 tensors with a multiple of 4 channels only (got %s %s)" % (x.dtype, tuple(x.shape)))
    return _Upsample2x.apply(x)
