"""Hand-written gfx950 convolution engine: autograd wrappers over dvs_conv2d_* of libdvslam_hip.so.

Tensors keep the reference's logical NCHW shapes but live in torch's channels_last memory format
(NHWC in HBM), weights likewise ([Cout][kh][kw][Cin] in HBM); nothing is transposed on the way in or
out of a kernel.  One call covers what the reference spells as several modules:

    ReflectionPad2d(1) -> Conv2d -> ELU                          (model/layers.py:106-136)
    upsample(x) ; cat([x, skip], 1) -> ConvBlock                 (model/depthnet.py:79-88)
    (x - 0.45) / 0.225 -> conv1                                  (model/resnet_encoder.py:102-103)
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib, gradsink, zeropool
from ._lib import ConvDesc, ConvFusion, check, ptr

ACT = {None: 0, "none": 0, "relu": 1, "elu": 2, "sigmoid": 3, "gelu": 4}
CL = torch.channels_last


def _has_grad(t):
    """A leaf tensor with an existing .grad (autograd will then add the returned gradient in place and never adopt it, so it
    may live in the step-scoped scratch pool); non-leaf weights (gamma * W, reshaped ConvTranspose weights) never qualify."""
    return t is not None and t.is_leaf and t.grad is not None


def _nhwc(t):
    """Contiguous-NHWC requirement (a [B,1,H,W] or [B,C,1,1] tensor is both NCHW and NHWC)."""
    if t.dtype != torch.float32:
        raise _lib.DvsError("fp32 tensors only (got %s)" % t.dtype)
    return t if t.is_contiguous(memory_format=CL) else t.contiguous(memory_format=CL)


def _desc(B, Cin, H, W, w_shape, stride, pad, reflect):
    Cout, _, kh, kw = w_shape
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, Cout
    d.kh, d.kw, d.stride, d.pad, d.pad_mode = kh, kw, stride, pad, int(bool(reflect))
    return d


def out_hw(H, W, kh, kw, stride, pad):
    return (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1


UPSAMPLE_ONLY = "upsample-only"   # x2 marker: the logical input is upsample2x(x) with nothing concatenated


def _geometry(x, w_shape, x2, nchw_planar):
    """(x, x2, B, Cin, H, W) with layouts normalised for the kernels."""
    if nchw_planar:
        x = x if x.is_contiguous() else x.contiguous()
        B, Cin, H, W = x.shape
    else:
        x = _nhwc(x)
        B, Cin, H, W = x.shape
        if x2 is UPSAMPLE_ONLY:
            H, W = 2 * H, 2 * W
        elif x2 is not None:
            x2 = _nhwc(x2)
            H, W = x2.shape[2], x2.shape[3]
            if (x.shape[2] * 2, x.shape[3] * 2) != (H, W):
                raise _lib.DvsError("upsample+concat fusion: skip tensor must be exactly 2x the coarse one")
            Cin = x.shape[1] + x2.shape[1]
    if w_shape[1] != Cin:
        raise _lib.DvsError("weight expects %d input channels, got %d" % (w_shape[1], Cin))
    return x, x2, B, Cin, H, W


def _fusion(x, x2, in_scale, in_shift, in_relu, nchw_planar, act=None, stats=None, stat_groups=1, residual=None, stat_slots=1):
    f = ConvFusion()
    if residual is not None:
        f.residual = residual.data_ptr()      # NHWC memory, checked by the caller
    if x2 is UPSAMPLE_ONLY:
        f.x2, f.C1 = x.data_ptr(), x.shape[1]      # C1 == Cin: the second source is never read
    elif x2 is not None:
        f.x2, f.C1 = x2.data_ptr(), x.shape[1]
    if in_scale is not None:
        f.in_scale, f.in_shift, f.in_relu = ptr(in_scale), ptr(in_shift), int(bool(in_relu))
    f.nchw_planar = int(bool(nchw_planar))
    f.act = ACT[act]
    if stats is not None:
        f.stats = ptr(stats)
        f.stat_groups = int(stat_groups)
        f.stat_slots = int(stat_slots)
    return f


def _pack_planar_weight(weight):
    """Encoder conv1: [Cout][Cin][kh][8] with kw zero-padded to 8 (K order (ci,ky,kx))."""
    return torch.nn.functional.pad(weight.contiguous(), (0, 8 - weight.shape[3]))


def conv2d_forward(x, weight, bias=None, stride=1, pad=0, reflect=False, act=None, x2=None, in_scale=None,
                   in_shift=None, in_relu=False, nchw_planar=False, stats=None, stat_groups=1, residual=None, stat_slots=1):
    """Raw forward launch (no autograd).  x: logical [B,Cin,H,W] (NHWC memory, or planar NCHW when
    nchw_planar); with x2 the logical input is cat([upsample2x(x), x2], 1); residual (inference only): a tensor of
    the output's shape added before the activation."""
    if not (x.is_cuda and weight.is_cuda):
        raise _lib.DvsError("conv2d: GPU tensors only; this package has no CPU path")
    x, x2, B, Cin, H, W = _geometry(x, tuple(weight.shape), x2, nchw_planar)
    w = _pack_planar_weight(weight) if nchw_planar else _nhwc(weight)
    d = _desc(B, Cin, H, W, weight.shape, stride, pad, reflect)
    Ho, Wo = out_hw(H, W, d.kh, d.kw, stride, pad)
    y = torch.empty((B, d.Cout, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    if residual is not None:
        residual = _nhwc(residual)
        if tuple(residual.shape) != tuple(y.shape):
            raise _lib.DvsError("conv2d_forward: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    f = _fusion(x, x2, in_scale, in_shift, in_relu, nchw_planar, act, stats, stat_groups, residual, stat_slots)
    check(_lib.lib().dvs_conv2d_fwd(x.data_ptr(), w.data_ptr(), ptr(bias), y.data_ptr(), C.byref(d), C.byref(f),
                                    _lib.stream()), "dvs_conv2d_fwd")
    return y


_prepacked = {}     # weight.data_ptr() -> (packed tensor, weight._version it was packed at, shape, weakref to the weight)


def conv2d_dgrad_padded(dz, weight, x_shape, split_c1=0, wino=False, p16=False, y_out=None, act=None):
    """Data gradient of ReflectionPad2d(1) + 3x3 stride-1 conv when dz is already the PRE-activation gradient: the
    gradient w.r.t. the padded input is a plain zero-padded correlation (the LDS-DMA kernel; no fold, no activation
    derivative in its gather), dvs_reflect_fold then folds the border back and splits / 2x2-sums for an upsample(+concat)
    input.  Returns dx, or (d coarse, d skip) as conv2d_dgrad does."""
    l = _lib.lib()
    B, Cin, H, W = x_shape
    if p16:         # bf16 mode: the full correlation on the patch kernel
        # (y_out / act: dz is still dY, the thin kernel multiplies by act'(y_out) as it stages)
        g = conv3x3_p16_gen(dz, None, weight, reflect=False, full=True, flip=True, dact_y=y_out, dact=act)
    elif wino:      # the same full correlation on the Winograd kernel (rotated / transposed filter operand)
        g = conv3x3_wino_gen(dz, None, weight, reflect=False, full=True, flip=True)
    else:
        g = conv2d_dgrad(dz, weight, (B, Cin, H + 2, W + 2), 1, 0, False, prepadded=True)    # [B, Cin, H+2, W+2] (NHWC memory)
    if split_c1:
        dx = torch.empty((B, H // 2, W // 2, split_c1), device=dz.device, dtype=torch.float32).permute(0, 3, 1, 2)
        dskip = (torch.empty((B, Cin - split_c1, H, W), device=dz.device, dtype=torch.float32, memory_format=CL)
                 if split_c1 < Cin else None)
        check(l.dvs_reflect_fold(g.data_ptr(), dx.data_ptr(), dskip.data_ptr() if dskip is not None else None, B, H, W, Cin,
                                 split_c1, _lib.stream()), "dvs_reflect_fold")
        return dx, dskip
    dx = torch.empty((B, Cin, H, W), device=dz.device, dtype=torch.float32, memory_format=CL)
    check(l.dvs_reflect_fold(g.data_ptr(), dx.data_ptr(), None, B, H, W, Cin, 0, _lib.stream()), "dvs_reflect_fold")
    return dx


class PackedWeights:
    """Data-gradient weight packs ([Cin][kh][kw][Cout]) of a fixed set of convolution weights, refreshed with ONE launch.
    The weights change once per optimiser step, so dp.FusedAdam owns one of these for its arena and calls repack()
    right after the Adam kernel; conv2d_dgrad then finds the pack here instead of transposing inside the backward pass
    (51 launches per step).  A pack is trusted only while the weight's torch version counter is the one it was packed
    at -- any in-place torch update (load_state_dict, torch.optim) falls back to packing on the fly."""

    def __init__(self, weights):
        self.weights = [w for w in weights if w.dim() == 4 and w.is_cuda and w.shape[0] % 4 == 0 and w.shape[1] % 4 == 0
                        and w.permute(0, 2, 3, 1).is_contiguous()]
        self.packs, rows, wg = [], [], 0
        for w in self.weights:
            co, ci, kh, kw = w.shape
            wt = torch.empty(ci * kh * kw * co, device=w.device, dtype=torch.float32)
            self.packs.append(wt)
            rows.append([w.data_ptr(), wt.data_ptr(), co | (ci << 32), (kh * kw) | (wg << 32)])
            wg += kh * kw * ((ci + 31) // 32) * ((co + 31) // 32)
        self.total_wgs = wg
        self.table = torch.tensor(rows, dtype=torch.int64, device=self.weights[0].device) if rows else None
        # Winograd operands (both orientations) of the stride-1 3x3 weights that qualify by shape
        self.wino = [w for w in self.weights if wino_eligible(w, 1, 1, False, None, None, False, None)]
        self.wino_ops, rows, wg = [], [], 0
        for w in self.wino:
            co, ci = w.shape[:2]
            u, uf = _wino_alloc(w), _wino_alloc(w)
            self.wino_ops.append((u, uf))
            rows.append([w.data_ptr(), u.data_ptr(), uf.data_ptr(), co | (ci << 32), wg])       # csrc/conv_wino.hip WinoEntry
            wg += (co * ci + 255) // 256
        self.wino_wgs = wg
        self.wino_table = torch.tensor(rows, dtype=torch.int64, device=self.weights[0].device) if rows else None

    def repack(self):
        if self.table is None:
            return
        check(_lib.lib().dvs_conv2d_pack_wt_batch(self.table.data_ptr(), len(self.weights), self.total_wgs, _lib.stream()),
              "dvs_conv2d_pack_wt_batch")
        for w, wt in zip(self.weights, self.packs):
            # the weak reference pins the entry to THIS tensor object: the caching allocator hands the address of a freed
            # arena to the next one, and a pack of the old network must never serve the new network's data gradient
            _prepacked[w.data_ptr()] = (wt, w._version, tuple(w.shape), weakref.ref(w))
        if self.wino_table is not None:
            check(_lib.lib().dvs_wino_weights_batch(self.wino_table.data_ptr(), len(self.wino), self.wino_wgs, _lib.stream()),
                  "dvs_wino_weights_batch")
            for w, (u, uf) in zip(self.wino, self.wino_ops):
                _wino_packed[w.data_ptr()] = [u, uf, w._version, tuple(w.shape), weakref.ref(w)]

    def release(self):
        """Drop this pack's entries -- only its own: a late-collected owner (trainer / optimiser __del__) must not evict what a
        newer owner has registered at the same, reused arena address (its entries are pinned to ITS weight objects)."""
        for w in self.weights:
            for table in (_prepacked, _wino_packed):
                ent = table.get(w.data_ptr())
                if ent is not None and ent[-1]() is w:
                    table.pop(w.data_ptr(), None)


def _packed_weight(w, weight):
    """[Cin][kh][kw][Cout] operand of the data gradient: the optimiser's pre-packed copy when it is current, else packed here."""
    ent = _prepacked.get(w.data_ptr())
    if ent is not None and ent[3]() is weight and ent[1] == weight._version and ent[2] == tuple(weight.shape):
        return ent[0]
    Cout, Cin, kh, kw = weight.shape
    wt = torch.empty(Cin * kh * kw * Cout, device=w.device, dtype=torch.float32)
    check(_lib.lib().dvs_conv2d_pack_wt(w.data_ptr(), wt.data_ptr(), Cout, Cin, kh, kw, _lib.stream()), "dvs_conv2d_pack_wt")
    return wt


# ---- Winograd F(2x2, 3x3) path of the stride-1 3x3 BasicBlock convolutions (csrc/conv_wino.hip) ------------------------
_WINO = os.environ.get("DVS_WINOGRAD", "1") != "0"
_WINO_FORCE = os.environ.get("DVS_WINOGRAD", "1") == "force"       # take the Winograd kernels whatever the cost model says (tests)


def _wino_on():
    """The Winograd kernels are fp32 kernels: in the bf16 mode (_lib.set_precision) the direct kernels on the bf16 matrix cores
    are faster than 2.25x fewer fp32 multiplies, so every 3x3 layer takes those."""
    return _WINO and _lib._precision != "bf16"


_wino_packed = {}   # weight.data_ptr() -> [u, u_flip, weight._version, shape, weakref]  (either operand may be None)


def wino_eligible(weight, stride, pad, reflect, act, x2, planar, scale):
    """Forward AND data gradient of this convolution run on the Winograd kernel: 3x3, stride 1, zero pad 1, no fused input
    transform or activation, both channel counts multiples of 16 (the kernels' K chunking) and wide enough to fill the
    32-channel MFMA columns."""
    co, ci, kh, kw = weight.shape
    return (_wino_on() and kh == 3 and kw == 3 and stride == 1 and pad == 1 and not reflect and act is None and x2 is None
            and not planar and scale is None and ci % 16 == 0 and co % 16 == 0 and ci >= 64 and co >= 64)


def wino_pays(B, H, W, k, n):
    """Cost model for ONE launch (microseconds; fitted to tools/wino_bench.py at batch 2 ... 24): the Winograd kernel runs one
    workgroup per CU whose duration is set by the reduction length alone, so it only wins when there are enough tiles --
    layer 4 at batch 2 has 20 workgroups of 132 us against 42 us for the direct kernel with its small tiles and split K."""
    if _WINO_FORCE:
        return True
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    mt, nc = (64, 64) if n <= 64 else (32, 128)
    wgs = -(-tiles // mt) * -(-n // nc)
    t_wino = -(-wgs // 256) * (7.0 + 0.245 * k)
    t_direct = 25.0 + 2.0 * B * H * W * n * k * 9 / 120e6
    return t_wino < 0.9 * t_direct


def _wino_alloc(weight):
    co, ci = weight.shape[:2]
    return torch.empty(co * ci * 16, device=weight.device, dtype=torch.float32)


def _wino_weight(w, weight, flip):
    """G g G^T operand of the Winograd kernel ([K][4][N][4]; flip: the data gradient's rotated / transposed filter): the
    optimiser's pre-transformed copy when it is current, else transformed here and kept until the weight changes."""
    # keyed by the WEIGHT's own address (w may be a temporary NHWC copy whose address the allocator hands out again) and
    # pinned to the weight OBJECT: a live parameter whose storage moved (an arena, .to()) leaves its old address to others
    ent = _wino_packed.get(weight.data_ptr())
    if not (ent is not None and ent[4]() is weight and ent[2] == weight._version and ent[3] == tuple(weight.shape)):
        if len(_wino_packed) > 1024:
            for k in [k for k, e in _wino_packed.items() if e[4]() is None]:
                del _wino_packed[k]
        ent = [None, None, weight._version, tuple(weight.shape), weakref.ref(weight)]
        _wino_packed[weight.data_ptr()] = ent
    if ent[int(flip)] is None:
        u = _wino_alloc(weight)
        check(_lib.lib().dvs_wino_weights(w.data_ptr(), u.data_ptr(), weight.shape[0], weight.shape[1], int(flip), _lib.stream()),
              "dvs_wino_weights")
        ent[int(flip)] = u
    return ent[int(flip)]


# ---- bf16 mode: the same layers on the patch kernel (csrc/conv_p16.hip) -------------------------------------------------------
_P16 = os.environ.get("DVS_BF16_PATCH", "1") != "0"
_p16_packed = {}    # weight.data_ptr() -> [pack, pack_flip, weight._version, shape, weakref]


def p16_eligible(weight, stride, pad, reflect, act, x2, planar, scale):
    """bf16 mode only: forward AND data gradient of this convolution run on the patch kernel -- 3x3, stride 1, zero pad 1, no
    fused input transform / bias / activation, channel counts multiples of 64."""
    co, ci, kh, kw = weight.shape
    return (_P16 and _lib._precision == "bf16" and kh == 3 and kw == 3 and stride == 1 and pad == 1 and not reflect and act is None
            and x2 is None and not planar and scale is None and ci % 64 == 0 and co % 64 == 0)


def _p16_weight(w, weight, flip):
    """bf16 [9][K/16][N][16] operand of the patch kernel (flip: the data gradient's), kept until the weight changes (same keying
    as _wino_weight)."""
    # (torch's version counter AND the package's generation counter: dp.FusedAdam and the other raw-pointer writers change the
    # weights behind torch's back and bump the latter -- nn_ops.bump_generation -- so a pack never outlives an optimiser step)
    from . import nn_ops
    stamp = (weight._version, nn_ops.generation())
    ent = _p16_packed.get(weight.data_ptr())
    if not (ent is not None and ent[4]() is weight and ent[2] == stamp and ent[3] == tuple(weight.shape)):
        if len(_p16_packed) > 1024:
            for k in [k for k, e in _p16_packed.items() if e[4]() is None]:
                del _p16_packed[k]
        ent = [None, None, stamp, tuple(weight.shape), weakref.ref(weight)]
        _p16_packed[weight.data_ptr()] = ent
    if ent[int(flip)] is None:
        k, n = (weight.shape[0], weight.shape[1]) if flip else (weight.shape[1], weight.shape[0])
        u = torch.empty(9 * k * ((n + 31) // 32 * 32), device=weight.device, dtype=torch.bfloat16)      # (N padded to 32 columns)
        check(_lib.lib().dvs_conv3x3_bf16_pack(w.data_ptr(), u.data_ptr(), weight.shape[0], weight.shape[1], int(flip), _lib.stream()),
              "dvs_conv3x3_bf16_pack")
        ent[int(flip)] = u
    return ent[int(flip)]


def conv3x3_p16(x, weight, stats=None, stat_groups=0, flip=False, residual=None, stat_slots=1):
    """y = conv3x3(x, weight) (stride 1, zero pad 1) with bf16 operands on the patch kernel; flip: the data gradient of that
    convolution, x = dY [B,Cout,H,W] -> dX [B,Cin,H,W] (+ residual).  `weight` must be the parameter object itself."""
    x, w = _nhwc(x), _nhwc(weight)
    co, ci = weight.shape[:2]
    k, n = (co, ci) if flip else (ci, co)
    B, cx, H, W = x.shape
    if cx != k:
        raise _lib.DvsError("conv3x3_p16: input has %d channels, the operand expects %d" % (cx, k))
    u = _p16_weight(w, weight, flip)
    y = torch.empty((B, n, H, W), device=x.device, dtype=torch.float32, memory_format=CL)
    if residual is not None:
        residual = _nhwc(residual)
        if tuple(residual.shape) != tuple(y.shape):
            raise _lib.DvsError("conv3x3_p16: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    check(_lib.lib().dvs_conv3x3_bf16_fwd(x.data_ptr(), u.data_ptr(), residual.data_ptr() if residual is not None else None, y.data_ptr(),
                                          ptr(stats), stat_groups if stats is not None else 0, int(stat_slots), B, H, W, k, n, int(flip),
                                          _lib.stream()), "dvs_conv3x3_bf16_fwd")
    return y


def p16_dec_eligible(weight, stride, pad, reflect, act, x, x2, planar, scale):
    """bf16 mode only: the decoder's Conv3x3 layers (ReflectionPad2d(1) + 3x3 [+ ELU], optionally nearest-2x upsample (+ concat) in
    the gather) on the patch kernels: 64-channel chunks for the wide levels, the thin kernel (32 output channels per workgroup,
    chunks of 32 / 16) for the 32- and 16-channel ones."""
    co, ci, kh, kw = weight.shape
    if not (_P16 and _lib._precision == "bf16" and kh == 3 and kw == 3 and stride == 1 and pad == 1 and reflect and act in (None, "elu")
            and not planar and scale is None and co % 16 == 0):
        return False
    c1 = x.shape[1]
    if x2 is None:
        return c1 == ci and c1 % 16 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2
    if x2 is UPSAMPLE_ONLY:
        return c1 == ci and c1 % 16 == 0
    return c1 % 16 == 0 and x2.shape[1] % 16 == 0 and c1 + x2.shape[1] == ci


def conv3x3_p16_gen(x, x2, weight, bias=None, act=None, reflect=True, full=False, flip=False, dact_y=None, dact=None):
    """Patch kernel with the general gather (bf16 operands).  x2: None, UPSAMPLE_ONLY or the skip tensor (x is then the
    half-resolution operand).  full: zero-padded full correlation, output [B, N, H+2, W+2] (with flip: the padded-domain data gradient)."""
    x, w = _nhwc(x), _nhwc(weight)
    co, ci = weight.shape[:2]
    k, n = (co, ci) if flip else (ci, co)
    up = x2 is not None
    skip = _nhwc(x2) if isinstance(x2, torch.Tensor) else None
    B, c1, hs, ws = x.shape
    H, W = (2 * hs, 2 * ws) if up else (hs, ws)
    c2 = skip.shape[1] if skip is not None else 0
    if c1 + c2 != k or (skip is not None and tuple(skip.shape) != (B, c2, H, W)):
        raise _lib.DvsError("conv3x3_p16_gen: operands %s / %s do not make the %d input channels at %dx%d"
                            % (tuple(x.shape), None if skip is None else tuple(skip.shape), k, H, W))
    u = _p16_weight(w, weight, flip)
    Ho, Wo, org = (H + 2, W + 2, 2) if full else (H, W, 1)
    y = torch.empty((B, n, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    check(_lib.lib().dvs_conv3x3_bf16_gen(x.data_ptr(), skip.data_ptr() if skip is not None else None, u.data_ptr(), ptr(bias), y.data_ptr(),
                                          B, H, W, c1, c2, n, Ho, Wo, org, int(up), int(bool(reflect)), ACT[act], int(flip),
                                          _nhwc(dact_y).data_ptr() if dact_y is not None else None, ACT[dact] if dact_y is not None else 0,
                                          _lib.stream()),
          "dvs_conv3x3_bf16_gen")
    return y


_P16_WGS = int(os.environ.get("DVS_BF16_WGRAD_WGS", "0"))        # workgroups of the bf16 weight gradient (0: the library's default)


def conv3x3_p16_wgrad(x, dy, weight_shape, dw_out=None, pooled=False):
    """Weight gradient of the stride-1 / pad-1 3x3 convolution with bf16 operands (csrc/conv_p16.hip).  dw_out: gradient sink to
    add into (returns None), else a zero-filled [Cout][kh][kw][Cin]-stored tensor of the weight's shape is returned."""
    x, dy = _nhwc(x), _nhwc(dy)
    co, ci = weight_shape[:2]
    B, _, H, W = x.shape
    if dw_out is not None:
        if tuple(dw_out.shape) != tuple(weight_shape) or not dw_out.permute(0, 2, 3, 1).is_contiguous():
            raise _lib.DvsError("conv3x3_p16_wgrad: gradient sink must be a [Cout][kh][kw][Cin]-stored tensor of the weight's shape")
        dw = dw_out
    else:
        dw = zeropool.zeros(tuple(weight_shape), dy.device, channels_last=True, pooled=pooled)
    check(_lib.lib().dvs_conv3x3_bf16_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ci, co, _P16_WGS, _lib.stream()),
          "dvs_conv3x3_bf16_wgrad")
    return None if dw_out is not None else dw


def conv3x3_p16_wgrad_gen(x, x2, dy, weight_shape, dw_out=None, pooled=False, y_out=None, act=None, want_bias=False, db_out=None):
    """Weight gradient of ReflectionPad2d(1) + [nearest 2x upsample of x (+ concat with x2)] + 3x3 with bf16 operands on the patch
    kernel (arguments and results as conv3x3_wino_wgrad_gen)."""
    x, dy = _nhwc(x), _nhwc(dy)
    co, ci = weight_shape[:2]
    up = x2 is not None
    skip = _nhwc(x2) if isinstance(x2, torch.Tensor) else None
    B, c1, hs, ws = x.shape
    H, W = (2 * hs, 2 * ws) if up else (hs, ws)
    c2 = skip.shape[1] if skip is not None else 0
    if c1 + c2 != ci or tuple(dy.shape) != (B, co, H, W) or (skip is not None and tuple(skip.shape) != (B, c2, H, W)):
        raise _lib.DvsError("conv3x3_p16_wgrad_gen: operands %s / %s / dy %s do not fit a %s weight at %dx%d"
                            % (tuple(x.shape), None if skip is None else tuple(skip.shape), tuple(dy.shape), tuple(weight_shape), H, W))
    dact = ACT[act]
    if (want_bias or db_out is not None) and not dact:
        raise _lib.DvsError("conv3x3_p16_wgrad_gen: the bias gradient rides on the activation-derivative path")
    if dact and (y_out is None or tuple(y_out.shape) != tuple(dy.shape)):
        raise _lib.DvsError("conv3x3_p16_wgrad_gen: the activation derivative needs the forward output")
    if dw_out is not None:
        if tuple(dw_out.shape) != tuple(weight_shape) or not dw_out.permute(0, 2, 3, 1).is_contiguous():
            raise _lib.DvsError("conv3x3_p16_wgrad_gen: gradient sink must be a [Cout][kh][kw][Cin]-stored tensor of the weight's shape")
        dw = dw_out
    else:
        dw = zeropool.zeros(tuple(weight_shape), dy.device, channels_last=True, pooled=pooled)
    db = db_out if db_out is not None else (zeropool.zeros((co,), dy.device, pooled=pooled) if want_bias else None)
    check(_lib.lib().dvs_conv3x3_bf16_wgrad_gen(x.data_ptr(), skip.data_ptr() if skip is not None else None, dy.data_ptr(),
                                                _nhwc(y_out).data_ptr() if dact else None, dw.data_ptr(), ptr(db), B, H, W, c1, c2, co,
                                                int(up), 1, dact, _P16_WGS, _lib.stream()), "dvs_conv3x3_bf16_wgrad_gen")
    return (None if dw_out is not None else dw), (None if db_out is not None else db)


STAT_SLOTS = int(os.environ.get("DVS_WINO_STAT_SLOTS", "16"))     # copies of the statistics table the Winograd forward spreads its atomics over


def conv3x3_wino(x, weight, stats=None, stat_groups=0, flip=False, residual=None, bias=None, relu=False, stat_slots=1):
    """y = [relu](conv3x3(x, weight) [+ bias]) (stride 1, zero pad 1) on the Winograd kernel; flip: the data gradient of that
    convolution, x = dY [B,Cout,H,W] -> dX [B,Cin,H,W].  `weight` must be the parameter object itself (the operand cache is pinned to it)."""
    x, w = _nhwc(x), _nhwc(weight)
    co, ci = weight.shape[:2]
    k, n = (co, ci) if flip else (ci, co)
    B, cx, H, W = x.shape
    if cx != k:
        raise _lib.DvsError("conv3x3_wino: input has %d channels, the operand expects %d" % (cx, k))
    u = _wino_weight(w, weight, flip)
    y = torch.empty((B, n, H, W), device=x.device, dtype=torch.float32, memory_format=CL)
    if residual is not None:
        residual = _nhwc(residual)
        if tuple(residual.shape) != tuple(y.shape):
            raise _lib.DvsError("conv3x3_wino: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    # stat_slots > 1: stats is [stat_slots][G][2][N]; the BatchNorm kernels add the copies up (bn.bn_act)
    check(_lib.lib().dvs_conv3x3_wino_fwd_slots(x.data_ptr(), u.data_ptr(), ptr(bias), residual.data_ptr() if residual is not None else None,
                                                y.data_ptr(), ptr(stats), stat_groups if stats is not None else 0, int(stat_slots),
                                                B, H, W, k, n, int(bool(relu)), int(flip), _lib.stream()), "dvs_conv3x3_wino_fwd")
    return y


_WINO_DEC = os.environ.get("DVS_WINOGRAD_DECODER", "1") != "0"


def wino_dec_eligible(weight, stride, pad, reflect, act, x, x2, planar, scale):
    """The decoder's wide Conv3x3 layers (ReflectionPad2d(1) + 3x3, ELU, optionally nearest-2x upsample (+ concat) in the
    gather; model/layers.py:26-41, model/depth_decoder.py:52-62) on the Winograd kernel's general gather."""
    co, ci, kh, kw = weight.shape
    if not (_wino_on() and _WINO_DEC and kh == 3 and kw == 3 and stride == 1 and pad == 1 and reflect and act in (None, "elu")
            and not planar and scale is None and ci % 16 == 0 and co % 16 == 0 and ci >= 64 and co >= 64):
        return False
    c1 = x.shape[1]
    if x2 is None:
        return c1 == ci and x.shape[2] >= 2 and x.shape[3] >= 2
    if x2 is UPSAMPLE_ONLY:
        return c1 == ci
    return c1 % 8 == 0 and c1 + x2.shape[1] == ci


def conv3x3_wino_gen(x, x2, weight, bias=None, act=None, reflect=True, full=False, flip=False):
    """Winograd kernel with the general gather.  x2: None, UPSAMPLE_ONLY or the skip tensor (x is then the half-resolution
    operand).  full: zero-padded full correlation, output [B, N, H+2, W+2] (with flip: the padded-domain data gradient)."""
    x, w = _nhwc(x), _nhwc(weight)
    co, ci = weight.shape[:2]
    k, n = (co, ci) if flip else (ci, co)
    up = x2 is not None
    skip = _nhwc(x2) if isinstance(x2, torch.Tensor) else None
    B, c1, hs, ws = x.shape
    H, W = (2 * hs, 2 * ws) if up else (hs, ws)
    c2 = skip.shape[1] if skip is not None else 0
    if c1 + c2 != k or (skip is not None and tuple(skip.shape) != (B, c2, H, W)):
        raise _lib.DvsError("conv3x3_wino_gen: operands %s / %s do not make the %d input channels at %dx%d"
                            % (tuple(x.shape), None if skip is None else tuple(skip.shape), k, H, W))
    u = _wino_weight(w, weight, flip)
    Ho, Wo, org = (H + 2, W + 2, 2) if full else (H, W, 1)
    y = torch.empty((B, n, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=CL)
    check(_lib.lib().dvs_conv3x3_wino_gen(x.data_ptr(), skip.data_ptr() if skip is not None else None, u.data_ptr(), ptr(bias), y.data_ptr(), B, H, W, c1, c2, n, Ho, Wo, org,
                                          int(up), int(bool(reflect)), ACT[act], int(flip), _lib.stream()), "dvs_conv3x3_wino_gen")
    return y


_WINO_WGRAD = os.environ.get("DVS_WINOGRAD_WGRAD", "1") != "0"
_WINO_WGS = int(os.environ.get("DVS_WINO_WGRAD_WGS", "0"))


def wino_wgrad_eligible(weight_shape, x=None):
    """32-channel blocks, and operands below 1 GiB (the kernel's 32-bit offsets carry two mask bits; x: the forward input, which
    has the output's spatial size for these stride-1 layers) -- larger ones keep the implicit-GEMM weight gradient."""
    co, ci = weight_shape[:2]
    fits = x is None or (x.numel() // x.shape[1] + x.shape[3] + 1) * max(co, ci) * 4 + 8192 < 2 ** 30
    return _wino_on() and _WINO_WGRAD and co % 32 == 0 and ci % 32 == 0 and fits


def conv3x3_wino_wgrad(x, dy, weight_shape, dw_out=None, pooled=False):
    """Weight gradient of the stride-1 / pad-1 3x3 convolution on the Winograd kernel.  dw_out: gradient sink to add into
    (returns None), else a zero-filled [Cout][kh][kw][Cin]-stored tensor of the weight's shape is returned."""
    x, dy = _nhwc(x), _nhwc(dy)
    co, ci = weight_shape[:2]
    B, _, H, W = x.shape
    if dw_out is not None:
        if tuple(dw_out.shape) != tuple(weight_shape) or not dw_out.permute(0, 2, 3, 1).is_contiguous():
            raise _lib.DvsError("conv3x3_wino_wgrad: gradient sink must be a [Cout][kh][kw][Cin]-stored tensor of the weight's shape")
        dw = dw_out
    else:
        dw = zeropool.zeros(tuple(weight_shape), dy.device, channels_last=True, pooled=pooled)
    ws, nws = _wgrad_workspace(B, H, W, ci, co, dy.device)
    check(_lib.lib().dvs_conv3x3_wino_wgrad_ws(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), B, H, W, ci, co, _WINO_WGS,
                                               ptr(ws), nws, _lib.stream()), "dvs_conv3x3_wino_wgrad")
    return None if dw_out is not None else dw


# Ordered weight gradient: DVS_WGRAD_ORDERED=1, or the deterministic mode (_lib.set_deterministic) -- the Winograd kernels store
# per-workgroup partial blocks in a workspace and a second kernel adds them in a fixed order instead of float atomics on dw.
_WGRAD_ORDERED = os.environ.get("DVS_WGRAD_ORDERED", "0") == "1"


def _wgrad_workspace(B, H, W, ci, co, device):
    if not (_WGRAD_ORDERED or _lib.deterministic()):
        return None, 0
    n = _lib.lib().dvs_conv3x3_wino_wgrad_workspace(B, H, W, ci, co, _WINO_WGS)
    return torch.empty(max(n // 4, 1), device=device, dtype=torch.float32), n


_WINO_DEC_WGRAD = os.environ.get("DVS_WINOGRAD_DECODER_WGRAD", "1") != "0"


def wino_dec_wgrad_eligible(weight_shape, x, x2):
    """The decoder's wide Conv3x3 layers (the ones wino_dec_eligible sends to the Winograd forward) whose sources split into
    32-channel blocks: their weight gradient runs on the Winograd kernel's reflect / upsample gathers."""
    co, ci = weight_shape[:2]
    c2 = x2.shape[1] if isinstance(x2, torch.Tensor) else 0
    up = 1 if x2 is None else 2
    return (_wino_on() and _WINO_WGRAD and _WINO_DEC_WGRAD and co % 32 == 0 and x.shape[1] % 32 == 0 and c2 % 32 == 0
            and x.shape[1] + c2 == ci and x.shape[2] * up >= 2 and x.shape[3] * up >= 2
            and (x.shape[0] * x.shape[2] * x.shape[3] * up * up + 3 * x.shape[3] * up + 3) * max(co, x.shape[1], c2) * 4 + 8192 < 2 ** 30)


def wino_dec_wgrad_pays(B, H, W, k, n):
    """Enough work for one round of workgroups (256 x four waves x eight k-steps of two tiles) over the (tile range, 32 x 32
    channel block) pairs: below that the split-K implicit-GEMM / row-ring kernels are faster (tools/dec_wgrad_bench.py: even at
    batch 2 every decoder layer of the 480 x 640 network is above it)."""
    return _WINO_FORCE or B * ((H + 1) // 2) * ((W + 1) // 2) * (k // 32) * (n // 32) >= 16384


def conv3x3_wino_wgrad_gen(x, x2, dy, weight_shape, dw_out=None, pooled=False, y_out=None, act=None, want_bias=False, db_out=None):
    """Weight gradient of ReflectionPad2d(1) + [nearest 2x upsample of x (+ concat with x2)] + 3x3 (model/layers.py:26-41,
    model/depth_decoder.py:52-62) on the Winograd kernel.  act None: dy is the gradient in front of the activation; "elu" /
    "relu": dy is multiplied by act'(y_out) as it is loaded and the bias gradient (want_bias, or the sink db_out) taken on the
    way.  x2: None, UPSAMPLE_ONLY or the skip tensor.  dw_out / db_out: gradient sinks to add into.  Returns (dw, db), None
    where a sink took it."""
    x, dy = _nhwc(x), _nhwc(dy)
    co, ci = weight_shape[:2]
    up = x2 is not None
    skip = _nhwc(x2) if isinstance(x2, torch.Tensor) else None
    B, c1, hs, ws = x.shape
    H, W = (2 * hs, 2 * ws) if up else (hs, ws)
    c2 = skip.shape[1] if skip is not None else 0
    if c1 + c2 != ci or tuple(dy.shape) != (B, co, H, W) or (skip is not None and tuple(skip.shape) != (B, c2, H, W)):
        raise _lib.DvsError("conv3x3_wino_wgrad_gen: operands %s / %s / dy %s do not fit a %s weight at %dx%d"
                            % (tuple(x.shape), None if skip is None else tuple(skip.shape), tuple(dy.shape), tuple(weight_shape), H, W))
    dact = ACT[act]
    if (want_bias or db_out is not None) and not dact:
        raise _lib.DvsError("conv3x3_wino_wgrad_gen: the bias gradient rides on the activation-derivative path")
    if dact and (y_out is None or tuple(y_out.shape) != tuple(dy.shape)):
        raise _lib.DvsError("conv3x3_wino_wgrad_gen: the activation derivative needs the forward output")
    if dw_out is not None:
        if tuple(dw_out.shape) != tuple(weight_shape) or not dw_out.permute(0, 2, 3, 1).is_contiguous():
            raise _lib.DvsError("conv3x3_wino_wgrad_gen: gradient sink must be a [Cout][kh][kw][Cin]-stored tensor of the weight's shape")
        dw = dw_out
    else:
        dw = zeropool.zeros(tuple(weight_shape), dy.device, channels_last=True, pooled=pooled)
    db = db_out if db_out is not None else (zeropool.zeros((co,), dy.device, pooled=pooled) if want_bias else None)
    ws, nws = _wgrad_workspace(B, H, W, max(c1, c2), co, dy.device)
    check(_lib.lib().dvs_conv3x3_wino_wgrad_gen_ws(x.data_ptr(), skip.data_ptr() if skip is not None else None, dy.data_ptr(),
                                                   _nhwc(y_out).data_ptr() if dact else None, dw.data_ptr(), ptr(db), B, H, W, c1, c2, co,
                                                   int(up), dact, _WINO_WGS, ptr(ws), nws, _lib.stream()), "dvs_conv3x3_wino_wgrad_gen")
    return (None if dw_out is not None else dw), (None if db_out is not None else db)


def conv2d_dgrad(dy, weight, x_shape, stride, pad, reflect, y_out=None, act=None, split_c1=0, prepadded=False, residual=None):
    """dx [B,Cin,H,W] (NHWC) of a forward conv described by (weight, x_shape, stride, pad, reflect).
    split_c1 > 0 (upsample+concat forward, x_shape = the concatenated full-resolution input): returns
    (d coarse [B,C1,H/2,W/2] -- the 2x2-summed gradient of the upsampled operand, d skip [B,Cin-C1,H,W] or None).
    residual [B,Cin,H,W]: another gradient of the same input (a skip / downsample path's), added in the kernel's epilogue."""
    l = _lib.lib()
    dy = _nhwc(dy)
    w = _nhwc(weight)
    Cout, Cin, kh, kw = weight.shape
    B, _, H, W = x_shape
    wt = _packed_weight(w, weight)
    d = _desc(B, Cin, H, W, weight.shape, stride, pad, reflect)
    if prepadded:
        d.pad_mode = 2           # x_shape is the reflection-padded input: zero padding, unpadded flops in the profile
    dact = ACT[act]
    yo = _nhwc(y_out).data_ptr() if dact else None
    if split_c1:
        if residual is not None:
            raise _lib.DvsError("conv2d_dgrad: a residual cannot be combined with the upsample+concat split")
        dx = zeropool.zeros((B, H // 2, W // 2, split_c1), dy.device).permute(0, 3, 1, 2)      # NHWC memory
        dskip = (torch.empty((B, Cin - split_c1, H, W), device=dy.device, dtype=torch.float32, memory_format=CL)
                 if split_c1 < Cin else None)
        check(l.dvs_conv2d_dgrad(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), yo, dact,
                                 dskip.data_ptr() if dskip is not None else None, split_c1, _lib.stream()),
              "dvs_conv2d_dgrad")
        return dx, dskip
    dx = torch.empty((B, Cin, H, W), device=dy.device, dtype=torch.float32, memory_format=CL)
    if residual is not None:
        residual = _nhwc(residual)
        if tuple(residual.shape) != tuple(dx.shape):
            raise _lib.DvsError("conv2d_dgrad: residual shape %s != gradient shape %s" % (tuple(residual.shape), tuple(dx.shape)))
    check(l.dvs_conv2d_dgrad_res(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), C.byref(d), yo, dact, None, 0,
                                 residual.data_ptr() if residual is not None else None, _lib.stream()), "dvs_conv2d_dgrad")
    return dx


def conv2d_wgrad(x, dy, weight_shape, stride, pad, reflect, want_bias, y_out=None, act=None, x2=None, in_scale=None,
                 in_shift=None, in_relu=False, nchw_planar=False, pooled=False, dw_out=None, db_out=None):
    """(dW with the weight's logical shape, dbias or None).  dw_out / db_out: gradient sinks (gradsink.py) --
    accumulate into these instead of fresh zero-filled tensors; the corresponding result is None."""
    l = _lib.lib()
    dy = _nhwc(dy)
    Cout, _, kh, kw = weight_shape
    x, x2, B, Cin, H, W = _geometry(x, tuple(weight_shape), x2, nchw_planar)
    d = _desc(B, Cin, H, W, weight_shape, stride, pad, reflect)
    if dw_out is not None:
        if nchw_planar or tuple(dw_out.shape) != tuple(weight_shape) or not dw_out.permute(0, 2, 3, 1).is_contiguous():
            raise _lib.DvsError("conv2d_wgrad: gradient sink must be a [Cout][kh][kw][Cin]-stored tensor of the weight's shape")
        dw = dw_out
    elif nchw_planar:
        dw = zeropool.zeros((Cout, Cin, kh, 8), dy.device, pooled=pooled)
    else:
        dw = zeropool.zeros(tuple(weight_shape), dy.device, channels_last=True, pooled=pooled)
    if db_out is not None:
        db = db_out
    else:
        db = zeropool.zeros((Cout,), dy.device, pooled=pooled) if want_bias else None
    f = _fusion(x, x2, in_scale, in_shift, in_relu, nchw_planar)
    dact = ACT[act]
    yo = _nhwc(y_out).data_ptr() if dact else None
    ws, nws = None, 0
    if _WGRAD_ORDERED or _lib.deterministic():               # slab workspace: partial tiles + an ordered second pass, no atomics on dw
        nws = l.dvs_conv2d_wgrad_workspace(C.byref(d), C.byref(f), dact, int(db is not None))
        ws = torch.empty(nws // 4, device=dy.device, dtype=torch.float32) if nws else None
    check(l.dvs_conv2d_wgrad_ws(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ptr(db), C.byref(d), C.byref(f), yo, dact,
                                ptr(ws), nws, _lib.stream()), "dvs_conv2d_wgrad")
    if nchw_planar:
        dw = dw[..., :kw]
    return (None if dw_out is not None else dw), (None if db_out is not None else db)


_DIRECT_SLOTS = os.environ.get("DVS_DIRECT_STAT_SLOTS", "1") != "0"    # also for the implicit-GEMM kernels' statistics epilogue
STATS_SLOTTED = 4      # flag in conv2d(want_stats=G | STATS_SLOTTED): the statistics may come back as [slots][G][2][C]
_PREACT = os.environ.get("DVS_CONV_PREACT", "1") != "0"
_PADDED = os.environ.get("DVS_CONV_PADDED_DGRAD", "1") != "0"


class _Conv2d(torch.autograd.Function):
    """y = act(conv(pad(x [, upsample+concat x2]), w) + b) with the hand-written forward / data-gradient /
    weight-gradient kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, x2, opts):
        ctx.gs = gradsink.active()               # the stream set of the trainer that builds this graph (used in backward)
        stride, pad, reflect, act, planar, scale, shift, want_stats = opts[:8]
        passthrough = len(opts) > 8 and opts[8]  # also return x itself: its gradient (a skip path's) meets the data gradient here
        ctx.set_materialize_grads(False)         # no zero-filled "gradient" for the statistics output
        slots_ok = bool(int(want_stats) & STATS_SLOTTED)   # the caller adds up a [slots][G][2][C] table (nn_ops -> bn.bn_act)
        want_stats = int(want_stats) & 3
        groups = int(want_stats)                 # 0: none, 1: [2][C], 2: [2][2][C] (first / second half of the batch)
        # bias + ReLU (PoseNet's decoder, model/posenet_single.py:160-164) ride in the Winograd epilogue; the backward then forms
        # dZ = dY * [Y > 0] and the bias gradient in one pre-activation pass (dvs_act_bwd), which needs Cout / 4 to divide 256
        wino_tail = (bias is None and act is None) or (act == "relu" and not groups and 256 % max(weight.shape[0] // 4, 1) == 0)
        ctx.wino = (wino_tail and wino_eligible(weight, stride, pad, reflect, None, x2, planar, scale) and x.shape[1] == weight.shape[1]
                    and x.numel() // x.shape[1] * max(weight.shape[0], weight.shape[1]) * 4 < 2 ** 31      # 32-bit buffer offsets
                    and wino_pays(x.shape[0], x.shape[2], x.shape[3], weight.shape[1], weight.shape[0]))
        ctx.p16 = (not ctx.wino and bias is None and act is None and p16_eligible(weight, stride, pad, reflect, None, x2, planar, scale)
                   and x.shape[1] == weight.shape[1] and x.numel() // x.shape[1] * max(weight.shape[0], weight.shape[1]) * 4 < 2 ** 31)
        up2 = 2 if x2 is not None else 1
        ctx.p16_dec = (not ctx.wino and not ctx.p16 and not groups and p16_dec_eligible(weight, stride, pad, reflect, act, x, x2, planar, scale)
                       and x.shape[0] * (x.shape[2] + 2) * (x.shape[3] + 2) * up2 * up2 * max(weight.shape[0], weight.shape[1]) * 4 < 2 ** 31)
        ctx.wino_dec = (not ctx.wino and not groups and wino_dec_eligible(weight, stride, pad, reflect, act, x, x2, planar, scale)
                        and x.shape[0] * (x.shape[2] + 2) * (x.shape[3] + 2) * up2 * up2
                        * max(weight.shape[0], weight.shape[1]) * 4 < 2 ** 31
                        and wino_pays(x.shape[0], up2 * x.shape[2], up2 * x.shape[3], weight.shape[1], weight.shape[0]))
        # the statistics epilogues end with same-address atomics: spread over STAT_SLOTS copies where the consumer adds them up
        slots = STAT_SLOTS if (groups and slots_ok and STAT_SLOTS > 1 and not planar and (ctx.wino or ctx.p16 or _DIRECT_SLOTS)) else 1
        if groups:
            shape = (2, weight.shape[0]) if groups == 1 else (groups, 2, weight.shape[0])
            stats = zeropool.zeros(((slots, groups) + shape[-2:]) if slots > 1 else shape, x.device)
        else:
            stats = None
        if ctx.wino:
            y = conv3x3_wino(x, weight, stats, groups, bias=bias, relu=act == "relu", stat_slots=slots)
        elif ctx.p16:
            y = conv3x3_p16(x, weight, stats, groups, stat_slots=slots)
        elif ctx.p16_dec:
            y = conv3x3_p16_gen(x, x2, weight, bias, act, reflect=True)
        elif ctx.wino_dec:
            y = conv3x3_wino_gen(x, x2, weight, bias, act, reflect=True)
        else:
            y = conv2d_forward(x, weight, bias, stride, pad, reflect, act, x2=x2, in_scale=scale, in_shift=shift,
                               nchw_planar=planar, stats=stats, stat_groups=max(groups, 1), stat_slots=slots)
        ctx.opts = opts[:7]
        ctx.x_shape = tuple(x.shape)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias                      # only to find its gradient sink in backward
        ctx.weight_ref = weight                  # the object the operand caches (_prepacked, _wino_packed) are pinned to
        ctx.up_only = x2 is UPSAMPLE_ONLY
        ctx.save_for_backward(x, weight, None if ctx.up_only else x2, y if ACT[act] else None)
        ctx.passthrough = passthrough
        if passthrough:
            xa = x.view_as(x)
            if want_stats:
                ctx.mark_non_differentiable(stats)
                return y, stats, xa
            return y, xa
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y

    @staticmethod
    def backward(ctx, dy, *more):
        # extra output gradients: (d stats,) and / or (d x_alias,) in the order forward returned them
        dxa = more[-1] if ctx.passthrough and more else None
        if ctx.passthrough and dy is None:
            return dxa, None, None, None, None
        x, weight, x2, y = ctx.saved_tensors     # (unpacking also runs autograd's in-place modification check)
        weight = ctx.weight_ref                  # same data; the parameter object itself for the cache lookups
        stride, pad, reflect, act, planar, scale, shift = ctx.opts
        dx = dx2 = dw = db = None
        if dy is None:
            return None, None, None, None, None
        if ctx.up_only:
            x2 = UPSAMPLE_ONLY
        preact = False
        pre_db = None        # bias gradient already taken by the pre-activation pass (tensor to hand back, or True if sunk)
        if ACT[act] and (ctx.wino or (_PREACT and reflect and not planar and weight.shape[0] >= 64
                                      and (ctx.needs_input_grad[0] or x2 is not None) and ctx.needs_input_grad[1])):
            # wide decoder layers: form dZ = dY * act'(Y) once instead of in both gradient kernels' gathers (the data
            # gradient re-derives it for every tap and N tile), and take the bias gradient (column sums of dZ) in the same
            # pass -- the weight gradient then has neither an activation nor a bias path and runs on the LDS-DMA kernel
            dy = _nhwc(dy)
            dz = torch.empty_like(dy)
            Cout = weight.shape[0]
            db_ptr = None
            if ctx.has_bias and ctx.needs_input_grad[2] and 256 % (Cout // 4) == 0:
                bs = gradsink.target(ctx.bias_ref)
                if bs is not None:
                    gradsink.note(ctx.bias_ref, gradsink.cur_stream())
                    db_ptr, pre_db = bs.data_ptr(), True
                else:
                    pre_db = zeropool.zeros((Cout,), dy.device, pooled=_has_grad(ctx.bias_ref))
                    db_ptr = pre_db.data_ptr()
            check(_lib.lib().dvs_act_bwd(dy.data_ptr(), _nhwc(y).data_ptr(), dz.data_ptr(), dy.numel(), ACT[act], db_ptr, Cout,
                                         _lib.stream()), "dvs_act_bwd")
            dy, y, act = dz, None, None
            preact = True
        need_x = ctx.needs_input_grad[0] or (isinstance(x2, torch.Tensor) and ctx.needs_input_grad[3])
        if need_x:
            if planar:
                raise _lib.DvsError("the planar image input of conv1 has no gradient path")
            B = ctx.x_shape[0]
            padded = (_PADDED and preact and reflect and stride == 1 and pad == 1 and weight.shape[2] == 3
                      and weight.shape[0] % 32 == 0 and ctx.x_shape[2] >= 2)
            # bf16 mode, thin decoder layers (no pre-activation pass): padded-domain gradient on the thin patch kernel with the
            # activation derivative fused into its staging, then the reflection fold / upsample split
            p16_thin = (ctx.p16_dec and _lib._precision == "bf16" and not preact and reflect and stride == 1 and pad == 1
                        and (weight.shape[0] % 64 != 0 or weight.shape[1] % 64 != 0) and ctx.x_shape[2] >= 2 and dxa is None)
            if p16_thin:
                yk = dict(y_out=y, act=act) if ACT[act] else {}
                if x2 is None:
                    dx = conv2d_dgrad_padded(dy, weight, ctx.x_shape, p16=True, **yk)
                else:
                    dx, dx2 = conv2d_dgrad_padded(dy, weight, (B, weight.shape[1], 2 * ctx.x_shape[2], 2 * ctx.x_shape[3]),
                                                  split_c1=ctx.x_shape[1], p16=True, **yk)
            elif ctx.wino:
                dx = conv3x3_wino(dy, weight, flip=True, residual=dxa)       # + the skip path's gradient in the epilogue
                dxa = None
            elif ctx.p16 and _lib._precision == "bf16":
                dx = conv3x3_p16(dy, weight, flip=True, residual=dxa)
                dxa = None
            elif x2 is None:
                if padded and ctx.x_shape[2] >= 3 and ctx.x_shape[3] >= 3:
                    dx = conv2d_dgrad_padded(dy, weight, ctx.x_shape, wino=ctx.wino_dec, p16=ctx.p16_dec and _lib._precision == "bf16")
                else:
                    dx = conv2d_dgrad(dy, weight, ctx.x_shape, stride, pad, reflect, y, act, residual=dxa)   # + the other paths' gradient
                    dxa = None
            else:
                C1, H, W = ctx.x_shape[1], 2 * ctx.x_shape[2], 2 * ctx.x_shape[3]
                # gradient of the nearest 2x upsample = 2x2 sum; of the concat = channel split: both done in
                # the data-gradient kernel's epilogue (or, on the padded-domain path, in dvs_reflect_fold)
                if padded:
                    dx, dx2 = conv2d_dgrad_padded(dy, weight, (B, weight.shape[1], H, W), split_c1=C1, wino=ctx.wino_dec,
                                                  p16=ctx.p16_dec and _lib._precision == "bf16")
                else:
                    dx, dx2 = conv2d_dgrad(dy, weight, (B, weight.shape[1], H, W), stride, pad, reflect, y, act, split_c1=C1)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # pool-backed scratch only when autograd will add it into an existing .grad (never adopt it)
            wsink = None if planar else gradsink.target(weight)
            bsink = None if planar else gradsink.target(ctx.bias_ref)
            # fully sunk (nothing goes back to autograd): run beside the data-gradient chain on the side stream
            want_b = ctx.has_bias and pre_db is None        # the pre-activation pass may already hold the bias gradient
            if pre_db is not None:
                bsink_w = None
            else:
                bsink_w = bsink
            side = gradsink.of(ctx).side_stream() if wsink is not None and (bsink is not None or not ctx.has_bias) else None
            # decoder layer on the Winograd kernels whose activation derivative and bias gradient the pre-activation pass took
            # decoder layers (wide ones: activation derivative and bias gradient already taken by the pre-activation pass; thin
            # ones with 32-channel blocks: both fused into the kernel's dY loads) on the Winograd kernel's reflect / upsample gathers
            wino_gen = (reflect and stride == 1 and pad == 1 and not planar and scale is None and weight.shape[2] == 3
                        and weight.shape[3] == 3 and act in (None, "elu", "relu") and (ACT[act] or not want_b)
                        and wino_dec_wgrad_eligible(weight.shape, x, x2)
                        and wino_dec_wgrad_pays(dy.shape[0], dy.shape[2], dy.shape[3], weight.shape[1], weight.shape[0]))
            p16_w = ctx.p16 and _lib._precision == "bf16" and not ctx.has_bias and not _lib.deterministic()
            # bf16 mode: the decoder's layers with 32-channel blocks on the patch kernel's reflect / upsample / concat gather
            c2w = x2.shape[1] if isinstance(x2, torch.Tensor) else 0
            p16_gen_w = (_P16 and _lib._precision == "bf16" and not _lib.deterministic() and reflect and stride == 1 and pad == 1
                         and not planar and scale is None and weight.shape[2] == 3 and weight.shape[3] == 3
                         and act in (None, "elu", "relu") and (ACT[act] or not want_b) and weight.shape[0] % 16 == 0
                         and x.shape[1] % 16 == 0 and c2w % 16 == 0 and (c2w == 0 or x.shape[1] % 32 == 0)
                         and x.shape[1] + c2w == weight.shape[1]
                         and dy.numel() // dy.shape[1] * max(weight.shape[0], weight.shape[1]) * 4 < 2 ** 31)
            if side is None:
                if wsink is not None or bsink is not None:
                    gradsink.note(weight, gradsink.cur_stream())
                    gradsink.note(ctx.bias_ref, gradsink.cur_stream())
                if ctx.wino and wino_wgrad_eligible(weight.shape, x):
                    dw = conv3x3_wino_wgrad(x, dy, tuple(weight.shape), dw_out=wsink, pooled=_has_grad(weight))
                elif p16_w:
                    dw = conv3x3_p16_wgrad(x, dy, tuple(weight.shape), dw_out=wsink, pooled=_has_grad(weight))
                elif p16_gen_w:
                    dw, db = conv3x3_p16_wgrad_gen(x, x2, dy, tuple(weight.shape), dw_out=wsink, pooled=_has_grad(weight), y_out=y,
                                                   act=act, want_bias=want_b, db_out=bsink_w)
                elif wino_gen:
                    dw, db = conv3x3_wino_wgrad_gen(x, x2, dy, tuple(weight.shape), dw_out=wsink, pooled=_has_grad(weight), y_out=y,
                                                    act=act, want_bias=want_b, db_out=bsink_w)
                else:
                    dw, db = conv2d_wgrad(x, dy, tuple(weight.shape), stride, pad, reflect, want_b, y, act, x2=x2,
                                          in_scale=scale, in_shift=shift, nchw_planar=planar,
                                          pooled=_has_grad(weight), dw_out=wsink, db_out=bsink_w)
                if pre_db is not None and pre_db is not True:
                    db = pre_db
            else:
                cur = gradsink.cur_stream()
                # layout copies (if any) run HERE, on the compute stream, in front of the side stream's wait
                x, dy = _nhwc(x), _nhwc(dy)
                y = _nhwc(y) if isinstance(y, torch.Tensor) else y
                x2 = _nhwc(x2) if isinstance(x2, torch.Tensor) else x2
                gradsink.note(weight, cur, side)
                gradsink.note(ctx.bias_ref, cur, side)
                side.wait_stream(cur)
                # (the ordered / deterministic forms allocate a workspace inside: those need torch's current stream switched)
                with (torch.cuda.stream(side) if (_WGRAD_ORDERED or _lib.deterministic()) else _lib.on_stream(side)):
                    if ctx.wino and wino_wgrad_eligible(weight.shape, x):
                        conv3x3_wino_wgrad(x, dy, tuple(weight.shape), dw_out=wsink)
                    elif p16_w:
                        conv3x3_p16_wgrad(x, dy, tuple(weight.shape), dw_out=wsink)
                    elif p16_gen_w:
                        conv3x3_p16_wgrad_gen(x, x2, dy, tuple(weight.shape), dw_out=wsink, y_out=y, act=act, db_out=bsink_w)
                    elif wino_gen:
                        conv3x3_wino_wgrad_gen(x, x2, dy, tuple(weight.shape), dw_out=wsink, y_out=y, act=act, db_out=bsink_w)
                    else:
                        conv2d_wgrad(x, dy, tuple(weight.shape), stride, pad, reflect, want_b, y, act, x2=x2,
                                     in_scale=scale, in_shift=shift, dw_out=wsink, db_out=bsink_w)
                for t in (x, dy, y, x2):
                    if isinstance(t, torch.Tensor):
                        t.record_stream(side)                # keep the allocator from recycling them under the kernel
        if dxa is not None:
            dx = dxa if dx is None else dx + dxa
        return dx, dw, db, dx2, None


class _HeadConv(torch.autograd.Function):
    """Narrow-output convolution (Cout <= 8, stride 1, 'same' size) on the vector ALUs: the disparity heads
    and PoseNet's last 1x1 (dvs_conv2d_head_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, reflect, act, passthrough=False):
        ctx.gs = gradsink.active()
        x_in = x
        x, w = _nhwc(x), _nhwc(weight)
        B, Cin, H, W = x.shape
        d = _desc(B, Cin, H, W, weight.shape, 1, pad, reflect)
        y = torch.empty((B, weight.shape[0], H, W), device=x.device, dtype=torch.float32, memory_format=CL)
        check(_lib.lib().dvs_conv2d_head_fwd(x.data_ptr(), w.data_ptr(), ptr(bias), y.data_ptr(), C.byref(d), ACT[act],
                                             _lib.stream()), "dvs_conv2d_head_fwd")
        ctx.cfg = (pad, reflect, act, bias is not None)
        ctx.pooled = _has_grad(weight) and (bias is None or _has_grad(bias))
        ctx.params = (weight, bias)              # only to find their gradient sinks in backward
        ctx.save_for_backward(x, w, y)
        ctx.passthrough = passthrough
        if passthrough:
            # x handed on as a second output (the decoder's next level reads it): its gradient comes back into THIS node and is
            # added in the data-gradient kernel instead of by an autograd accumulation pass
            ctx.set_materialize_grads(False)
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, dy, dxa=None):
        if dy is None:                           # (passthrough only) the head's own output was not used
            return dxa, None, None, None, None, None, None
        gradsink.of(ctx).wait_pending(dy)        # a disparity gradient may still be on its way on another stream
        x, w, y = ctx.saved_tensors
        pad, reflect, act, has_bias = ctx.cfg
        B, Cin, H, W = x.shape
        d = _desc(B, Cin, H, W, w.shape, 1, pad, reflect)
        dy = _nhwc(dy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        wsink, bsink = gradsink.target(ctx.params[0]), gradsink.target(ctx.params[1])
        if wsink is not None and not wsink.permute(0, 2, 3, 1).is_contiguous():
            wsink = None
        gradsink.note(ctx.params[0], gradsink.cur_stream())
        gradsink.note(ctx.params[1], gradsink.cur_stream())
        dw = wsink if wsink is not None else zeropool.zeros(tuple(w.shape), x.device, channels_last=True, pooled=ctx.pooled)
        db = bsink if bsink is not None else (zeropool.zeros((w.shape[0],), x.device, pooled=ctx.pooled) if has_bias else None)
        res = None
        if dxa is not None and dx is not None:
            res = _nhwc(dxa)
        check(_lib.lib().dvs_conv2d_head_bwd_res(x.data_ptr(), w.data_ptr(), y.data_ptr(), dy.data_ptr(),
                                                 dx.data_ptr() if dx is not None else None, dw.data_ptr(), ptr(db), C.byref(d),
                                                 ACT[act], res.data_ptr() if res is not None else None, _lib.stream()),
              "dvs_conv2d_head_bwd")
        if wsink is not None:
            dw = None
        if bsink is not None:
            db = None
        return dx, dw, db, None, None, None, None


def head_supported(x, weight, stride, padding, reflect_pad, x2=None, upsample=False, planar=False):
    cout, cin, kh, kw = weight.shape
    pad = reflect_pad if reflect_pad else padding
    return (cout in (1, 2, 6, 8) and cin % 4 == 0 and stride == 1 and kh == kw and 2 * pad == kh - 1 and x2 is None
            and not upsample and not planar and cout * kh * kw * cin * 4 <= 60 * 1024)


def head_conv2d(x, weight, bias, padding, reflect_pad, act, passthrough=False):
    """passthrough: returns (y, x') with x' = x as a second output of the same autograd node (see _HeadConv.forward)."""
    pad, reflect = (reflect_pad, True) if reflect_pad else (padding, False)
    return _HeadConv.apply(x, weight, bias, pad, reflect, act, bool(passthrough))


def supported(x, weight, x2=None, planar=False, upsample=False):
    """True when the hand-written kernels cover this problem (16-byte NHWC gathers need channel counts
    that are multiples of 4; the 1- and 6-channel heads are handled elsewhere)."""
    cout, cin = weight.shape[0], weight.shape[1]
    if cout % 4:
        return False
    if planar:
        return weight.shape[3] <= 8
    if cin % 4:
        return False
    if x2 is not None and x.shape[1] % 32:
        return False
    return True


def conv2d(x, weight, bias=None, stride=1, padding=0, reflect_pad=0, act=None, x2=None, upsample=False,
           planar_norm=None, want_stats=False, passthrough=False):
    """Differentiable fused convolution.  reflect_pad=1 means ReflectionPad2d(1) in front of a valid conv;
    x2 / upsample select the decoder's upsample(+concat) gather; planar_norm=(scale, shift) selects the
    encoder-conv1 path (planar image in, normalisation fused)."""
    planar = planar_norm is not None
    scale, shift = planar_norm if planar else (None, None)
    pad, reflect = (reflect_pad, True) if reflect_pad else (padding, False)
    if x2 is None and upsample:
        x2 = UPSAMPLE_ONLY
    # want_stats: also return [2][Cout] per-channel sum / sum of squares of y (BatchNorm batch statistics)
    # passthrough: the result tuple ends with x itself as an output of the same autograd node -- a BasicBlock takes its identity
    # branch from it, so that the skip gradient arrives in THIS node's backward and is added in the data-gradient kernel's
    # epilogue instead of by a separate autograd accumulation pass
    return _Conv2d.apply(x, weight, bias, x2, (stride, pad, reflect, act, planar, scale, shift, int(want_stats), bool(passthrough)))
