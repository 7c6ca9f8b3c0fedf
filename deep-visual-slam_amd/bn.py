"""Training-mode BatchNorm2d (+ residual, + ReLU) on NHWC activations over dvs_bn_* of libdvslam_hip.so.

The batch statistics arrive from the producing convolution's epilogue (`stats` = per-channel sum and
sum of squares), so nn.BatchNorm2d + add + ReLU of a torchvision BasicBlock tail is one HBM pass forward
and two backward (model/resnet_encoder.py:100-111 through torchvision's BasicBlock).
"""
import os

import torch

from . import _lib, gradsink, zeropool
from ._lib import check, ptr

# DVS_BN_BWD_FUSED=1: reduce + apply without the partial-row sum kernel (dvs_bn_bwd).  Measured SLOWER over the step (26.96 -> 27.07 ms):
# the deep layers' tables are large next to their data (C = 512: 450 workgroups x 1 024 float atomics, then 128 KB of copies read by
# every workgroup of the apply pass: +12 us on each 11 us kernel), only the 64-channel layers break even -- so it stays off; the stem
# tail (C = 64, 10^6 rows) always uses the slot table.
_BWD_FUSED = os.environ.get("DVS_BN_BWD_FUSED", "0") == "1"
_YMASK = os.environ.get("DVS_BN_YMASK", "1") != "0"     # ReLU mask of residual-free BatchNorms recomputed from y in the backward

CL = torch.channels_last


def _nn_generation_bump():
    from . import nn_ops
    nn_ops.bump_generation()


def _finalize_groups(stats, count, bn, groups):
    """stats [G][2][C] (sum, sum of squares per sub-batch) -> [G][4][C] = scale, shift, mean, invstd, one launch;
    running statistics and num_batches_tracked are updated as G successive forward calls of nn.BatchNorm2d in
    training mode would have done."""
    C = stats.shape[-1]
    slots = stats.shape[0] if stats.dim() == 4 else 1        # [slots][G][2][C]: copies the Winograd forward spread its atomics over
    out = torch.empty(groups, 4, C, device=stats.device, dtype=torch.float32)
    train_stats = bn.training and bn.track_running_stats
    nbt = bn.num_batches_tracked if train_stats else None
    if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
        raise _lib.DvsError("bn: num_batches_tracked must be an int64 GPU tensor")
    if train_stats:
        _nn_generation_bump()
    check(_lib.lib().dvs_bn_finalize_slots(stats.data_ptr(), slots, float(count), ptr(bn.weight), ptr(bn.bias),
                                     ptr(bn.running_mean) if train_stats else None,
                                     ptr(bn.running_var) if train_stats else None,
                                     float(bn.momentum if bn.momentum is not None else 0.1), float(bn.eps),
                                     out[0, 0].data_ptr(), out[0, 1].data_ptr(), out[0, 2].data_ptr(), out[0, 3].data_ptr(), C,
                                     nbt.data_ptr() if nbt is not None else None, groups, _lib.stream()), "dvs_bn_finalize")
    return out


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, residual, res_gamma, res_beta, stat_args, res_fin, relu, groups):
        """z = act(bn(y) [+ residual | + bn_r(residual)]) in ONE launch: stat_args = (stats [G][2][C] from the conv
        epilogue, count, running_mean, running_var, momentum, eps, num_batches_tracked); scale / shift are derived in
        the kernel, which also leaves fin = [G][scale, shift, mean, invstd] for the backward and updates the running
        statistics.  res_fin = the downsample branch's own table.  With G = 2 the first and the second half of the
        batch are normalised with their own statistics."""
        l = _lib.lib()
        B, C, H, W = y.shape
        M = B * H * W // groups                       # rows per group
        z = torch.empty_like(y)
        stats, count, rmean, rvar, momentum, eps, nbt = stat_args
        slots = stats.shape[0] if stats.dim() == 4 else 1    # [slots][G][2][C] from the Winograd forward: added up in the kernel
        fin = torch.empty(groups, 4, C, device=y.device, dtype=torch.float32)
        # group g = rows [g M, (g + 1) M) with row g of the [G][.][C] tables
        check(l.dvs_bn_fwd_slots(y.data_ptr(), stats.data_ptr(), slots, float(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
                           float(momentum), float(eps), nbt.data_ptr() if nbt is not None else None, fin.data_ptr(),
                           residual.data_ptr() if residual is not None else None,
                           res_fin[0, 0].data_ptr() if res_fin is not None else None,
                           res_fin[0, 1].data_ptr() if res_fin is not None else None,
                                 z.data_ptr(), M, C, int(relu), groups, _lib.stream()), "dvs_bn_fwd")
        ctx.relu, ctx.groups = relu, groups
        ctx.affine = (gamma, beta, res_gamma, res_beta)          # only to find their gradient sinks in backward
        # ReLU without a residual: the backward recomputes the mask from y (dvs_bn_bwd_*_ymask) and does not need z
        ctx.ymask = bool(relu) and residual is None and _YMASK
        ctx.save_for_backward(y, gamma, residual, res_gamma, fin, res_fin, z if relu and not ctx.ymask else None)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, residual, res_gamma, fin, res_fin, z = ctx.saved_tensors
        l = _lib.lib()
        G = ctx.groups
        B, C, H, W = y.shape
        M = B * H * W // G
        dz = dz if dz.is_contiguous(memory_format=CL) else dz.contiguous(memory_format=CL)
        st = _lib.stream()
        need_du = (ctx.relu or residual is not None) and not ctx.ymask
        du = torch.empty_like(y) if need_du else dz
        dy = torch.empty_like(y)
        pooled = gamma.is_leaf and gamma.grad is not None
        fused = not _lib.deterministic() and _BWD_FUSED      # dvs_bn_bwd: no partial-row sum kernel between the two passes
        if fused:
            sums = torch.empty((G, 2, C), device=y.device, dtype=torch.float32)
            ws = zeropool.zeros((l.dvs_bn_bwd_slot_floats(C, G),), y.device)
        else:
            sums = zeropool.zeros((G, 2, C), y.device, pooled=pooled)
            ws = torch.empty(l.dvs_bn_bwd_workspace(M, C, G) // 4, device=y.device, dtype=torch.float32)
        g_par, b_par, rg_par, rb_par = ctx.affine
        gs, bs = gradsink.target(g_par), gradsink.target(b_par)
        sunk = gs is not None and bs is not None
        for par in ctx.affine:
            gradsink.note(par, gradsink.cur_stream())
        ds_res = residual is not None and res_fin is not None
        d_res = None
        if residual is not None:
            d_res = torch.empty_like(residual) if ds_res else du
        if ds_res:
            rsums = (torch.empty((G, 2, C), device=y.device, dtype=torch.float32) if fused
                     else zeropool.zeros((G, 2, C), y.device, pooled=pooled))
            rgs, rbs = gradsink.target(rg_par), gradsink.target(rb_par)
            rsunk = rgs is not None and rbs is not None
        if fused:
            check(l.dvs_bn_bwd(dz.data_ptr(), z.data_ptr() if (ctx.relu and not ctx.ymask) else None, y.data_ptr(), fin.data_ptr(),
                               int(ctx.ymask), ptr(gamma), du.data_ptr() if need_du else None, dy.data_ptr(), ws.data_ptr(),
                               None if sunk else sums.data_ptr(), M, C, ptr(gs) if sunk else None, ptr(bs) if sunk else None, G, st),
                  "dvs_bn_bwd")
            if ds_res:
                ws2 = zeropool.zeros((l.dvs_bn_bwd_slot_floats(C, G),), y.device)
                check(l.dvs_bn_bwd(du.data_ptr(), None, residual.data_ptr(), res_fin.data_ptr(), 0, ptr(res_gamma), None,
                                   d_res.data_ptr(), ws2.data_ptr(), None if rsunk else rsums.data_ptr(), M, C,
                                   ptr(rgs) if rsunk else None, ptr(rbs) if rsunk else None, G, st), "dvs_bn_bwd")
        elif ctx.ymask:
            check(l.dvs_bn_bwd_reduce_ymask(dz.data_ptr(), y.data_ptr(), fin[0, 2].data_ptr(), fin[0, 3].data_ptr(),
                                            fin[0, 0].data_ptr(), fin[0, 1].data_ptr(), sums.data_ptr(), ptr(ws), M, C, G, st),
                  "dvs_bn_bwd_reduce_ymask")
            check(l.dvs_bn_bwd_apply_ymask(dz.data_ptr(), y.data_ptr(), fin[0, 2].data_ptr(), fin[0, 3].data_ptr(),
                                           fin[0, 0].data_ptr(), fin[0, 1].data_ptr(), ptr(gamma), sums.data_ptr(), dy.data_ptr(), M, C,
                                           ptr(gs) if sunk else None, ptr(bs) if sunk else None, G, st), "dvs_bn_bwd_apply_ymask")
        else:
            check(l.dvs_bn_bwd_reduce(dz.data_ptr(), z.data_ptr() if ctx.relu else None, y.data_ptr(),
                                      fin[0, 2].data_ptr(), fin[0, 3].data_ptr(), du.data_ptr() if need_du else None,
                                      sums.data_ptr(), ptr(ws), M, C, G, st), "dvs_bn_bwd_reduce")
            check(l.dvs_bn_bwd_apply(du.data_ptr(), y.data_ptr(), fin[0, 2].data_ptr(), fin[0, 3].data_ptr(),
                                     ptr(gamma), sums.data_ptr(), dy.data_ptr(), M, C, ptr(gs) if sunk else None,
                                     ptr(bs) if sunk else None, G, st), "dvs_bn_bwd_apply")
        if ds_res and not fused:
            check(l.dvs_bn_bwd_reduce(du.data_ptr(), None, residual.data_ptr(), res_fin[0, 2].data_ptr(),
                                      res_fin[0, 3].data_ptr(), None, rsums.data_ptr(), ptr(ws), M, C, G, st),
                  "dvs_bn_bwd_reduce")
            check(l.dvs_bn_bwd_apply(du.data_ptr(), residual.data_ptr(), res_fin[0, 2].data_ptr(),
                                     res_fin[0, 3].data_ptr(), ptr(res_gamma), rsums.data_ptr(),
                                     d_res.data_ptr(), M, C, ptr(rgs) if rsunk else None,
                                     ptr(rbs) if rsunk else None, G, st), "dvs_bn_bwd_apply")
        d_gamma = d_beta = d_rg = d_rb = None
        if not sunk:
            d_gamma, d_beta = (sums[0, 1], sums[0, 0]) if G == 1 else (sums[:, 1].sum(0), sums[:, 0].sum(0))
        if ds_res and not rsunk:
            d_rg, d_rb = (rsums[0, 1], rsums[0, 0]) if G == 1 else (rsums[:, 1].sum(0), rsums[:, 0].sum(0))
        return dy, d_gamma, d_beta, d_res, d_rg, d_rb, None, None, None, None


class _BNReluPool(torch.autograd.Function):
    """(maxpool3x3s2(z) [, z]) with z = relu(bn(y)) -- the stem tail of model/resnet_encoder.py:102-104 (bn1, relu, maxpool) in
    one forward pass over y (dvs_bn_relu_maxpool_fwd; z is only written when a caller reads it: DepthNet's finest skip) and two
    backward passes (dvs_bn_relu_maxpool_bwd) that gather the pool gradient instead of reading a materialised dz."""

    @staticmethod
    def forward(ctx, y, gamma, beta, fin, groups, need_z):
        B, C, H, W = y.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        pooled = torch.empty((B, C, Ho, Wo), device=y.device, dtype=torch.float32, memory_format=CL)
        idx = torch.empty((B, Ho, Wo, C), device=y.device, dtype=torch.uint8)
        z = torch.empty_like(y) if need_z else None
        check(_lib.lib().dvs_bn_relu_maxpool_fwd(y.data_ptr(), fin.data_ptr(), z.data_ptr() if need_z else None, pooled.data_ptr(), idx.data_ptr(), B, H, W, C,
                                                 groups, _lib.stream()), "dvs_bn_relu_maxpool_fwd")
        ctx.groups = groups
        ctx.affine = (gamma, beta)               # only to find their gradient sinks in backward
        ctx.save_for_backward(y, gamma, fin, idx)
        ctx.set_materialize_grads(False)
        return (pooled, z) if need_z else pooled

    @staticmethod
    def backward(ctx, dpool, dz=None):
        if dpool is None and dz is None:
            return None, None, None, None, None, None
        y, gamma, fin, idx = ctx.saved_tensors
        l = _lib.lib()
        G = ctx.groups
        B, C, H, W = y.shape
        M = (B // G) * H * W
        if dpool is None:
            dpool = zeropool.zeros((B, idx.shape[1], idx.shape[2], C), y.device).permute(0, 3, 1, 2)
        dpool = dpool if dpool.is_contiguous(memory_format=CL) else dpool.contiguous(memory_format=CL)
        if dz is not None:
            dz = dz if dz.is_contiguous(memory_format=CL) else dz.contiguous(memory_format=CL)
        dy = torch.empty_like(y)
        sums = torch.empty((G, 2, C), device=y.device, dtype=torch.float32)
        ws = zeropool.zeros((l.dvs_bn_bwd_slot_floats(C, G),), y.device)      # the slot table of dvs_bn_bwd
        g_par, b_par = ctx.affine
        gs, bs = gradsink.target(g_par), gradsink.target(b_par)
        sunk = gs is not None and bs is not None
        for par in ctx.affine:
            gradsink.note(par, gradsink.cur_stream())
        check(l.dvs_bn_relu_maxpool_bwd(dpool.data_ptr(), idx.data_ptr(), dz.data_ptr() if dz is not None else None, y.data_ptr(), fin.data_ptr(), ptr(gamma),
                                        sums.data_ptr(), ptr(ws), dy.data_ptr(), B, H, W, C, ptr(gs) if sunk else None,
                                        ptr(bs) if sunk else None, G, _lib.stream()), "dvs_bn_relu_maxpool_bwd")
        d_gamma = d_beta = None
        if not sunk:
            d_gamma, d_beta = (sums[0, 1], sums[0, 0]) if G == 1 else (sums[:, 1].sum(0), sums[:, 0].sum(0))
        return dy, d_gamma, d_beta, None, None, None


def bn_relu_pool(y, bn, stats, groups=1, need_z=True):
    """(z or None, maxpool3x3s2(z)) with z = relu(bn(y)), training-mode BatchNorm with the batch statistics in `stats` (from the
    conv epilogue), running statistics updated as bn_act does."""
    if not y.is_cuda:
        raise _lib.DvsError("bn_relu_pool: GPU tensors only; this package has no CPU path")
    B, C, H, W = y.shape
    if B % groups:
        raise _lib.DvsError("bn_relu_pool: batch %d does not split into %d groups" % (B, groups))
    count = (B // groups) * H * W
    y = y if y.is_contiguous(memory_format=CL) else y.contiguous(memory_format=CL)
    st = stats if stats.dim() >= 3 else stats.unsqueeze(0)
    fin = _finalize_groups(st, count, bn, groups)
    out = _BNReluPool.apply(y, bn.weight, bn.bias, fin, groups, bool(need_z))
    return (out[1], out[0]) if need_z else (None, out)


def channel_stats(y, groups=1):
    """[G][2][C] per-channel sum / sum of squares of y (NHWC) in a fixed summation order: the deterministic stand-in for the
    convolution's atomic statistics epilogue (dvs_set_deterministic).  It is the BatchNorm backward's reduction kernel with
    (dz, mean, invstd) = (y, 0, 1): sum g = sum y, sum g * (y - 0) * 1 = sum y^2."""
    B, C, H, W = y.shape
    M = B * H * W // groups
    y = y if y.is_contiguous(memory_format=CL) else y.contiguous(memory_format=CL)
    l = _lib.lib()
    table = torch.zeros(groups, 4, C, device=y.device, dtype=torch.float32)        # rows 2, 3: mean = 0, invstd = 1
    table[:, 3].fill_(1.0)
    sums = torch.zeros(groups, 2, C, device=y.device, dtype=torch.float32)
    ws = torch.empty(l.dvs_bn_bwd_workspace(M, C, groups) // 4, device=y.device, dtype=torch.float32)
    check(l.dvs_bn_bwd_reduce(y.data_ptr(), None, y.data_ptr(), table[0, 2].data_ptr(), table[0, 3].data_ptr(), None,
                              sums.data_ptr(), ptr(ws), M, C, groups, _lib.stream()), "dvs_bn_bwd_reduce")
    return sums if groups > 1 else sums[0]


def supported_c(C, bn):
    """The fused kernels cover training-mode affine BatchNorm with C/4 dividing 256 (C = 16 ... 1024)."""
    return bn.training and bn.affine and C % 4 == 0 and (256 % (C // 4) == 0)


def bn_act(y, bn, stats, relu=False, residual=None, res_bn=None, res_stats=None, groups=1):
    """act(bn(y) + residual') with batch statistics taken from `stats` (filled by the conv that produced y);
    residual' = residual, or res_bn(residual) with its own `res_stats` (the BasicBlock downsample branch).
    groups = 2: y holds two independent batches back to back (PoseNet's two frame pairs); each half is
    normalised with its own statistics (stats = [2][2][C]) and the running statistics get both updates in order."""
    if not y.is_cuda:
        raise _lib.DvsError("bn_act: GPU tensors only; this package has no CPU path")
    B, C, H, W = y.shape
    if B % groups:
        raise _lib.DvsError("bn_act: batch %d does not split into %d groups" % (B, groups))
    count = (B // groups) * H * W
    y = y if y.is_contiguous(memory_format=CL) else y.contiguous(memory_format=CL)
    train_stats = bn.training and bn.track_running_stats
    nbt = bn.num_batches_tracked if train_stats else None
    if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
        raise _lib.DvsError("bn: num_batches_tracked must be an int64 GPU tensor")
    if train_stats:
        _nn_generation_bump()          # the kernel updates running_mean / running_var through raw pointers
    stat_args = (stats, count, bn.running_mean if train_stats else None, bn.running_var if train_stats else None,
                 bn.momentum if bn.momentum is not None else 0.1, bn.eps, nbt)
    res_fin = None
    if residual is not None:
        residual = residual if residual.is_contiguous(memory_format=CL) else residual.contiguous(memory_format=CL)
        if res_bn is not None:
            res_fin = _finalize_groups(res_stats, count, res_bn, groups)
    return _BNAct.apply(y, bn.weight, bn.bias, residual, res_bn.weight if res_bn is not None else None,
                        res_bn.bias if res_bn is not None else None, stat_args, res_fin, relu, groups)


class _AffineAct(torch.autograd.Function):
    """z = act(y * scale[c] + shift[c] [+ residual]) on NHWC tensors -- an eval-mode BatchNorm2d (running statistics) with
    the BasicBlock tail, differentiable w.r.t. y, scale, shift and residual.  Same kernels as the training path with the
    statistics frozen: dvs_bn_apply_fwd forward; backward = dvs_bn_bwd_reduce with (mean, invstd) = (0, 1), which yields
    du = dz * [z > 0], sum(du) = d shift and sum(du * y) = d scale, then dvs_bn_bwd_apply with zero sums and gamma = scale,
    which is dy = scale * du."""

    @staticmethod
    def forward(ctx, y, scale, shift, residual, relu):
        B, C, H, W = y.shape
        M = B * H * W
        scale, shift = scale.contiguous(), shift.contiguous()
        z = torch.empty_like(y)
        check(_lib.lib().dvs_bn_apply_fwd(y.data_ptr(), ptr(scale), ptr(shift), residual.data_ptr() if residual is not None else None,
                                          None, None, z.data_ptr(), M, C, int(relu), 1, _lib.stream()), "dvs_bn_apply_fwd")
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.save_for_backward(y, scale, z if relu else None)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, scale, z = ctx.saved_tensors
        l = _lib.lib()
        B, C, H, W = y.shape
        M = B * H * W
        dz = dz if dz.is_contiguous(memory_format=CL) else dz.contiguous(memory_format=CL)
        st = _lib.stream()
        zero_one = torch.zeros(3, C, device=y.device, dtype=torch.float32)      # rows: mean = 0, invstd = 1, zero sums (x2 below)
        zero_one[1].fill_(1.0)
        zeros2 = torch.zeros(2, C, device=y.device, dtype=torch.float32)
        sums = torch.zeros(2, C, device=y.device, dtype=torch.float32)
        du = torch.empty_like(y) if ctx.relu else dz
        ws = torch.empty(l.dvs_bn_bwd_workspace(M, C, 1) // 4, device=y.device, dtype=torch.float32)
        check(l.dvs_bn_bwd_reduce(dz.data_ptr(), z.data_ptr() if ctx.relu else None, y.data_ptr(), zero_one[0].data_ptr(),
                                  zero_one[1].data_ptr(), du.data_ptr() if ctx.relu else None, sums.data_ptr(), ptr(ws), M, C, 1, st),
              "dvs_bn_bwd_reduce")
        dy = torch.empty_like(y)
        check(l.dvs_bn_bwd_apply(du.data_ptr(), y.data_ptr(), zero_one[0].data_ptr(), zero_one[1].data_ptr(), ptr(scale),
                                 zeros2.data_ptr(), dy.data_ptr(), M, C, None, None, 1, st), "dvs_bn_bwd_apply")
        return dy, sums[1], sums[0], (du if ctx.has_res else None), None


def affine_supported(C):
    return C % 4 == 0 and 256 % (C // 4) == 0


def affine_act(y, scale, shift, relu=False, residual=None):
    """act(y * scale + shift [+ residual]); scale / shift are [C] tensors (eval-mode BatchNorm as an affine map)."""
    if not y.is_cuda:
        raise _lib.DvsError("affine_act: GPU tensors only; this package has no CPU path")
    C = y.shape[1]
    if not affine_supported(C):
        raise _lib.DvsError("affine_act: C %% 4 == 0 and C/4 dividing 256 required (got C = %d)" % C)
    y = y if y.is_contiguous(memory_format=CL) else y.contiguous(memory_format=CL)
    if residual is not None:
        residual = residual if residual.is_contiguous(memory_format=CL) else residual.contiguous(memory_format=CL)
        if residual.shape != y.shape:
            raise _lib.DvsError("affine_act: residual shape %s != %s" % (tuple(residual.shape), tuple(y.shape)))
    return _AffineAct.apply(y, scale, shift, residual, relu)
