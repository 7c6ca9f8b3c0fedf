"""Training-mode BatchNorm2d (+ residual, + ReLU) on NHWC activations over dvs_bn_* of libdvslam_hip.so.

The batch statistics arrive from the producing convolution's epilogue (`stats` = per-channel sum and
sum of squares), so nn.BatchNorm2d + add + ReLU of a torchvision BasicBlock tail is one HBM pass forward
and two backward (model/resnet_encoder.py:100-111 through torchvision's BasicBlock).
"""
import torch

from . import _lib, gradsink, zeropool
from ._lib import check, ptr

CL = torch.channels_last


def _finalize(stats, count, bn):
    C = stats.shape[1]
    dev = stats.device
    out = torch.empty(4, C, device=dev, dtype=torch.float32)      # scale, shift, mean, invstd
    train_stats = bn.training and bn.track_running_stats
    nbt = bn.num_batches_tracked if train_stats else None
    if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
        raise _lib.DvsError("bn: num_batches_tracked must be an int64 GPU tensor")
    check(_lib.lib().dvs_bn_finalize(ptr(stats), float(count), ptr(bn.weight), ptr(bn.bias),
                                     ptr(bn.running_mean) if train_stats else None,
                                     ptr(bn.running_var) if train_stats else None,
                                     float(bn.momentum if bn.momentum is not None else 0.1), float(bn.eps),
                                     out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), C,
                                     nbt.data_ptr() if nbt is not None else None, _lib.stream()), "dvs_bn_finalize")
    return out


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, residual, res_gamma, res_beta, fin, res_fin, relu):
        """z = act(bn(y) [+ residual | + bn_r(residual)]); fin / res_fin = [scale, shift, mean, invstd]."""
        l = _lib.lib()
        B, C, H, W = y.shape
        M = B * H * W
        z = torch.empty_like(y)
        r_sc = res_fin[0].data_ptr() if res_fin is not None else None
        r_sh = res_fin[1].data_ptr() if res_fin is not None else None
        check(l.dvs_bn_apply_fwd(y.data_ptr(), fin[0].data_ptr(), fin[1].data_ptr(),
                                 residual.data_ptr() if residual is not None else None, r_sc, r_sh, z.data_ptr(), M, C,
                                 int(relu), _lib.stream()), "dvs_bn_apply_fwd")
        ctx.relu = relu
        ctx.affine = (gamma, beta, res_gamma, res_beta)          # only to find their gradient sinks in backward
        ctx.save_for_backward(y, gamma, residual, res_gamma, fin, res_fin, z if relu else None)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, gamma, residual, res_gamma, fin, res_fin, z = ctx.saved_tensors
        l = _lib.lib()
        B, C, H, W = y.shape
        M = B * H * W
        dz = dz if dz.is_contiguous(memory_format=CL) else dz.contiguous(memory_format=CL)
        st = _lib.stream()
        need_du = ctx.relu or residual is not None
        du = torch.empty_like(y) if need_du else dz
        sums = zeropool.zeros((2, C), y.device, pooled=gamma.grad is not None)
        ws = torch.empty(l.dvs_bn_bwd_workspace(M, C) // 4, device=y.device, dtype=torch.float32)
        check(l.dvs_bn_bwd_reduce(dz.data_ptr(), z.data_ptr() if ctx.relu else None, y.data_ptr(), fin[2].data_ptr(),
                                  fin[3].data_ptr(), du.data_ptr() if need_du else None, ptr(sums), ptr(ws), M, C, st),
              "dvs_bn_bwd_reduce")
        dy = torch.empty_like(y)
        g_par, b_par, rg_par, rb_par = ctx.affine
        gs, bs = gradsink.target(g_par), gradsink.target(b_par)
        sunk = gs is not None and bs is not None
        check(l.dvs_bn_bwd_apply(du.data_ptr(), y.data_ptr(), fin[2].data_ptr(), fin[3].data_ptr(), ptr(gamma), ptr(sums),
                                 dy.data_ptr(), M, C, ptr(gs) if sunk else None, ptr(bs) if sunk else None, st),
              "dvs_bn_bwd_apply")
        if sunk:
            d_gamma = d_beta = None
        else:
            d_gamma, d_beta = sums[1], sums[0]
        d_res = d_rg = d_rb = None
        if residual is not None:
            if res_fin is None:
                d_res = du
            else:
                rsums = zeropool.zeros((2, C), y.device, pooled=gamma.grad is not None)
                check(l.dvs_bn_bwd_reduce(du.data_ptr(), None, residual.data_ptr(), res_fin[2].data_ptr(),
                                          res_fin[3].data_ptr(), None, ptr(rsums), ptr(ws), M, C, st), "dvs_bn_bwd_reduce")
                d_res = torch.empty_like(residual)
                gs, bs = gradsink.target(rg_par), gradsink.target(rb_par)
                sunk = gs is not None and bs is not None
                check(l.dvs_bn_bwd_apply(du.data_ptr(), residual.data_ptr(), res_fin[2].data_ptr(), res_fin[3].data_ptr(),
                                         ptr(res_gamma), ptr(rsums), d_res.data_ptr(), M, C, ptr(gs) if sunk else None,
                                         ptr(bs) if sunk else None, st), "dvs_bn_bwd_apply")
                if not sunk:
                    d_rg, d_rb = rsums[1], rsums[0]
        return dy, d_gamma, d_beta, d_res, d_rg, d_rb, None, None, None


def supported_c(C, bn):
    """The fused kernels cover training-mode affine BatchNorm with C/4 dividing 256 (C = 16 ... 1024)."""
    return bn.training and bn.affine and C % 4 == 0 and (256 % (C // 4) == 0)


def bn_act(y, bn, stats, relu=False, residual=None, res_bn=None, res_stats=None):
    """act(bn(y) + residual') with batch statistics taken from `stats` (filled by the conv that produced y);
    residual' = residual, or res_bn(residual) with its own `res_stats` (the BasicBlock downsample branch)."""
    if not y.is_cuda:
        raise _lib.DvsError("bn_act: GPU tensors only; this package has no CPU path")
    B, C, H, W = y.shape
    count = B * H * W
    y = y if y.is_contiguous(memory_format=CL) else y.contiguous(memory_format=CL)
    fin = _finalize(stats, count, bn)
    res_fin = None
    if residual is not None:
        residual = residual if residual.is_contiguous(memory_format=CL) else residual.contiguous(memory_format=CL)
        if res_bn is not None:
            res_fin = _finalize(res_stats, count, res_bn)
    return _BNAct.apply(y, bn.weight, bn.bias, residual, res_bn.weight if res_bn is not None else None,
                        res_bn.bias if res_bn is not None else None, fin, res_fin, relu)
