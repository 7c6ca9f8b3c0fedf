"""torch.autograd.Function wrappers over the C-ABI (include/dvslam.h).  PyTorch only provides device
memory, the stream and the autograd graph here; all arithmetic is in libdvslam_hip.so."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ChainBwdIO, ChainCfg, ChainFwdIO, check, ptr

MAX_SCALES = _lib.MAX_SCALES


def _f32c(t):
    if t.dtype != torch.float32:
        raise _lib.DvsError("fp32 tensors only (got %s)" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------- a4
class _PoseToMat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        from . import gradsink
        ctx.gs = gradsink.active()
        aa = _f32c(axisangle.reshape(-1, 3))
        tr = _f32c(translation.reshape(-1, 3))
        B = aa.shape[0]
        M = torch.empty(B, 4, 4, device=aa.device, dtype=torch.float32)
        check(_lib.lib().dvs_pose_to_mat_fwd(ptr(aa), ptr(tr), int(bool(invert)), ptr(M), B, _lib.stream()),
              "dvs_pose_to_mat_fwd")
        ctx.save_for_backward(aa, tr)
        ctx.invert = int(bool(invert))
        ctx.shapes = (axisangle.shape, translation.shape)
        return M

    @staticmethod
    def backward(ctx, dM):
        from . import gradsink
        gradsink.of(ctx).wait_pending(dM)
        aa, tr = ctx.saved_tensors
        dM = _f32c(dM)
        d_aa, d_tr = torch.empty_like(aa), torch.empty_like(tr)
        check(_lib.lib().dvs_pose_to_mat_bwd(ptr(aa), ptr(tr), ctx.invert, ptr(dM), ptr(d_aa), ptr(d_tr),
                                             aa.shape[0], _lib.stream()), "dvs_pose_to_mat_bwd")
        return d_aa.reshape(ctx.shapes[0]), d_tr.reshape(ctx.shapes[1]), None


def pose_to_mat(axisangle, translation, invert=False):
    """transformation_from_parameters (vo/learner_func.py:29-46) on the GPU."""
    return _PoseToMat.apply(axisangle, translation, invert)


# --------------------------------------------------------------------------------------- a5-a12
def make_chain_cfg(B, H, W, disp_shapes, auto_mask=True, min_depth=0.1, max_depth=10.0, ssim_ratio=0.85,
                   smoothness_ratio=1e-3):
    cfg = ChainCfg()
    cfg.B, cfg.H, cfg.W, cfg.num_scales = B, H, W, len(disp_shapes)
    for s, (h, w) in enumerate(disp_shapes):
        cfg.hs[s], cfg.ws[s] = h, w
    cfg.auto_mask = int(bool(auto_mask))
    cfg.min_depth, cfg.max_depth = min_depth, max_depth
    cfg.ssim_ratio, cfg.smoothness_ratio = ssim_ratio, smoothness_ratio
    return cfg


def chain_workspace_bytes(cfg):
    sizes = [C.c_size_t() for _ in range(4)]
    check(_lib.lib().dvs_chain_workspace(C.byref(cfg), *[C.byref(s) for s in sizes]), "dvs_chain_workspace")
    return [s.value for s in sizes]


# Loss-chain backward by scale on two streams (see _LossChain.backward): used when the caller has given a second stream
# (loss_chain(aux_stream=...): MonodepthTrainer passes its PoseNet stream) AND every gradient it hands out late goes to a consumer
# that waits for it (StreamSet.wait_pending: the disparity heads' and the pose-matrix backward); DVS_CHAIN_SPLIT=0 switches it off.
_CHAIN_SPLIT = os.environ.get("DVS_CHAIN_SPLIT", "1") != "0"


class _LossChain(torch.autograd.Function):
    """losses[s] = loss/s of MonodepthTrainer._compute_losses (vo/learner_new.py:175-258) as one fused
    forward launch; backward gives d/d disp_s and d/d cam_T_cam."""

    @staticmethod
    def forward(ctx, opts, target, src_l, src_r, K, inv_K, T_l, T_r, noise, *disps):
        B, _, H, W = target.shape
        dev = target.device
        target, src_l, src_r = _f32c(target), _f32c(src_l), _f32c(src_r)
        K, inv_K, T_l, T_r = _f32c(K), _f32c(inv_K), _f32c(T_l), _f32c(T_r)
        disps = [_f32c(d) for d in disps]
        S = len(disps)
        cfg = make_chain_cfg(B, H, W, [tuple(d.shape[2:]) for d in disps], opts["auto_mask"], opts["min_depth"],
                             opts["max_depth"], opts["ssim_ratio"], opts["smoothness_ratio"])
        nb = chain_workspace_bytes(cfg)
        partials = torch.empty(nb[0] // 4, device=dev, dtype=torch.float32)
        sel = torch.empty(B, H, W, device=dev, dtype=torch.uint8)
        stats = torch.empty(nb[2] // 4, device=dev, dtype=torch.float32)   # [B,S,4] sums + the per-image camera table
        losses = torch.empty(S, device=dev, dtype=torch.float32)
        io = ChainFwdIO()
        io.target = ptr(target)
        io.source[0], io.source[1] = ptr(src_l), ptr(src_r)
        for s in range(S):
            io.disp[s] = ptr(disps[s])
        io.K, io.inv_K = ptr(K), ptr(inv_K)
        io.T[0], io.T[1] = ptr(T_l), ptr(T_r)
        if noise is not None:
            noise = _f32c(noise)
            if tuple(noise.shape) != (S, B, 2, H, W):
                raise _lib.DvsError("noise must be [S,B,2,H,W]")
            io.noise = ptr(noise)
        io.seed = int(opts.get("seed", 0))
        io.partials, io.sel, io.stats, io.losses = ptr(partials), ptr(sel), ptr(stats), ptr(losses)
        extra = []
        if opts.get("materialize", False):
            for s in range(S):
                du = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
                dp = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
                io.disp_up[s], io.depth[s] = ptr(du), ptr(dp)
                extra += [du, dp]
                for f in range(2):
                    g = torch.empty(B, H, W, 2, device=dev, dtype=torch.float32)
                    c = torch.empty(B, 3, H, W, device=dev, dtype=torch.float32)
                    io.grid[s][f], io.color[s][f] = ptr(g), ptr(c)
                    extra += [g, c]
        check(_lib.lib().dvs_chain_fwd(C.byref(cfg), C.byref(io), _lib.stream()), "dvs_chain_fwd")
        ctx.save_for_backward(target, src_l, src_r, K, inv_K, T_l, T_r, noise, sel, stats, *disps)
        ctx.opts = {k: v for k, v in opts.items() if k != "aux_stream"}
        ctx.nbwd = nb[3]
        ctx.split_ok = bool(opts.get("split_ok", False))
        ctx.aux_stream = opts.get("aux_stream")          # the caller's second stream (MonodepthTrainer: the PoseNet stream)
        from . import gradsink
        ctx.gs = gradsink.active()
        ctx.mark_non_differentiable(sel, *extra)
        return (losses, sel, *extra)

    @staticmethod
    def backward(ctx, d_losses, *_unused):
        # (The end-of-backward fence of the weight-gradient side streams is queued by the first Function that uses a side
        # stream -- gradsink.StreamSet.side_stream -- whichever loss the pass started from.)
        target, src_l, src_r, K, inv_K, T_l, T_r, noise, sel, stats, *disps = ctx.saved_tensors
        B, _, H, W = target.shape
        dev = target.device
        S = len(disps)
        opts = ctx.opts
        cfg = make_chain_cfg(B, H, W, [tuple(d.shape[2:]) for d in disps], opts["auto_mask"], opts["min_depth"],
                             opts["max_depth"], opts["ssim_ratio"], opts["smoothness_ratio"])
        io = ChainFwdIO()
        io.target = ptr(target)
        io.source[0], io.source[1] = ptr(src_l), ptr(src_r)
        for s in range(S):
            io.disp[s] = ptr(disps[s])
        io.K, io.inv_K = ptr(K), ptr(inv_K)
        io.T[0], io.T[1] = ptr(T_l), ptr(T_r)
        io.sel, io.stats = ptr(sel), ptr(stats)
        g = ChainBwdIO()
        d_losses = _f32c(d_losses)
        g.d_losses = ptr(d_losses)
        d_disps = [torch.empty_like(d) for d in disps]
        for s in range(S):
            g.d_disp[s] = ptr(d_disps[s])
        d_T = [torch.empty(B, 4, 4, device=dev, dtype=torch.float32) for _ in range(2)]
        g.d_T[0], g.d_T[1] = ptr(d_T[0]), ptr(d_T[1])
        bwd_partials = torch.empty(ctx.nbwd // 4, device=dev, dtype=torch.float32)
        g.bwd_partials = ptr(bwd_partials)
        aux = ctx.aux_stream if (_CHAIN_SPLIT and S > 1 and ctx.split_ok) else None
        if aux is None:
            check(_lib.lib().dvs_chain_bwd(C.byref(cfg), C.byref(io), C.byref(g), _lib.stream()), "dvs_chain_bwd")
            return (None, None, None, None, None, None, d_T[0], d_T[1], None, *d_disps)
        # By scale: the decoder's backward needs d disp_0 first and the coarser scales only later, the PoseNet stream is
        # idle until d_T exists.  Scale 0 on this stream; scales 1 .. S-1 one after the other on `aux`, then the d_T
        # reduction there; the consumers (disparity heads, pose-matrix backward) wait for their own event.
        gs = ctx.gs
        l = _lib.lib()
        main = torch.cuda.current_stream()
        gs.clear_pending()
        aux.wait_stream(main)                                # d_losses, saved state, the fresh gradient buffers
        g.scale_begin, g.scale_end, g.phase = 0, 1, 1
        check(l.dvs_chain_bwd(C.byref(cfg), C.byref(io), C.byref(g), main.cuda_stream), "dvs_chain_bwd")
        ev_a = torch.cuda.Event()
        ev_a.record(main)
        with torch.cuda.stream(aux):
            for s in range(1, S):
                g.scale_begin, g.scale_end, g.phase = s, s + 1, 1
                check(l.dvs_chain_bwd(C.byref(cfg), C.byref(io), C.byref(g), aux.cuda_stream), "dvs_chain_bwd")
                ev = torch.cuda.Event()
                ev.record(aux)
                gs.set_pending(d_disps[s], ev)
            aux.wait_event(ev_a)                             # scale 0's partial sums
            g.scale_begin, g.scale_end, g.phase = 0, 0, 2
            check(l.dvs_chain_bwd(C.byref(cfg), C.byref(io), C.byref(g), aux.cuda_stream), "dvs_chain_bwd")
            ev_t = torch.cuda.Event()
            ev_t.record(aux)
        for t in d_T:
            gs.set_pending(t, ev_t)
        # everything the aux-stream kernels read or write must outlive them in the caching allocator's eyes -- including
        # the contiguous copies _f32c may have made of the images / intrinsics / poses in the forward
        for t in d_disps[1:] + d_T + [bwd_partials, d_losses, sel, stats, target, src_l, src_r, K, inv_K, T_l, T_r] + list(disps):
            t.record_stream(aux)
        return (None, None, None, None, None, None, d_T[0], d_T[1], None, *d_disps)


def loss_chain(target, src_l, src_r, K, inv_K, T_l, T_r, disps, noise=None, seed=0, materialize=False,
               auto_mask=True, min_depth=0.1, max_depth=10.0, ssim_ratio=0.85, smoothness_ratio=1e-3, aux_stream=None):
    """Fused view-synthesis loss chain.  Returns (losses[S], sel[B,H,W] uint8, extras) where extras is a
    per-scale list of dicts {disp_up, depth, grid:(l,r), color:(l,r)} when materialize=True, else []."""
    opts = dict(auto_mask=auto_mask, min_depth=min_depth, max_depth=max_depth, ssim_ratio=ssim_ratio,
                smoothness_ratio=smoothness_ratio, seed=seed, materialize=materialize, aux_stream=aux_stream)
    # the backward may hand d disp_1.. and d T out before they are complete (computed on a second stream) only if their
    # consumers are the nodes that wait for them: the disparity heads and the pose-matrix Function
    def fn_name(t):
        return type(t.grad_fn).__name__ if isinstance(t, torch.Tensor) and t.grad_fn is not None else None
    def no_hooks(t):
        return not getattr(t, "_backward_hooks", None)
    opts["split_ok"] = (all(fn_name(d) == "_HeadConvBackward" and no_hooks(d) for d in disps[1:])
                        and all(fn_name(t) in ("_PoseToMatBackward", None) and no_hooks(t) for t in (T_l, T_r)))
    out = _LossChain.apply(opts, target, src_l, src_r, K, inv_K, T_l, T_r, noise, *disps)
    losses, sel, flat = out[0], out[1], out[2:]
    extras = []
    if materialize:
        for s in range(len(disps)):
            e = flat[s * 6:(s + 1) * 6]
            extras.append({"disp_up": e[0], "depth": e[1], "grid": (e[2], e[4]), "color": (e[3], e[5])})
    return losses, sel, extras


# ------------------------------------------------------------------- standalone operators (a6-a11)
class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        depth, inv_K = _f32c(depth), _f32c(inv_K)
        B, _, H, W = depth.shape
        cam = torch.empty(B, 4, H * W, device=depth.device, dtype=torch.float32)
        check(_lib.lib().dvs_backproject_fwd(ptr(depth), ptr(inv_K), ptr(cam), B, H, W, _lib.stream()),
              "dvs_backproject_fwd")
        ctx.save_for_backward(inv_K)
        ctx.shape = depth.shape
        return cam

    @staticmethod
    def backward(ctx, d_cam):
        (inv_K,) = ctx.saved_tensors
        B, _, H, W = ctx.shape
        d_depth = torch.empty(ctx.shape, device=d_cam.device, dtype=torch.float32)
        check(_lib.lib().dvs_backproject_bwd(ptr(_f32c(d_cam)), ptr(inv_K), ptr(d_depth), B, H, W, _lib.stream()),
              "dvs_backproject_bwd")
        return d_depth, None


def backproject(depth, inv_K):
    return _Backproject.apply(depth, inv_K)


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, eps):
        points, K, T = _f32c(points), _f32c(K), _f32c(T)
        B = points.shape[0]
        grid = torch.empty(B, H, W, 2, device=points.device, dtype=torch.float32)
        check(_lib.lib().dvs_project_fwd(ptr(points), ptr(K), ptr(T), ptr(grid), B, H, W, eps, _lib.stream()),
              "dvs_project_fwd")
        ctx.save_for_backward(points, K, T)
        ctx.dims = (B, H, W, eps)
        return grid

    @staticmethod
    def backward(ctx, d_grid):
        points, K, T = ctx.saved_tensors
        B, H, W, eps = ctx.dims
        l = _lib.lib()
        d_points = torch.empty_like(points)
        d_T = torch.empty(B, 4, 4, device=points.device, dtype=torch.float32)
        ws = torch.empty(l.dvs_project_bwd_workspace(B, H, W) // 4, device=points.device, dtype=torch.float32)
        check(l.dvs_project_bwd(ptr(points), ptr(K), ptr(T), ptr(_f32c(d_grid)), ptr(d_points), ptr(d_T), ptr(ws),
                                B, H, W, eps, _lib.stream()), "dvs_project_bwd")
        return d_points, None, d_T, None, None, None


def project(points, K, T, H, W, eps=1e-7):
    return _Project.apply(points, K, T, H, W, eps)


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = _f32c(x), _f32c(y)
        B, Cc, H, W = x.shape
        out = torch.empty_like(x)
        check(_lib.lib().dvs_ssim_fwd(ptr(x), ptr(y), ptr(out), B * Cc, H, W, _lib.stream()), "dvs_ssim_fwd")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, y = ctx.saved_tensors
        B, Cc, H, W = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        check(_lib.lib().dvs_ssim_bwd(ptr(x), ptr(y), ptr(_f32c(d_out)), ptr(dx), ptr(dy), B * Cc, H, W,
                                      _lib.stream()), "dvs_ssim_bwd")
        return dx, dy


def ssim(x, y):
    return _SSIM.apply(x, y)


class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img):
        disp, img = _f32c(disp), _f32c(img)
        B, Cd, H, W = disp.shape
        if Cd != 1:
            raise _lib.DvsError("get_smooth_loss: disparity must have one channel")
        l = _lib.lib()
        out = torch.empty(1, device=disp.device, dtype=torch.float32)
        ws = torch.empty(l.dvs_smooth_workspace(B, H, W) // 4, device=disp.device, dtype=torch.float32)
        check(l.dvs_smooth_fwd(ptr(disp), ptr(img), ptr(out), ptr(ws), B, img.shape[1], H, W, _lib.stream()),
              "dvs_smooth_fwd")
        ctx.save_for_backward(disp, img)
        return out[0]

    @staticmethod
    def backward(ctx, d_out):
        disp, img = ctx.saved_tensors
        B, _, H, W = disp.shape
        d_disp = torch.empty_like(disp)
        check(_lib.lib().dvs_smooth_bwd(ptr(disp), ptr(img), ptr(_f32c(d_out.reshape(1))), ptr(d_disp), B,
                                        img.shape[1], H, W, _lib.stream()), "dvs_smooth_bwd")
        return d_disp, None


def smooth_loss(disp, img):
    return _Smooth.apply(disp, img)


# ------------------------------------------------------------------- supervised depth learner
class _DepthLoss(torch.autograd.Function):
    """(silog_s, smooth_s) for every scale of DepthLearner.multi_scale_loss (depth/depth_learner.py:97-117) in one
    forward and one backward launch: out = [silog_0..S-1, smooth_0..S-1]."""

    @staticmethod
    def forward(ctx, gt, mask, rgb, variance_focus, *preds):
        gt, rgb = _f32c(gt), _f32c(rgb)
        preds = [_f32c(d) for d in preds]
        B, _, H, W = rgb.shape
        if mask.dtype != torch.uint8:
            mask = mask.to(torch.uint8)
        mask = mask.contiguous()
        if gt.numel() != B * H * W or mask.numel() != B * H * W:
            raise _lib.DvsError("depth loss: gt_depth / valid_mask must be [B,1,H,W] or [B,H,W] of the image size")
        S = len(preds)
        cfg = _lib.DepthLossCfg()
        cfg.B, cfg.H, cfg.W, cfg.num_scales, cfg.variance_focus = B, H, W, S, variance_focus
        for s, d in enumerate(preds):
            if d.shape[0] != B or d.shape[1] != 1:
                raise _lib.DvsError("depth loss: pred_depths[%d] must be [B,1,h,w]" % s)
            cfg.hs[s], cfg.ws[s] = d.shape[2], d.shape[3]
        l = _lib.lib()
        nbytes = l.dvs_depth_loss_workspace(C.byref(cfg))
        if nbytes == 0:
            raise _lib.DvsError("dvs_depth_loss_workspace: %s" % l.dvs_last_error().decode())
        ws = torch.empty(nbytes // 4, device=rgb.device, dtype=torch.float32)
        out = torch.empty(2 * S, device=rgb.device, dtype=torch.float32)
        pp = (C.c_void_p * S)(*[ptr(d) for d in preds])
        check(l.dvs_depth_loss_fwd(C.byref(cfg), pp, ptr(gt), ptr(mask), ptr(rgb), ptr(ws), ptr(out), _lib.stream()),
              "dvs_depth_loss_fwd")
        ctx.save_for_backward(gt, mask, rgb, ws, *preds)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, d_out):
        gt, mask, rgb, ws, *preds = ctx.saved_tensors
        S = len(preds)
        d_preds = [torch.empty_like(d) for d in preds]
        pp = (C.c_void_p * S)(*[ptr(d) for d in preds])
        dd = (C.c_void_p * S)(*[ptr(d) for d in d_preds])
        check(_lib.lib().dvs_depth_loss_bwd(C.byref(ctx.cfg), pp, ptr(gt), ptr(mask), ptr(rgb), ptr(ws), ptr(_f32c(d_out)), dd,
                                            _lib.stream()), "dvs_depth_loss_bwd")
        return (None, None, None, None, *d_preds)


def depth_multiscale_losses(pred_depths, gt_depth, rgb, valid_mask, variance_focus=0.85):
    """(silog [S], smooth [S]) of the upsampled per-scale depth predictions (depth/depth_learner.py:97-117)."""
    out = _DepthLoss.apply(gt_depth, valid_mask, rgb, float(variance_focus), *pred_depths)
    S = len(pred_depths)
    return out[:S], out[S:]
