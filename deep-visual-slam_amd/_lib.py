"""ctypes binding of libdvslam_hip.so (the C-ABI declared in include/dvslam.h).

There is no fallback: `lib()` raises if the shared library has not been built, and every wrapper
raises `DvsError` on a non-zero status.  torch must be imported before the library is loaded so
that both share one HIP runtime (libamdhip64.so.7) and therefore one set of streams.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (loads torch's libamdhip64 first; see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVS_LIB") or os.path.join(HERE, "libdvslam_hip.so")    # DVS_LIB: A/B builds (tools/build_variant.py)
MAX_SCALES = 4
ABI_VERSION = 8

_vp = C.c_void_p


class DvsError(RuntimeError):
    pass


class ChainCfg(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("num_scales", C.c_int),
                ("hs", C.c_int * MAX_SCALES), ("ws", C.c_int * MAX_SCALES),
                ("auto_mask", C.c_int), ("min_depth", C.c_float), ("max_depth", C.c_float),
                ("ssim_ratio", C.c_float), ("smoothness_ratio", C.c_float)]


class ChainFwdIO(C.Structure):
    _fields_ = [("target", _vp), ("source", _vp * 2), ("disp", _vp * MAX_SCALES),
                ("K", _vp), ("inv_K", _vp), ("T", _vp * 2), ("noise", _vp), ("seed", C.c_uint64),
                ("partials", _vp), ("sel", _vp), ("stats", _vp), ("losses", _vp),
                ("disp_up", _vp * MAX_SCALES), ("depth", _vp * MAX_SCALES),
                ("grid", (_vp * 2) * MAX_SCALES), ("color", (_vp * 2) * MAX_SCALES)]


class ChainBwdIO(C.Structure):
    _fields_ = [("d_losses", _vp), ("d_disp", _vp * MAX_SCALES), ("d_T", _vp * 2),
                ("bwd_partials", _vp), ("scale_begin", C.c_int), ("scale_end", C.c_int), ("phase", C.c_int)]


class DepthLossCfg(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("num_scales", C.c_int),
                ("hs", C.c_int * MAX_SCALES), ("ws", C.c_int * MAX_SCALES), ("variance_focus", C.c_float)]


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("pad_mode", C.c_int)]


class ConvFusion(C.Structure):
    _fields_ = [("x2", _vp), ("C1", C.c_int), ("in_scale", _vp), ("in_shift", _vp), ("in_relu", C.c_int),
                ("nchw_planar", C.c_int), ("act", C.c_int), ("stats", _vp), ("stat_groups", C.c_int), ("residual", _vp),
                ("stat_slots", C.c_int)]


_SIGNATURES = {
    "dvs_conv2d_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(ConvDesc), C.POINTER(ConvFusion), _vp]),
    "dvs_wino_weights": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_conv3x3_wino_gen": (C.c_int, [_vp, _vp, _vp, _vp, _vp] + [C.c_int] * 13 + [_vp]),
    "dvs_conv3x3_wino_wgrad": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_conv3x3_wino_wgrad_gen": (C.c_int, [_vp] * 6 + [C.c_int] * 9 + [_vp]),
    "dvs_conv3x3_wino_wgrad_workspace": (C.c_size_t, [C.c_int] * 6),
    "dvs_conv3x3_wino_wgrad_ws": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "dvs_conv3x3_wino_wgrad_gen_ws": (C.c_int, [_vp] * 6 + [C.c_int] * 9 + [_vp, C.c_size_t, _vp]),
    "dvs_peak_probe_mfma": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), _vp]),
    "dvs_peak_probe_copy": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "dvs_wino_weights_batch": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "dvs_conv3x3_wino_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _vp]),
    "dvs_conv3x3_wino_fwd_slots": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 9 + [_vp]),
    "dvs_conv3x3_bf16_pack": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_conv3x3_bf16_wgrad": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 6 + [_vp]),
    "dvs_conv3x3_bf16_wgrad_gen": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 10 + [_vp]),
    "dvs_conv3x3_bf16_gen": (C.c_int, [_vp, _vp, _vp, _vp, _vp] + [C.c_int] * 13 + [_vp, C.c_int, _vp]),
    "dvs_conv3x3_bf16_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp] + [C.c_int] * 8 + [_vp]),
    "dvs_conv2d_pack_wt": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_conv2d_pack_wt_batch": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "dvs_reflect_fold": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_act_bwd": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp, C.c_int, _vp]),
    "dvs_conv2d_dgrad": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvDesc), _vp, C.c_int, _vp, C.c_int, _vp]),
    "dvs_conv2d_dgrad_res": (C.c_int, [_vp, _vp, _vp, C.POINTER(ConvDesc), _vp, C.c_int, _vp, C.c_int, _vp, _vp]),
    "dvs_conv2d_wgrad": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(ConvDesc), C.POINTER(ConvFusion), _vp, C.c_int, _vp]),
    "dvs_conv2d_wgrad_workspace": (C.c_size_t, [C.POINTER(ConvDesc), C.POINTER(ConvFusion), C.c_int, C.c_int]),
    "dvs_conv2d_wgrad_ws": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(ConvDesc), C.POINTER(ConvFusion), _vp, C.c_int, _vp, C.c_size_t, _vp]),
    "dvs_conv2d_head_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(ConvDesc), C.c_int, _vp]),
    "dvs_conv2d_head_bwd": (C.c_int, [_vp] * 7 + [C.POINTER(ConvDesc), C.c_int, _vp]),
    "dvs_conv2d_head_bwd_res": (C.c_int, [_vp] * 7 + [C.POINTER(ConvDesc), C.c_int, _vp, _vp]),
    "dvs_maxpool3x3s2_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_maxpool3x3s2_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_maxpool3x3s2_bwd_res": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_upsample2x_fwd": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_upsample2x_bwd": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_bn_bwd_slot_floats": (C.c_int, [C.c_int, C.c_int]),
    "dvs_bn_bwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int, _vp, _vp, C.c_int, _vp]),
    "dvs_bn_relu_maxpool_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_bn_relu_maxpool_bwd": (C.c_int, [_vp] * 9 + [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "dvs_bn_finalize": (C.c_int, [_vp, C.c_double, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "dvs_bn_fwd": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                              C.c_size_t, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_bn_finalize_slots": (C.c_int, [_vp, C.c_int, C.c_double, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, C.c_int, _vp,
                                        C.c_int, _vp]),
    "dvs_bn_fwd_slots": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                                    C.c_size_t, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_bn_apply_fwd": (C.c_int, [_vp] * 7 + [C.c_size_t, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_bn_bwd_workspace": (C.c_size_t, [C.c_size_t, C.c_int, C.c_int]),
    "dvs_bn_bwd_reduce": (C.c_int, [_vp] * 8 + [C.c_size_t, C.c_int, C.c_int, _vp]),
    "dvs_bn_bwd_apply": (C.c_int, [_vp] * 7 + [C.c_size_t, C.c_int, _vp, _vp, C.c_int, _vp]),
    "dvs_bn_bwd_reduce_ymask": (C.c_int, [_vp] * 8 + [C.c_size_t, C.c_int, C.c_int, _vp]),
    "dvs_bn_bwd_apply_ymask": (C.c_int, [_vp] * 9 + [C.c_size_t, C.c_int, _vp, _vp, C.c_int, _vp]),
    "dvs_last_error": (C.c_char_p, []),
    "dvs_set_deterministic": (C.c_int, [C.c_int]),
    "dvs_get_deterministic": (C.c_int, []),
    "dvs_set_precision": (C.c_int, [C.c_int]),
    "dvs_get_precision": (C.c_int, []),
    "dvs_abi_version": (C.c_int, []),
    "dvs_arch": (C.c_char_p, []),
    "dvs_profile_enable": (C.c_int, [C.c_int]),
    "dvs_profile_slots": (C.c_int, []),
    "dvs_profile_slot_name": (C.c_char_p, [C.c_int]),
    "dvs_profile_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "dvs_profile_work": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "dvs_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_int, C.c_float, C.c_int, _vp]),
    "dvs_pose_to_mat_fwd": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "dvs_pose_to_mat_bwd": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp]),
    "dvs_chain_workspace": (C.c_int, [C.POINTER(ChainCfg)] + [C.POINTER(C.c_size_t)] * 4),
    "dvs_chain_fwd": (C.c_int, [C.POINTER(ChainCfg), C.POINTER(ChainFwdIO), _vp]),
    "dvs_chain_bwd": (C.c_int, [C.POINTER(ChainCfg), C.POINTER(ChainFwdIO), C.POINTER(ChainBwdIO), _vp]),
    "dvs_backproject_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_backproject_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_project_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "dvs_project_bwd_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dvs_project_bwd": (C.c_int, [_vp] * 7 + [C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "dvs_ssim_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_ssim_bwd": (C.c_int, [_vp] * 5 + [C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_smooth_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dvs_smooth_fwd": (C.c_int, [_vp] * 4 + [C.c_int] * 4 + [_vp]),
    "dvs_smooth_bwd": (C.c_int, [_vp] * 4 + [C.c_int] * 4 + [_vp]),
    "dvs_resample_u8": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_u8_to_f32_planar": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_color_jitter_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "dvs_color_jitter": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_attention_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "dvs_attention_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "dvs_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "dvs_act_fwd": (C.c_int, [_vp, _vp, C.c_size_t, C.c_int, _vp]),
    "dvs_act_bwd_in": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    "dvs_resize_bilinear_ac_bwd": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_deconv_unshuffle": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, _vp]),
    "dvs_vit_patchify": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_vit_assemble": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_resize_bilinear_ac": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_deconv_shuffle": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "dvs_depth_loss_workspace": (C.c_size_t, [C.POINTER(DepthLossCfg)]),
    "dvs_depth_loss_fwd": (C.c_int, [C.POINTER(DepthLossCfg), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp]),
    "dvs_depth_loss_bwd": (C.c_int, [C.POINTER(DepthLossCfg), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp), _vp]),
}

_lib = None


def exported_symbols():
    """Every entry point include/dvslam.h declares (checked by tests/test_abi.py)."""
    return sorted(_SIGNATURES)


def lib():
    """Load libdvslam_hip.so once; raise loudly if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DvsError("libdvslam_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `python -m deep_visual_slam_amd.build`." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.dvs_abi_version() != ABI_VERSION:
            raise DvsError("libdvslam_hip.so ABI %d != binding ABI %d; rebuild" % (l.dvs_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


_deterministic = os.environ.get("DVS_DETERMINISTIC", "0") == "1"


def set_deterministic(on):
    """Deterministic forward pass (include/dvslam.h: dvs_set_deterministic): a test / debugging mode, ~10 % slower."""
    global _deterministic
    _deterministic = bool(on)
    check(lib().dvs_set_deterministic(int(_deterministic)), "dvs_set_deterministic")


def deterministic():
    if _deterministic and _lib is not None and not _lib.dvs_get_deterministic():
        _lib.dvs_set_deterministic(1)           # DVS_DETERMINISTIC=1 from the environment: tell the library once it is loaded
    return _deterministic


PRECISIONS = {"fp32": 0, "bf16": 1}


def set_precision(name):
    """Arithmetic of the implicit-GEMM convolutions (include/dvslam.h: dvs_set_precision): "fp32" (default, the parity mode) or
    "bf16" (bf16 operands, fp32 accumulate: the opt-in counterpart of the reference's use_amp)."""
    global _precision
    check(lib().dvs_set_precision(PRECISIONS[name]), "dvs_set_precision")
    _precision = name


_precision = "fp32"


def precision():
    return _precision


def check(rc, what):
    if rc != 0:
        raise DvsError("%s failed (%d): %s" % (what, rc, lib().dvs_last_error().decode()))


def ptr(t):
    """Device pointer of a contiguous fp32/u8 CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise DvsError("libdvslam_hip operates on GPU tensors only (got %s); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise DvsError("tensor must be contiguous")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


_stream_tls = threading.local()


class on_stream:
    """`with _lib.on_stream(side):` -- the dvs_* calls inside are handed `side` instead of torch's current stream, WITHOUT
    switching torch's current stream (torch.cuda.stream() costs ~15 us per enter / exit pair, and the weight-gradient branch of every
    convolution's backward takes one).  Only for code that allocates nothing inside: torch's allocator would hand out memory
    that belongs to the current stream."""
    __slots__ = ("handle", "prev")

    def __init__(self, torch_stream):
        self.handle = torch_stream.cuda_stream

    def __enter__(self):
        self.prev = getattr(_stream_tls, "override", None)
        _stream_tls.override = self.handle
        return self

    def __exit__(self, *exc):
        _stream_tls.override = self.prev
        return False


def stream():
    """hipStream_t of torch's current stream on the current device (what every dvs_* call is handed), or the stream of an
    enclosing `on_stream`.  Uses torch's raw accessor when it exists: `torch.cuda.current_stream()` builds a Stream object
    (~10 us), and this is called once per kernel launch -- 2.5 ms of host time per training step."""
    o = getattr(_stream_tls, "override", None)
    if o is not None:
        return o
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream
