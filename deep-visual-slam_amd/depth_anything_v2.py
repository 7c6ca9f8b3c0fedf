"""DepthAnythingV2 (ViT-S/B/L encoders + DPT head) -- drop-in for the reference's model/depth_anything_v2/dpt.py:154-222 and
dinov2.py:37-414 on the MI355X inference path (BASELINE.json configs[4]).

Same module tree, so `load_state_dict` of a `depth_anything_v2_vit{s,b,l}.pth` checkpoint works (keys `pretrained.*`,
`depth_head.*`); `forward(x [B,3,H,W]) -> depth [B,H,W]` with H, W multiples of 14.  The nn.Modules are parameter
containers only; the arithmetic is HIP:

  * every Linear / 1x1 / 3x3 / strided convolution is a launch of the fp32-MFMA implicit-GEMM engine (conv.conv2d_forward on
    NHWC tensors; a [M,K] token matrix is a [1,1,M,K] NHWC map), with bias, GELU / ReLU / sigmoid, the LayerScale
    (folded into the proj / fc2 weights) and the residual add in the epilogue;
  * attention, LayerNorm, patchify, cls / pos-embed assembly, align_corners bilinear resizes and the stride == kernel
    ConvTranspose2d scatter are the kernels of csrc/vit.hip.

Two paths behind the same `forward`: under `torch.no_grad()` the fused inference path (LayerScale folded into the proj / fc2
weights, residuals and GELU in the GEMM epilogues, no saved tensors); with autograd enabled the TRAINABLE path -- the same
kernels as autograd Functions (dvs_attention_fwd / _bwd with the saved log-sum-exp, dvs_layernorm_bwd, GELU / ReLU as their
own passes so the pre-activation is kept, the implicit-GEMM engine's data / weight-gradient kernels for every Linear and
convolution, dvs_resize_bilinear_ac_bwd, dvs_deconv_unshuffle), so the encoder swap can be fine-tuned by the VO trainer.
ViT-G's SwiGLU FFN is not built (`encoder="vitg"` raises).
"""
import math

import torch
import torch.nn as nn

from . import _lib, conv as _conv
from ._lib import check, ptr

CL = torch.channels_last


# ---------------------------------------------------------------------------------------------- low-level wrappers
def _as_map(x2d):
    """[M,K] row-major -> logical [1,K,1,M] tensor in channels_last memory (no copy)."""
    M, K = x2d.shape
    return x2d.view(1, 1, M, K).permute(0, 3, 1, 2)


def gemm(x2d, w4d, bias=None, act=None, residual=None):
    """act(x2d [M,K] @ w^T + bias [+ residual [M,N]]) on the implicit-GEMM engine; w4d = weight as [N,K,1,1] (channels_last)."""
    M = x2d.shape[0]
    res = _as_map(residual) if residual is not None else None
    y = _conv.conv2d_forward(_as_map(x2d), w4d, bias, 1, 0, act=act, residual=res)
    return y.permute(0, 2, 3, 1).reshape(M, w4d.shape[0])


def layernorm(x2d, weight, bias, eps):
    y = torch.empty_like(x2d)
    check(_lib.lib().dvs_layernorm_fwd(ptr(x2d), ptr(weight), ptr(bias), ptr(y), x2d.shape[0], x2d.shape[1], eps, _lib.stream()),
          "dvs_layernorm_fwd")
    return y


def attention(qkv2d, B, N, heads, head_dim):
    out = torch.empty(B * N, heads * head_dim, device=qkv2d.device, dtype=torch.float32)
    check(_lib.lib().dvs_attention_fwd(ptr(qkv2d), ptr(out), None, B, N, heads, head_dim, head_dim ** -0.5, _lib.stream()), "dvs_attention_fwd")
    return out


def resize_bilinear_ac(x, H, W):
    """F.interpolate(x, (H, W), mode="bilinear", align_corners=True) on an NHWC-memory tensor."""
    B, C, h, w = x.shape
    x = x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)
    y = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32, memory_format=CL)
    check(_lib.lib().dvs_resize_bilinear_ac(x.data_ptr(), y.data_ptr(), B, h, w, H, W, C, _lib.stream()), "dvs_resize_bilinear_ac")
    return y


def conv_transpose_s(x, w_gemm, bias_rep, k, cout):
    """nn.ConvTranspose2d(kernel_size=k, stride=k): one 1x1 GEMM to k*k*Cout channels, then the scatter."""
    B, _, h, w = x.shape
    g = _conv.conv2d_forward(x, w_gemm, bias_rep, 1, 0)
    y = torch.empty((B, cout, h * k, w * k), device=x.device, dtype=torch.float32, memory_format=CL)
    check(_lib.lib().dvs_deconv_shuffle(g.data_ptr(), y.data_ptr(), B, h, w, k, cout, _lib.stream()), "dvs_deconv_shuffle")
    return y


# ---------------------------------------------------------------------------------------------- autograd Functions (training)
class _AttentionF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv2d, B, N, heads, head_dim):
        qkv2d = qkv2d.contiguous()
        out = torch.empty(B * N, heads * head_dim, device=qkv2d.device, dtype=torch.float32)
        lse = torch.empty(B, heads, N, device=qkv2d.device, dtype=torch.float32)
        check(_lib.lib().dvs_attention_fwd(ptr(qkv2d), ptr(out), ptr(lse), B, N, heads, head_dim, head_dim ** -0.5, _lib.stream()),
              "dvs_attention_fwd")
        ctx.save_for_backward(qkv2d, out, lse)
        ctx.dims = (B, N, heads, head_dim)
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv2d, out, lse = ctx.saved_tensors
        B, N, heads, hd = ctx.dims
        d_out = d_out.contiguous()
        d_qkv = torch.empty_like(qkv2d)
        delta = torch.empty_like(lse)
        check(_lib.lib().dvs_attention_bwd(ptr(qkv2d), ptr(out), ptr(d_out), ptr(lse), ptr(delta), ptr(d_qkv), B, N, heads, hd,
                                           hd ** -0.5, _lib.stream()), "dvs_attention_bwd")
        return d_qkv, None, None, None, None


class _LayerNormF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, weight, bias, eps):
        x2d = x2d.contiguous()
        y = layernorm(x2d, weight.detach(), bias.detach(), eps)
        ctx.save_for_backward(x2d, weight)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x2d, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x2d)
        dg, db = torch.zeros_like(weight), torch.zeros_like(weight)
        check(_lib.lib().dvs_layernorm_bwd(ptr(x2d), ptr(weight.detach().contiguous()), ptr(dy), ptr(dx), ptr(dg), ptr(db), x2d.shape[0],
                                           x2d.shape[1], ctx.eps, _lib.stream()), "dvs_layernorm_bwd")
        return dx, dg, db, None


_ACT_CODE = {"relu": 1, "gelu": 4}


class _ActF(torch.autograd.Function):
    """ReLU / GELU as its own pass over a dense tensor (any layout: elementwise); the backward reads the saved INPUT."""

    @staticmethod
    def forward(ctx, x, act):
        if x.numel() % 4:
            raise _lib.DvsError("activation: element count must be a multiple of 4")
        dense = x if (x.is_contiguous() or x.is_contiguous(memory_format=CL)) else x.contiguous()
        y = torch.empty_like(dense)
        check(_lib.lib().dvs_act_fwd(dense.data_ptr(), y.data_ptr(), dense.numel(), _ACT_CODE[act], _lib.stream()), "dvs_act_fwd")
        ctx.save_for_backward(dense)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        if dy.stride() != x.stride():
            dy = dy.contiguous(memory_format=CL) if x.dim() == 4 and x.is_contiguous(memory_format=CL) and not x.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(x)
        check(_lib.lib().dvs_act_bwd_in(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _ACT_CODE[ctx.act], _lib.stream()), "dvs_act_bwd_in")
        return dx, None


class _ResizeF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        ctx.in_hw = tuple(x.shape[2:])
        return resize_bilinear_ac(x, H, W)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = dy.shape
        h, w = ctx.in_hw
        dy = dy if dy.is_contiguous(memory_format=CL) else dy.contiguous(memory_format=CL)
        dx = torch.empty((B, C, h, w), device=dy.device, dtype=torch.float32, memory_format=CL)
        check(_lib.lib().dvs_resize_bilinear_ac_bwd(dy.data_ptr(), dx.data_ptr(), B, h, w, H, W, C, _lib.stream()), "dvs_resize_bilinear_ac_bwd")
        return dx, None, None


class _ShuffleF(torch.autograd.Function):
    """[B, k*k*Co, h, w] (NHWC memory) -> [B, Co, h*k, w*k]: the scatter half of a stride == kernel ConvTranspose2d."""

    @staticmethod
    def forward(ctx, g, k, cout):
        B, _, h, w = g.shape
        g = g if g.is_contiguous(memory_format=CL) else g.contiguous(memory_format=CL)
        y = torch.empty((B, cout, h * k, w * k), device=g.device, dtype=torch.float32, memory_format=CL)
        check(_lib.lib().dvs_deconv_shuffle(g.data_ptr(), y.data_ptr(), B, h, w, k, cout, _lib.stream()), "dvs_deconv_shuffle")
        ctx.dims = (B, h, w, k, cout)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, h, w, k, cout = ctx.dims
        dy = dy if dy.is_contiguous(memory_format=CL) else dy.contiguous(memory_format=CL)
        dg = torch.empty((B, k * k * cout, h, w), device=dy.device, dtype=torch.float32, memory_format=CL)
        check(_lib.lib().dvs_deconv_unshuffle(dy.data_ptr(), dg.data_ptr(), B, h, w, k, cout, _lib.stream()), "dvs_deconv_unshuffle")
        return dg, None, None


def linear_train(x2d, weight, bias):
    """x2d @ weight^T + bias with autograd, on the implicit-GEMM engine's forward / data-gradient / weight-gradient kernels
    (a Linear is a 1x1 convolution of the [1, K, 1, M] token map)."""
    N, K = weight.shape
    y = _conv.conv2d(_as_map(x2d.contiguous()), weight.reshape(N, K, 1, 1), bias, 1, 0, 0, None)
    return y.permute(0, 2, 3, 1).reshape(x2d.shape[0], N)


def _require_gpu(x, who):
    if not x.is_cuda:
        raise _lib.DvsError("%s: GPU tensors only (got %s); this package has no CPU path" % (who, x.device))


def _training_path(module):
    """The differentiable path is taken when autograd is on and some parameter wants a gradient."""
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


# ---------------------------------------------------------------------------------------------- DINOv2 (dinov2.py)
class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patches_resolution = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden, bias=True)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim, bias=True)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, init_values):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.ls1 = LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = LayerScale(dim, init_values)


def _trunc_normal(t, std):
    return nn.init.trunc_normal_(t, std=std)


class DinoVisionTransformer(nn.Module):
    """dinov2.py:37-330 (block_chunks=0, no register tokens, LayerScale init 1.0, MLP FFN)."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, init_values=1.0,
                 interpolate_offset=0.1):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens, self.n_blocks, self.num_heads, self.patch_size = 1, depth, num_heads, patch_size
        self.num_register_tokens, self.interpolate_antialias, self.interpolate_offset = 0, False, interpolate_offset
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.register_tokens = None
        self.chunked_blocks = False
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        # dinov2.py:176-181,326-331
        _trunc_normal(self.pos_embed, 0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                _trunc_normal(m.weight, 0.02)
                nn.init.zeros_(m.bias)
        self._prep, self._prep_sig = None, None

    # ---- weight preparation: GEMM operand layouts, LayerScale folded into proj / fc2 --------------------------------
    def _signature(self):
        return (sum(p._version for p in self.parameters()), next(self.parameters()).data_ptr())

    def _prepare(self):
        sig = self._signature()
        if self._prep is not None and self._prep_sig == sig:
            return self._prep
        with torch.no_grad():
            P = self.patch_size
            K = 3 * P * P
            Kp = (K + 31) // 32 * 32                        # pad the patch rows to whole 32-deep GEMM stages
            w = self.patch_embed.proj.weight.reshape(self.embed_dim, K)
            wpe = torch.zeros(self.embed_dim, Kp, device=w.device, dtype=torch.float32)
            wpe[:, :K] = w
            prep = {"kp": Kp, "patch_w": wpe.view(self.embed_dim, Kp, 1, 1).contiguous(memory_format=CL),
                    "patch_b": self.patch_embed.proj.bias.detach().contiguous(), "blocks": []}
            as4 = lambda t: t.detach().reshape(t.shape[0], t.shape[1], 1, 1).contiguous(memory_format=CL)
            for blk in self.blocks:
                g1, g2 = blk.ls1.gamma.detach(), blk.ls2.gamma.detach()
                prep["blocks"].append({
                    "qkv_w": as4(blk.attn.qkv.weight), "qkv_b": blk.attn.qkv.bias.detach().contiguous(),
                    # x + ls1(proj(a)) = x + (gamma * W) a + gamma * b: the LayerScale rides in the weights, the residual in the epilogue
                    "proj_w": as4(blk.attn.proj.weight * g1[:, None]), "proj_b": (blk.attn.proj.bias * g1).detach().contiguous(),
                    "fc1_w": as4(blk.mlp.fc1.weight), "fc1_b": blk.mlp.fc1.bias.detach().contiguous(),
                    "fc2_w": as4(blk.mlp.fc2.weight * g2[:, None]), "fc2_b": (blk.mlp.fc2.bias * g2).detach().contiguous()})
        self._prep, self._prep_sig = prep, sig
        return prep

    def interpolate_pos_encoding(self, npatch, w, h):
        """dinov2.py:183-213: the table itself at the native 37x37 grid, bicubic resampling otherwise (parameter
        preparation, once per input size -- torch's interpolate, not part of the per-frame path)."""
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed[0]
        key = (w, h, self.pos_embed._version)
        cacheable = not _training_path(self)
        if cacheable and getattr(self, "_pos_key", None) == key:
            return self._pos_cache
        pos = self.pos_embed.float()
        dim = pos.shape[-1]
        w0, h0 = w // self.patch_size + self.interpolate_offset, h // self.patch_size + self.interpolate_offset
        sq = math.sqrt(N)
        patch = nn.functional.interpolate(pos[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2),
                                          scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic",
                                          antialias=self.interpolate_antialias)
        assert int(w0) == patch.shape[-2] and int(h0) == patch.shape[-1]
        out = torch.cat((pos[:, 0], patch.permute(0, 2, 3, 1).reshape(-1, dim)), 0).contiguous()
        if cacheable:
            self._pos_key, self._pos_cache = key, out.detach()
        return out

    def prepare_tokens(self, x):
        """dinov2.py:215-235 -> [B*(Np+1), C] tokens."""
        B, _, H, W = x.shape
        P, C = self.patch_size, self.embed_dim
        if H % P or W % P:
            raise _lib.DvsError("DINOv2: image size %dx%d is not a multiple of the patch size %d" % (H, W, P))
        prep = self._prepare()
        Np = (H // P) * (W // P)
        x = x if x.is_contiguous() else x.contiguous()
        rows = torch.empty(B * Np, prep["kp"], device=x.device, dtype=torch.float32)
        l = _lib.lib()
        check(l.dvs_vit_patchify(ptr(x), ptr(rows), B, H, W, P, prep["kp"], _lib.stream()), "dvs_vit_patchify")
        tok = gemm(rows, prep["patch_w"], prep["patch_b"])
        pos = self.interpolate_pos_encoding(Np, H, W).contiguous()
        out = torch.empty(B * (Np + 1), C, device=x.device, dtype=torch.float32)
        check(l.dvs_vit_assemble(ptr(tok), ptr(self.cls_token.detach().reshape(-1).contiguous()), ptr(pos), ptr(out), B, Np, C, _lib.stream()),
              "dvs_vit_assemble")
        return out, B, Np + 1

    def _block(self, x, blk, w, B, N):
        """block.py:82-107 (eval): x + ls1(attn(norm1(x))); x + ls2(mlp(norm2(x)))."""
        h = layernorm(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        qkv = gemm(h, w["qkv_w"], w["qkv_b"])
        a = attention(qkv, B, N, self.num_heads, self.embed_dim // self.num_heads)
        x = gemm(a, w["proj_w"], w["proj_b"], residual=x)
        h = layernorm(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = gemm(h, w["fc1_w"], w["fc1_b"], act="gelu")
        return gemm(h, w["fc2_w"], w["fc2_b"], residual=x)

    # ---- trainable path (autograd on): the same kernels as autograd Functions ------------------------------------------
    def _prepare_tokens_train(self, x):
        B, _, H, W = x.shape
        P, C = self.patch_size, self.embed_dim
        if H % P or W % P:
            raise _lib.DvsError("DINOv2: image size %dx%d is not a multiple of the patch size %d" % (H, W, P))
        K = 3 * P * P
        Kp = (K + 31) // 32 * 32
        Np = (H // P) * (W // P)
        x = x.detach()
        x = x if x.is_contiguous() else x.contiguous()
        rows = torch.empty(B * Np, Kp, device=x.device, dtype=torch.float32)
        check(_lib.lib().dvs_vit_patchify(ptr(x), ptr(rows), B, H, W, P, Kp, _lib.stream()), "dvs_vit_patchify")
        w = nn.functional.pad(self.patch_embed.proj.weight.reshape(C, K), (0, Kp - K))      # tiny; autograd carries dW back
        tok = linear_train(rows, w, self.patch_embed.proj.bias).view(B, Np, C)
        pos = self.interpolate_pos_encoding(Np, H, W)
        if pos.dim() == 2:
            pos = pos.unsqueeze(0)
        # cat + add: two small elementwise launches whose backward (slice, batch sum) autograd already has
        xt = torch.cat((self.cls_token.expand(B, -1, -1), tok), 1) + pos
        return xt.reshape(B * (Np + 1), C), B, Np + 1

    def _block_train(self, x, blk, B, N):
        C, heads = self.embed_dim, self.num_heads
        h = _LayerNormF.apply(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        qkv = linear_train(h, blk.attn.qkv.weight, blk.attn.qkv.bias)
        a = _AttentionF.apply(qkv, B, N, heads, C // heads)
        g1, g2 = blk.ls1.gamma, blk.ls2.gamma
        # LayerScale rides in the weights here too (gamma * W, gamma * b are [C,C] / [C] products; autograd splits the
        # gradient between gamma and W), so no [M,C] pass is spent on it
        x = x + linear_train(a, blk.attn.proj.weight * g1[:, None], blk.attn.proj.bias * g1)
        h = _LayerNormF.apply(x, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = _ActF.apply(linear_train(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias), "gelu")
        return x + linear_train(h, blk.mlp.fc2.weight * g2[:, None], blk.mlp.fc2.bias * g2)

    def get_intermediate_layers(self, x, n=1, reshape=False, return_class_token=False, norm=True):
        """dinov2.py:297-321."""
        _require_gpu(x, "DINOv2")
        train = _training_path(self)
        take = list(range(len(self.blocks) - n, len(self.blocks))) if isinstance(n, int) else list(n)
        outs = []
        if train:
            tok, B, N = self._prepare_tokens_train(x)
            for i, blk in enumerate(self.blocks):
                tok = self._block_train(tok, blk, B, N)
                if i in take:
                    outs.append(tok)
        else:
            prep = self._prepare()
            tok, B, N = self.prepare_tokens(x)
            for i, blk in enumerate(self.blocks):
                tok = self._block(tok, blk, prep["blocks"][i], B, N)
                if i in take:
                    outs.append(tok)
        assert len(outs) == len(take)
        if norm and train:
            outs = [_LayerNormF.apply(o, self.norm.weight, self.norm.bias, self.norm.eps) for o in outs]
        elif norm:
            outs = [layernorm(o, self.norm.weight, self.norm.bias, self.norm.eps) for o in outs]
        outs = [o.view(B, N, self.embed_dim) for o in outs]
        cls = [o[:, 0] for o in outs]
        outs = [o[:, 1:] for o in outs]
        if reshape:
            _, _, H, W = x.shape
            outs = [o.reshape(B, H // self.patch_size, W // self.patch_size, -1).permute(0, 3, 1, 2).contiguous() for o in outs]
        return tuple(zip(outs, cls)) if return_class_token else tuple(outs)


_VITS = {"vits": dict(embed_dim=384, depth=12, num_heads=6), "vitb": dict(embed_dim=768, depth=12, num_heads=12),
         "vitl": dict(embed_dim=1024, depth=24, num_heads=16)}


def DINOv2(model_name):
    """dinov2.py:395-414."""
    if model_name not in _VITS:
        raise NotImplementedError("DINOv2(%r): the SwiGLU ViT-G encoder is not on the MI355X path" % model_name)
    return DinoVisionTransformer(img_size=518, patch_size=14, init_values=1.0, interpolate_offset=0.1, **_VITS[model_name])


# ---------------------------------------------------------------------------------------------- DPT head (dpt.py, util/blocks.py)
class ResidualConvUnit(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)

    def forward(self, x, ones, zeros):
        """blocks.py:63-83: conv2(relu(conv1(relu(x)))) + x -- the leading ReLU is applied in conv1's gather (the skip needs
        the un-activated x), the second one in conv1's epilogue, the skip add in conv2's."""
        if torch.is_grad_enabled() and (x.requires_grad or self.conv1.weight.requires_grad):
            out = _conv.conv2d(_ActF.apply(x, "relu"), self.conv1.weight, self.conv1.bias, 1, 1, 0, "relu")
            return _conv.conv2d(out, self.conv2.weight, self.conv2.bias, 1, 1, 0, None) + x
        out = _conv.conv2d_forward(x, self.conv1.weight, self.conv1.bias, 1, 1, act="relu", in_scale=ones, in_shift=zeros, in_relu=True)
        return _conv.conv2d_forward(out, self.conv2.weight, self.conv2.bias, 1, 1, residual=x)


class FeatureFusionBlock(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)
        self.size = None

    def forward(self, ones, zeros, *xs, size=None):
        """blocks.py:119-147."""
        out = xs[0]
        if len(xs) == 2:
            # output + resConfUnit1(xs[1]): the sum rides in conv2's epilogue too (residual = xs[1] + output is not
            # available there, so add the unit's result to `output` with a second residual pass: conv2(...) + xs[1] first)
            res = self.resConfUnit1(xs[1], ones, zeros)
            out = _add(out, res)
        out = self.resConfUnit2(out, ones, zeros)
        H, W = (size if size is not None else (out.shape[2] * 2, out.shape[3] * 2))
        if torch.is_grad_enabled() and out.requires_grad:
            out = _ResizeF.apply(out, int(H), int(W))
            return _conv.conv2d(out, self.out_conv.weight, self.out_conv.bias, 1, 0, 0, None)
        out = resize_bilinear_ac(out, int(H), int(W))
        return _conv.conv2d_forward(out, self.out_conv.weight, self.out_conv.bias, 1, 0)


def _add(a, b):
    """a + b for two NHWC maps (one elementwise pass; plumbing between two fused convolutions)."""
    return torch.add(a, b)


class _Scratch(nn.Module):
    pass


class DPTHead(nn.Module):
    """dpt.py:38-149 (use_bn=False, use_clstoken=False: what DepthAnythingV2 constructs)."""

    def __init__(self, in_channels, features=256, use_bn=False, out_channels=(256, 512, 1024, 1024), use_clstoken=False):
        super().__init__()
        if use_bn or use_clstoken:
            raise NotImplementedError("DPTHead: use_bn / use_clstoken are not used by DepthAnythingV2 and not built")
        self.use_clstoken = False
        oc = list(out_channels)
        self.projects = nn.ModuleList([nn.Conv2d(in_channels, c, 1, 1, 0) for c in oc])
        self.resize_layers = nn.ModuleList([nn.ConvTranspose2d(oc[0], oc[0], 4, 4, 0), nn.ConvTranspose2d(oc[1], oc[1], 2, 2, 0),
                                            nn.Identity(), nn.Conv2d(oc[3], oc[3], 3, 2, 1)])
        s = _Scratch()
        s.layer1_rn = nn.Conv2d(oc[0], features, 3, 1, 1, bias=False)
        s.layer2_rn = nn.Conv2d(oc[1], features, 3, 1, 1, bias=False)
        s.layer3_rn = nn.Conv2d(oc[2], features, 3, 1, 1, bias=False)
        s.layer4_rn = nn.Conv2d(oc[3], features, 3, 1, 1, bias=False)
        s.stem_transpose = None
        s.refinenet1, s.refinenet2 = FeatureFusionBlock(features), FeatureFusionBlock(features)
        s.refinenet3, s.refinenet4 = FeatureFusionBlock(features), FeatureFusionBlock(features)
        s.output_conv1 = nn.Conv2d(features, features // 2, 3, 1, 1)
        s.output_conv2 = nn.Sequential(nn.Conv2d(features // 2, 32, 3, 1, 1), nn.ReLU(True), nn.Conv2d(32, 1, 1, 1, 0), nn.Sigmoid())
        self.scratch = s
        self.features = features
        self.to(memory_format=CL)
        self._prep, self._prep_sig = None, None

    def _prepare(self):
        sig = (sum(p._version for p in self.parameters()), next(self.parameters()).data_ptr())
        if self._prep is not None and self._prep_sig == sig:
            return self._prep
        with torch.no_grad():
            dev = next(self.parameters()).device
            prep = {"ones": torch.ones(self.features, device=dev), "zeros": torch.zeros(self.features, device=dev), "deconv": []}
            for i, k in ((0, 4), (1, 2)):
                m = self.resize_layers[i]
                ci, co = m.weight.shape[0], m.weight.shape[1]
                # [ci, co, a, c] -> rows (a*k + c)*co + co_index of a [k*k*co, ci] GEMM weight
                wg = m.weight.detach().permute(2, 3, 1, 0).reshape(k * k * co, ci, 1, 1).contiguous(memory_format=CL)
                prep["deconv"].append((wg, m.bias.detach().repeat(k * k).contiguous(), k, co))
        self._prep, self._prep_sig = prep, sig
        return prep

    def _forward_train(self, out_features, patch_h, patch_w):
        """dpt.py:116-149 with autograd: every convolution through conv.conv2d (forward, data- and weight-gradient kernels)."""
        out = []
        for i, x in enumerate(out_features):
            x = x[0]
            B, Np, C = x.shape
            x = x.reshape(B, patch_h, patch_w, C).permute(0, 3, 1, 2)
            x = x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)
            p = self.projects[i]
            x = _conv.conv2d(x, p.weight, p.bias, 1, 0, 0, None)
            if i < 2:
                m = self.resize_layers[i]
                k, co = m.kernel_size[0], m.weight.shape[1]
                wg = m.weight.permute(2, 3, 1, 0).reshape(k * k * co, m.weight.shape[0], 1, 1)
                x = _ShuffleF.apply(_conv.conv2d(x, wg, m.bias.repeat(k * k), 1, 0, 0, None), k, co)
            elif i == 3:
                r = self.resize_layers[3]
                x = _conv.conv2d(x, r.weight, r.bias, 2, 1, 0, None)
            out.append(x)
        s = self.scratch
        l1, l2, l3, l4 = (_conv.conv2d(o, m.weight, None, 1, 1, 0, None) for o, m in zip(out, (s.layer1_rn, s.layer2_rn, s.layer3_rn, s.layer4_rn)))
        path_4 = s.refinenet4(None, None, l4, size=l3.shape[2:])
        path_3 = s.refinenet3(None, None, path_4, l3, size=l2.shape[2:])
        path_2 = s.refinenet2(None, None, path_3, l2, size=l1.shape[2:])
        path_1 = s.refinenet1(None, None, path_2, l1)
        o = _conv.conv2d(path_1, s.output_conv1.weight, s.output_conv1.bias, 1, 1, 0, None)
        o = _ResizeF.apply(o, int(patch_h * 14), int(patch_w * 14))
        c0, c2 = s.output_conv2[0], s.output_conv2[2]
        o = _conv.conv2d(o, c0.weight, c0.bias, 1, 1, 0, "relu")
        return _conv.head_conv2d(o, c2.weight, c2.bias, 0, 0, "sigmoid")

    def forward(self, out_features, patch_h, patch_w):
        if torch.is_grad_enabled() and (out_features[0][0].requires_grad or _training_path(self)):
            return self._forward_train(out_features, patch_h, patch_w)
        prep = self._prepare()
        ones, zeros = prep["ones"], prep["zeros"]
        out = []
        for i, x in enumerate(out_features):
            x = x[0]                                                # [B, Np, C] patch tokens; the class token is not used
            B, Np, C = x.shape
            x = x.reshape(B, patch_h, patch_w, C).permute(0, 3, 1, 2)       # NHWC memory already: tokens are pixels
            x = x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)
            p = self.projects[i]
            x = _conv.conv2d_forward(x, p.weight, p.bias, 1, 0)
            if i < 2:
                x = conv_transpose_s(x, *prep["deconv"][i])
            elif i == 3:
                r = self.resize_layers[3]
                x = _conv.conv2d_forward(x, r.weight, r.bias, 2, 1)
            out.append(x)
        s = self.scratch
        l1, l2, l3, l4 = (_conv.conv2d_forward(o, m.weight, None, 1, 1) for o, m in zip(out, (s.layer1_rn, s.layer2_rn, s.layer3_rn, s.layer4_rn)))
        path_4 = s.refinenet4(ones, zeros, l4, size=l3.shape[2:])
        path_3 = s.refinenet3(ones, zeros, path_4, l3, size=l2.shape[2:])
        path_2 = s.refinenet2(ones, zeros, path_3, l2, size=l1.shape[2:])
        path_1 = s.refinenet1(ones, zeros, path_2, l1)
        o = _conv.conv2d_forward(path_1, s.output_conv1.weight, s.output_conv1.bias, 1, 1)
        o = resize_bilinear_ac(o, int(patch_h * 14), int(patch_w * 14))
        c0, c2 = s.output_conv2[0], s.output_conv2[2]
        o = _conv.conv2d_forward(o, c0.weight, c0.bias, 1, 1, act="relu")
        return _conv.head_conv2d(o, c2.weight, c2.bias, 0, 0, "sigmoid")


class DepthAnythingV2(nn.Module):
    """dpt.py:154-222."""

    def __init__(self, encoder="vitl", features=256, out_channels=(256, 512, 1024, 1024), use_bn=False, use_clstoken=False,
                 max_depth=20.0):
        super().__init__()
        self.intermediate_layer_idx = {"vits": [2, 5, 8, 11], "vitb": [2, 5, 8, 11], "vitl": [4, 11, 17, 23], "vitg": [9, 19, 29, 39]}
        self.max_depth = max_depth
        self.encoder = encoder
        self.pretrained = DINOv2(model_name=encoder)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn, out_channels=out_channels, use_clstoken=use_clstoken)

    def forward(self, x):
        _require_gpu(x, "DepthAnythingV2")
        patch_h, patch_w = x.shape[-2] // 14, x.shape[-1] // 14
        feats = self.pretrained.get_intermediate_layers(x, self.intermediate_layer_idx[self.encoder], return_class_token=True)
        depth = self.depth_head(feats, patch_h, patch_w) * self.max_depth
        return depth.squeeze(1)

    def disp_outputs(self, x, scales=(0,)):
        """Adapter to the VO path's contract (`{("disp", s): [B,1,H/2^s,W/2^s]}`, model/depthnet.py:64-90): the head's
        sigmoid output IS a disparity-like map in (0, 1); coarser scales are its align_corners resizes."""
        d = (self.forward(x) / self.max_depth).unsqueeze(1)
        out = {("disp", 0): d}
        for s in scales:
            if s:
                out[("disp", s)] = nn.functional.interpolate(d, scale_factor=1.0 / (2 ** s), mode="bilinear", align_corners=True)
        return out


class DepthAnythingDispNet(nn.Module):
    """The "encoder swap" of BASELINE configs[4]: a DepthNet-shaped wrapper (`forward(x [B,3,H,W] in [0,1]) ->
    {("disp", s): [B,1,H/2^s,W/2^s]}`, model/depthnet.py:64-90) around DepthAnythingV2, so that MonodepthTrainer / vo/train.py
    can train it in place of the ResNet DepthNet.  Resize policy: the frame is resampled (bilinear, align_corners=True) to
    the largest multiples of 14 that fit (480x640 -> 476x630 = 34x45 patches), normalised with the ImageNet statistics the
    reference's `image2tensor` applies (dpt.py:214-216), and the head's sigmoid map is resampled back to H/2^s x W/2^s."""

    def __init__(self, encoder="vits", features=64, out_channels=(48, 96, 192, 384), scales=range(4)):
        super().__init__()
        self.net = DepthAnythingV2(encoder=encoder, features=features, out_channels=list(out_channels))
        self.scales = list(scales)
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)

    def forward(self, x):
        _require_gpu(x, "DepthAnythingDispNet")
        B, _, H, W = x.shape
        h14, w14 = H // 14 * 14, W // 14 * 14
        xin = x
        if (h14, w14) != (H, W):
            xin = resize_bilinear_ac(x.contiguous(memory_format=CL), h14, w14).contiguous()
        xin = (xin - self.mean) / self.std
        d = (self.net(xin) / self.net.max_depth).unsqueeze(1)               # the head's sigmoid output, [B,1,h14,w14]
        train = torch.is_grad_enabled() and d.requires_grad
        out = {}
        for s in self.scales:
            hs, ws = H // (2 ** s), W // (2 ** s)
            if (hs, ws) == tuple(d.shape[2:]):
                out[("disp", s)] = d
            else:
                out[("disp", s)] = _ResizeF.apply(d, hs, ws) if train else resize_bilinear_ac(d, hs, ws)
        return out
