"""ORACLE (test infrastructure only) -- CPU restatement of Depth-Anything-V2 (DINOv2 ViT encoder + DPT head).

Only ``tests/`` and ``bench.py``'s cpu_baseline leg may import this.  Plain torch-CPU functional ops on a reference-layout
``state_dict`` (keys ``pretrained.*`` / ``depth_head.*``), restating

  * model/depth_anything_v2/dinov2.py:183-235,297-321  (pos-embed interpolation, token preparation, get_intermediate_layers)
  * dinov2_layers/block.py:82-107, attention.py:49-62, mlp.py:35-41, layer_scale.py:27, patch_embed.py:69-82
  * dpt.py:116-149,192-199 (DPTHead.forward, DepthAnythingV2.forward), util/blocks.py:63-83,119-147

Pinning: the ENCODER is PINNED -- tests/golden/make_golden_dav2.py imports the reference's dinov2.py (torch only) and stores
`get_intermediate_layers` outputs for seeded weights (tests/golden/dav2_*.npz, tests/test_oracle_golden.py); the DPT head's
building blocks (ResidualConvUnit, FeatureFusionBlock, _make_scratch) are PINNED the same way from util/blocks.py.
DPTHead.forward's own glue (project -> resize -> fuse -> output convs) is restated from dpt.py:116-149 and PARITY UNPINNED
by reference-run vectors: dpt.py imports cv2 and torchvision, both absent from the container.
"""
import math

import torch
import torch.nn.functional as F


def seeded_weights(template, seed=0):
    """Deterministic, key-ordered weights for any state_dict-shaped template: {key: tensor} with the template's shapes.
    The golden generator fills the REFERENCE model with these and the tests fill the product's modules -- no checkpoint
    has to travel.  Scales keep activations O(1) through 12 blocks."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template):
        shape = tuple(template[k].shape)
        r = torch.randn(shape, generator=g)
        if k.endswith("gamma") or (".norm" in k and k.endswith("weight")) or k.endswith("norm.weight"):
            out[k] = 1.0 + 0.1 * r
        elif k.endswith("bias") or "token" in k or "pos_embed" in k:
            out[k] = 0.02 * r
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[k] = r * (1.0 / max(fan_in, 1)) ** 0.5
    return out


def interpolate_pos_encoding(pos_embed, npatch, w, h, patch=14, offset=0.1):
    """dinov2.py:183-213."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    dim = pos_embed.shape[-1]
    w0, h0 = w // patch + offset, h // patch + offset
    sq = math.sqrt(N)
    p = F.interpolate(pos_embed[:, 1:].reshape(1, int(sq), int(sq), dim).permute(0, 3, 1, 2),
                      scale_factor=(float(w0) / sq, float(h0) / sq), mode="bicubic", antialias=False)
    assert int(w0) == p.shape[-2] and int(h0) == p.shape[-1]
    return torch.cat((pos_embed[:, :1], p.permute(0, 2, 3, 1).reshape(1, -1, dim)), 1)


def _block(x, sd, p, heads):
    """block.py:82-107 in eval mode."""
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (a @ v).transpose(1, 2).reshape(B, N, C)
    a = F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    x = x + a * sd[p + "ls1.gamma"]
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h * sd[p + "ls2.gamma"]


def dinov2_intermediate(x, sd, taps, heads, prefix="pretrained.", norm=True):
    """get_intermediate_layers(x, taps, return_class_token=True): tuple of (patch tokens [B,Np,C], class token [B,C])."""
    B, _, H, W = x.shape
    t = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=14).flatten(2).transpose(1, 2)
    t = torch.cat((sd[prefix + "cls_token"].expand(B, -1, -1), t), 1)
    t = t + interpolate_pos_encoding(sd[prefix + "pos_embed"], t.shape[1] - 1, H, W)
    depth = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith(prefix + "blocks."))
    outs = []
    for i in range(depth):
        t = _block(t, sd, "%sblocks.%d." % (prefix, i), heads)
        if i in taps:
            outs.append(t)
    if norm:
        C = t.shape[-1]
        outs = [F.layer_norm(o, (C,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-6) for o in outs]
    return tuple((o[:, 1:], o[:, 0]) for o in outs)


def residual_conv_unit(x, sd, p):
    """blocks.py:63-83 (bn=False)."""
    out = F.conv2d(F.relu(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], 1, 1)
    out = F.conv2d(F.relu(out), sd[p + "conv2.weight"], sd[p + "conv2.bias"], 1, 1)
    return out + x


def fusion_block(sd, p, *xs, size=None):
    """blocks.py:119-147 (align_corners=True)."""
    out = xs[0]
    if len(xs) == 2:
        out = out + residual_conv_unit(xs[1], sd, p + "resConfUnit1.")
    out = residual_conv_unit(out, sd, p + "resConfUnit2.")
    kw = {"scale_factor": 2} if size is None else {"size": tuple(size)}
    out = F.interpolate(out, **kw, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def dpt_head(feats, sd, patch_h, patch_w, prefix="depth_head."):
    """dpt.py:116-149 (use_clstoken=False)."""
    out = []
    for i, (x, _cls) in enumerate(feats):
        x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], patch_h, patch_w)
        x = F.conv2d(x, sd["%sprojects.%d.weight" % (prefix, i)], sd["%sprojects.%d.bias" % (prefix, i)])
        if i == 0:
            x = F.conv_transpose2d(x, sd[prefix + "resize_layers.0.weight"], sd[prefix + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, sd[prefix + "resize_layers.1.weight"], sd[prefix + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = F.conv2d(x, sd[prefix + "resize_layers.3.weight"], sd[prefix + "resize_layers.3.bias"], 2, 1)
        out.append(x)
    s = prefix + "scratch."
    l1, l2, l3, l4 = (F.conv2d(o, sd["%slayer%d_rn.weight" % (s, i + 1)], None, 1, 1) for i, o in enumerate(out))
    p4 = fusion_block(sd, s + "refinenet4.", l4, size=l3.shape[2:])
    p3 = fusion_block(sd, s + "refinenet3.", p4, l3, size=l2.shape[2:])
    p2 = fusion_block(sd, s + "refinenet2.", p3, l2, size=l1.shape[2:])
    p1 = fusion_block(sd, s + "refinenet1.", p2, l1)
    o = F.conv2d(p1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], 1, 1)
    o = F.interpolate(o, (int(patch_h * 14), int(patch_w * 14)), mode="bilinear", align_corners=True)
    o = F.relu(F.conv2d(o, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], 1, 1))
    return torch.sigmoid(F.conv2d(o, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"]))


def depth_anything_v2(x, sd, taps=(2, 5, 8, 11), heads=6, max_depth=20.0):
    """dpt.py:192-199 -> depth [B,H,W]."""
    ph, pw = x.shape[-2] // 14, x.shape[-1] // 14
    feats = dinov2_intermediate(x, sd, taps, heads)
    return (dpt_head(feats, sd, ph, pw) * max_depth).squeeze(1)
