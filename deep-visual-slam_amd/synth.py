"""Synthetic frame triplets with the reference's ``sample`` schema.

Schema follows the reference loader (vo/dataset/common.py:48-92): per batch
``("K", s)``, ``("inv_K", s)`` [B,4,4] for s=0..3 and the three fp32 images
``("source_left",0)``, ``("target_image",0)``, ``("source_right",0)`` [B,3,H,W] in [0,1].
Intrinsics are the Redwood defaults (README.md:135-138, vo/dataset/redwood.py:178-182)
rescaled to the requested image size.
"""
import math

import numpy as np
import torch

# Redwood pinhole camera at 640x480 (README.md:135-138)
_FX, _FY, _CX, _CY = 525.0, 525.0, 319.5, 239.5
_W0, _H0 = 640, 480


def intrinsics(batch, h, w, num_scales=4):
    """K / inv_K pyramid exactly as vo/dataset/common.py:65-75 builds it (pinv in float64,
    cast to float32)."""
    out = {}
    K0 = np.eye(4, dtype=np.float32)
    K0[0, 0] = _FX * w / _W0
    K0[1, 1] = _FY * h / _H0
    K0[0, 2] = _CX * w / _W0
    K0[1, 2] = _CY * h / _H0
    for s in range(num_scales):
        wn, hn = w // (2 ** s), h // (2 ** s)
        K = K0.copy()
        K[0, :] *= wn / w
        K[1, :] *= hn / h
        inv_K = np.linalg.pinv(K)
        out[("K", s)] = torch.from_numpy(K).float().unsqueeze(0).repeat(batch, 1, 1).contiguous()
        out[("inv_K", s)] = torch.from_numpy(inv_K).float().unsqueeze(0).repeat(batch, 1, 1).contiguous()
    return out


def _texture(batch, h, w, seed, shift_x=0.0, shift_y=0.0):
    """Smooth procedural texture: 4 low-frequency sinusoids per channel + 0.02 uniform noise."""
    rng = np.random.RandomState(seed)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    xs = xs + shift_x
    ys = ys + shift_y
    img = np.zeros((batch, 3, h, w), dtype=np.float64)
    for b in range(batch):
        for c in range(3):
            acc = np.zeros((h, w))
            for _ in range(4):
                fx = rng.uniform(0.5, 4.0) * 2 * math.pi / w
                fy = rng.uniform(0.5, 4.0) * 2 * math.pi / h
                ph = rng.uniform(0, 2 * math.pi)
                amp = rng.uniform(0.05, 0.12)
                acc += amp * np.sin(fx * xs + fy * ys + ph)
            img[b, c] = 0.5 + acc
    return img


def parity_sample(batch, h, w, seed=2024):
    """Parity set: the two sources are the target texture shifted by
    (+-2.5, +-1.25) px so that warps are well conditioned and argmin ties are rare."""
    rng = np.random.RandomState(seed + 99)
    sample = intrinsics(batch, h, w)
    tgt = _texture(batch, h, w, seed)
    left = _texture(batch, h, w, seed, 2.5, 1.25)
    right = _texture(batch, h, w, seed, -2.5, -1.25)
    for key, img in ((("target_image", 0), tgt), (("source_left", 0), left), (("source_right", 0), right)):
        img = img + 0.02 * rng.uniform(-1, 1, size=img.shape)
        sample[key] = torch.from_numpy(np.clip(img, 0.0, 1.0)).float().contiguous()
    return sample


def throughput_sample(batch, h, w, rank=0, device="cpu"):
    """Throughput set: i.i.d. uniform images from seed 1234 + rank."""
    g = torch.Generator().manual_seed(1234 + rank)
    sample = intrinsics(batch, h, w)
    for key in (("source_left", 0), ("target_image", 0), ("source_right", 0)):
        sample[key] = torch.rand(batch, 3, h, w, generator=g)
    if device != "cpu":
        sample = {k: v.to(device) for k, v in sample.items()}
    return sample


def parity_disps(batch, h, w, seed=5, num_scales=4):
    """Smooth disparity pyramids in (0.05, 0.95) standing in for DepthNet's sigmoid outputs."""
    rng = np.random.RandomState(seed)
    out = []
    for s in range(num_scales):
        hs, ws = h // 2 ** s, w // 2 ** s
        ys, xs = np.meshgrid(np.linspace(0, 1, hs), np.linspace(0, 1, ws), indexing="ij")
        d = np.zeros((batch, 1, hs, ws))
        for b in range(batch):
            d[b, 0] = (0.5 + 0.25 * np.sin(2 * math.pi * (rng.uniform(0.5, 2) * xs + rng.uniform(0, 1)))
                       * np.cos(2 * math.pi * (rng.uniform(0.5, 2) * ys + rng.uniform(0, 1)))
                       + 0.05 * rng.uniform(-1, 1, size=(hs, ws)))
        out.append(torch.from_numpy(np.clip(d, 0.05, 0.95)).float().contiguous())
    return out


def parity_poses(batch, seed=11):
    """Small 6-DoF motions of the magnitude PoseNet emits (0.01 x conv mean,
    model/posenet_single.py:196): (axisangle_left, trans_left, axisangle_right, trans_right),
    each [B,1,1,3]."""
    rng = np.random.RandomState(seed)
    vals = []
    for scale in (0.02, 0.05, 0.02, 0.05):
        vals.append(torch.from_numpy(rng.uniform(-1, 1, size=(batch, 1, 1, 3)) * scale).float())
    return vals
