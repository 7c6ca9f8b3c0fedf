"""ORACLE (test infrastructure only) -- CPU restatement of the reference's per-sample input preparation.

Only ``tests/`` may import this.  Restates vo/dataset/common.py:38-92: `transforms.ToTensor` (uint8 HWC -> fp32 CHW / 255),
`transforms.ColorJitter(0.3, 0.3, 0.3, 0.2)` applied to the three stacked frames, and the K / inv_K pyramid.

Pinning: ToTensor and the K pyramid are fixed by the reference source (division by 255; numpy pinv).  ColorJitter is
torchvision code -- an un-vendored third-party dependency that is ABSENT from this container and un-pinned by the
reference (requirements.txt names no version) -- so its published tensor algorithm
(torchvision.transforms._functional_tensor: _blend, rgb_to_grayscale with 0.2989/0.587/0.114, _rgb2hsv / _hsv2rgb,
adjust_hue's `(h + f) % 1.0`) is restated here: PARITY UNPINNED for that one operator (no torchvision to run, no fixture in
the reference)."""
import numpy as np
import torch


def pil_resize_bilinear(img_u8, out_h, out_w):
    """PIL's Image.resize((out_w, out_h), Image.BILINEAR) restated on a uint8 array [h,w,3] (Pillow's Resample.c:
    precompute_coeffs + normalize_coeffs_8bpc + the horizontal, then the vertical 8-bit pass, each rounding to uint8).
    PINNED by fixtures produced with Pillow itself (tests/golden/make_golden_pipeline.py -> pil_resize_*.npz)."""
    import math

    def tables(n_in, n_out):
        scale = n_in / n_out
        fs = max(scale, 1.0)
        support = fs
        ksize = int(math.ceil(support)) * 2 + 1
        out = []
        for xx in range(n_out):
            center = (xx + 0.5) * scale
            xmin = max(int(center - support + 0.5), 0)
            xmax = min(int(center + support + 0.5), n_in) - xmin
            w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) / fs)) for x in range(xmax)]
            ww = sum(w)
            k = [(v / ww if ww != 0.0 else v) for v in w]
            out.append((xmin, [int(0.5 + v * (1 << 22)) for v in k]))
        return out

    def one_pass(a, n_out, axis):
        a = np.moveaxis(a, axis, 0).astype(np.int64)
        res = np.empty((n_out,) + a.shape[1:], dtype=np.uint8)
        for o, (lo, ks) in enumerate(tables(a.shape[0], n_out)):
            acc = np.full(a.shape[1:], 1 << 21, dtype=np.int64)
            for j, kv in enumerate(ks):
                acc += a[lo + j] * kv
            res[o] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
        return np.moveaxis(res, 0, axis)

    a = np.asarray(img_u8)
    if a.shape[1] != out_w:
        a = one_pass(a, out_w, 1)
    if a.shape[0] != out_h:
        a = one_pass(a, out_h, 0)
    return a


def to_tensor(frames_u8):
    """[N,H,W,3] uint8 (numpy / tensor) -> [N,3,H,W] fp32 in [0,1]."""
    t = torch.as_tensor(np.asarray(frames_u8))
    return t.permute(0, 3, 1, 2).float().div(255.0)


def _gray(img):
    r, g, b = img.unbind(-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def _rgb2hsv(img):
    r, g, b = img.unbind(-3)
    maxc, minc = torch.max(img, dim=-3).values, torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - s * f), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - s * (1.0 - f)), 0.0, 1.0)
    mask = i.unsqueeze(-3) == torch.arange(6).view(-1, 1, 1)
    a1, a2, a3 = torch.stack((v, q, p, p, t, v), -3), torch.stack((t, v, v, q, p, p), -3), torch.stack((p, p, t, v, v, q), -3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(img.dtype), a4)


def color_jitter(img, order, factor):
    """img [..., 3, H, W] in [0,1]; order: the four adjustment ids in application order (0 brightness, 1 contrast,
    2 saturation, 3 hue); factor[id] the adjustment's factor (hue: shift).  The contrast mean is per image."""
    for op in order:
        if op == 0:
            img = _blend(img, torch.zeros_like(img), float(factor[0]))
        elif op == 1:
            mean = torch.mean(_gray(img), dim=(-3, -2, -1), keepdim=True)
            img = _blend(img, mean, float(factor[1]))
        elif op == 2:
            img = _blend(img, _gray(img), float(factor[2]))
        elif op == 3:
            hsv = _rgb2hsv(img)
            h, s, v = hsv.unbind(-3)
            h = (h + float(factor[3])) % 1.0
            img = _hsv2rgb(torch.stack((h, s, v), dim=-3))
    return img
