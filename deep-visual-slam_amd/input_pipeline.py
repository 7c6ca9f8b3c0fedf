"""Input side of the training path on the GPU.

The reference prepares a sample on CPU workers (vo/dataset/common.py:38-92): PIL decode + resize, `ToTensor`,
`ColorJitter(0.3, 0.3, 0.3, 0.2)` on the three stacked frames with probability 0.5, the K / inv_K pyramid, then the default
collate and one fp32 H2D copy per tensor in `process_batch` (vo/learner_new.py:93-95).  At 34 ms per batch-12 step those
workers are the next bottleneck, so here they only DECODE: uint8 HWC frames travel through pinned staging buffers and PCIe
(a quarter of the fp32 bytes), and ToTensor + ColorJitter run as two HIP kernels behind the copy
(csrc/preprocess.hip), on a copy stream one batch ahead of the training step.

    pipe = Prefetcher(loader_of_u8_batches, device, augment=True)
    for sample in pipe:                     # the reference's `sample` dict, already resident in HBM
        trainer.process_batch(sample)

A batch from the loader is a dict with `frames` uint8 [B,3,H,W,3] (source_left, target, source_right; RGB) and `K` float
[B,4,4] (absolute pixel units at full size); `make_sample` turns it into the reference schema (Appendix B of SURVEY.md).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


class JitterParams:
    """torchvision.transforms.ColorJitter.get_params for a batch: per sample one random order of the four adjustments and
    one factor each (uniform in [1 - x, 1 + x], hue in [-h, h]); `apply[b]` is the reference's coin flip
    (`random.random() < 0.5`, common.py:79-81).  Host-side, numpy Generator."""

    def __init__(self, batch, rng, brightness=0.3, contrast=0.3, saturation=0.3, hue=0.2, p=0.5):
        self.order = np.stack([rng.permutation(4) for _ in range(batch)]).astype(np.int32)
        f = np.empty((batch, 4), dtype=np.float32)
        f[:, 0] = rng.uniform(1 - brightness, 1 + brightness, batch)
        f[:, 1] = rng.uniform(1 - contrast, 1 + contrast, batch)
        f[:, 2] = rng.uniform(1 - saturation, 1 + saturation, batch)
        f[:, 3] = rng.uniform(-hue, hue, batch)
        self.factor = f
        self.apply = rng.random(batch) < p

    def records(self, frames_per_sample=3):
        """[B * frames, 8] int32 view of the kernel's record table (order as ints, factors as float bits): the frames of
        a sample share one record; a sample that is not jittered gets order = -1."""
        order = np.where(self.apply[:, None], self.order, -1).astype(np.int32)
        rec = np.concatenate([order, self.factor.view(np.int32)], 1)
        return np.repeat(rec, frames_per_sample, 0)


def pil_bilinear_tables(in_size, out_size):
    """(bounds int32 [out,2], coef int32 [out,ksize]) of one axis of PIL's Image.resize(..., Image.BILINEAR): the triangle
    filter stretched by the scale factor when shrinking (Pillow's precompute_coeffs), normalised, converted to 22-bit
    integers with round-half-away (normalize_coeffs_8bpc).  Host side, double precision, as Pillow computes them."""
    import math
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + support + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = []
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - v if v < 1.0 else 0.0)
        ww = sum(w)
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            coef[xx, x] = int(-0.5 + k * (1 << 22)) if k < 0 else int(0.5 + k * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, coef


_resize_tables = {}


def resize_u8(frames_u8, out_h, out_w):
    """PIL `img.resize((out_w, out_h), Image.BILINEAR)` of uint8 GPU frames [N,h,w,3] (vo/dataset/common.py:38-44), bit exact:
    horizontal pass, then vertical pass, each rounding to uint8 (an axis whose size does not change is skipped)."""
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise _lib.DvsError("resize_u8: uint8 GPU tensor [N,h,w,3] expected")
    x = frames_u8.contiguous()
    N, h, w, _ = x.shape
    l = _lib.lib()
    for axis, (n_in, n_out) in ((0, (w, out_w)), (1, (h, out_h))):
        if n_in == n_out:
            continue
        key = (n_in, n_out, x.device)
        if key not in _resize_tables:
            b, c = pil_bilinear_tables(n_in, n_out)
            _resize_tables[key] = (torch.from_numpy(b).to(x.device), torch.from_numpy(c).to(x.device))
        b, c = _resize_tables[key]
        cur_h, cur_w = x.shape[1], x.shape[2]
        oh, ow = (cur_h, out_w) if axis == 0 else (out_h, cur_w)
        y = torch.empty(N, oh, ow, 3, device=x.device, dtype=torch.uint8)
        check(l.dvs_resample_u8(ptr(x), ptr(y), ptr(b), ptr(c), c.shape[1], N, cur_h, cur_w, oh, ow, axis, _lib.stream()), "dvs_resample_u8")
        x = y
    return x


def u8_to_f32_planar(frames_u8, bgr=False, out=None):
    """[N,H,W,3] uint8 -> [N,3,H,W] fp32 in [0,1] (ToTensor; bgr=True also swaps to RGB)."""
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8:
        raise _lib.DvsError("u8_to_f32_planar: uint8 GPU tensor expected (got %s on %s)" % (frames_u8.dtype, frames_u8.device))
    frames_u8 = frames_u8.contiguous()
    N, H, W, ch = frames_u8.shape
    if ch != 3:
        raise _lib.DvsError("u8_to_f32_planar: [N,H,W,3] expected")
    if out is None:
        out = torch.empty(N, 3, H, W, device=frames_u8.device, dtype=torch.float32)
    check(_lib.lib().dvs_u8_to_f32_planar(ptr(frames_u8), ptr(out), N, H, W, int(bool(bgr)), _lib.stream()), "dvs_u8_to_f32_planar")
    return out


def color_jitter_(images, records):
    """In-place ColorJitter of fp32 [N,3,H,W]; records: int32 [N,8] (JitterParams.records) on the host or the device."""
    N, ch, H, W = images.shape
    if ch != 3 or images.dtype != torch.float32 or not images.is_cuda or not images.is_contiguous():
        raise _lib.DvsError("color_jitter_: contiguous fp32 GPU tensor [N,3,H,W] expected")
    rec = torch.as_tensor(records, dtype=torch.int32)
    if tuple(rec.shape) != (N, 8):
        raise _lib.DvsError("color_jitter_: records must be [N,8] int32")
    rec = rec.to(images.device, non_blocking=True).contiguous()
    l = _lib.lib()
    ws = torch.empty(l.dvs_color_jitter_workspace(N, H, W) // 4, device=images.device, dtype=torch.float32)
    check(l.dvs_color_jitter(ptr(images), ptr(rec), ptr(ws), N, H, W, _lib.stream()), "dvs_color_jitter")
    return images


def intrinsics_pyramid(K, h, w, num_scales=4):
    """K / inv_K for s = 0..3 exactly as vo/dataset/common.py:65-75 (row scaling, pinv in float64 -> float32);
    K: [B,4,4] numpy or tensor in absolute pixel units at (h, w)."""
    K = np.asarray(K, dtype=np.float32)
    out = {}
    for s in range(num_scales):
        wn, hn = w // (2 ** s), h // (2 ** s)
        Ks = K.copy()
        Ks[:, 0, :] *= wn / w
        Ks[:, 1, :] *= hn / h
        inv = np.stack([np.linalg.pinv(k) for k in Ks])
        out[("K", s)] = torch.from_numpy(Ks).float()
        out[("inv_K", s)] = torch.from_numpy(inv).float()
    return out


def make_sample(frames_u8, K, records=None, bgr=False, image_size=None):
    """GPU sample dict of the reference's schema from uint8 frames [B,3,h,w,3] already on the device; image_size = (H, W)
    resizes them first as the reference's loader does (PIL bilinear, common.py:38-44; K is in pixels of the RESIZED image,
    as the reference's intrinsics are)."""
    B, F, H, W, _ = frames_u8.shape
    flat = frames_u8.view(B * F, H, W, 3)
    if image_size is not None and tuple(image_size) != (H, W):
        H, W = image_size
        flat = resize_u8(flat, H, W)
    imgs = u8_to_f32_planar(flat, bgr=bgr)
    if records is not None:
        color_jitter_(imgs, records)
    imgs = imgs.view(B, F, 3, H, W)
    sample = {k: v.to(frames_u8.device, non_blocking=True) for k, v in intrinsics_pyramid(K, H, W).items()}
    sample[("source_left", 0)], sample[("target_image", 0)], sample[("source_right", 0)] = imgs[:, 0], imgs[:, 1], imgs[:, 2]
    return sample


class Prefetcher:
    """Double-buffered H2D + GPU preprocessing one batch ahead of the consumer.

    `batches`: iterable of {"frames": uint8 [B,3,H,W,3] (numpy or CPU tensor), "K": [B,4,4]}.  Each batch is copied into
    one of two pinned staging buffers, sent with one async copy on a private stream, converted and jittered there, and
    handed out with an event the consumer's stream waits on -- the training step never waits for PCIe or for the
    augmentation unless the loader itself is late."""

    def __init__(self, batches, device, augment=True, seed=0, bgr=False, image_size=None):
        self.batches, self.device, self.augment, self.bgr = batches, torch.device(device), augment, bgr
        self.image_size = image_size
        self.rng = np.random.default_rng(seed)
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned = [None, None]
        self._slot = 0

    def _stage(self, batch):
        frames = torch.as_tensor(batch["frames"])
        if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[-1] != 3:
            raise _lib.DvsError("Prefetcher: frames must be uint8 [B,3,H,W,3]")
        slot = self._slot
        self._slot ^= 1
        pin = self._pinned[slot]
        if pin is None or pin[0].shape != frames.shape:
            pin = (torch.empty(frames.shape, dtype=torch.uint8).pin_memory(), torch.cuda.Event())
            self._pinned[slot] = pin
        else:
            pin[1].synchronize()                 # the copy that last read this staging buffer has finished
        pin[0].copy_(frames)
        rec = JitterParams(frames.shape[0], self.rng).records(frames.shape[1]) if self.augment else None
        with torch.cuda.stream(self.stream):
            dev = pin[0].to(self.device, non_blocking=True)
            pin[1].record(self.stream)
            sample = make_sample(dev, batch["K"], rec, bgr=self.bgr, image_size=self.image_size)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return sample, ready, dev

    def __iter__(self):
        it = iter(self.batches)
        nxt = None
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            sample, ready, dev = nxt
            try:
                nxt = self._stage(next(it))      # issue the next batch's copy + preprocessing before handing this one out
            except StopIteration:
                nxt = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ready)
            for t in list(sample.values()) + [dev]:
                t.record_stream(cur)
            yield sample
