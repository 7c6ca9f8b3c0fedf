"""Step-scoped pool of zero-initialised scratch tensors.

The split-K weight gradients, BatchNorm statistics and reduction buffers of one training step need ~300
small zero-filled tensors; as separate torch.zeros calls they cost ~1.4 ms of GPU time per step in fill
launches.  The pool hands out views of one arena that is re-zeroed with a single memset when the step
begins (`reset()`, called by MonodepthTrainer.process_batch).  Callers that never reset simply exhaust
the arena and fall back to torch.zeros, so the pool is an optimisation, never a requirement.
"""
import os

import torch

_CAP = 40 * 1024 * 1024          # floats: all weight gradients (26.8 M) + statistics with room to spare
_CHECK = os.environ.get("DVS_POOL_CHECK") == "1"      # debugging: verify at every reset that nothing beyond the handed-out prefix was written


class _Pool:
    def __init__(self):
        self.buf = None
        self.off = 0
        self.active = False

    def reset(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(_CAP, device=device, dtype=torch.float32)
            self.off = 0
        elif self.off or _CHECK:
            if _CHECK:
                torch.cuda.synchronize()
                dirty = int((self.buf[self.off:] != 0).sum())
                if dirty:
                    idx = int((self.buf[self.off:] != 0).nonzero()[0]) + self.off
                    raise RuntimeError("zeropool: %d non-zero floats beyond the handed-out prefix (%d), first at %d" % (dirty, self.off, idx))
            self.buf[:self.off].zero_()
            self.off = 0
        self.active = True

    def zeros(self, shape, device, channels_last=False, pooled=True):
        """`pooled=False` forces a private allocation: required whenever the tensor may outlive the step, e.g. a
        gradient that autograd could adopt as `param.grad` (it only adds in place when `.grad` already exists)."""
        n = 1
        for s in shape:
            n *= s
        n_al = (n + 3) // 4 * 4
        if not pooled or not self.active or self.buf is None or self.buf.device != device or self.off + n_al > _CAP:
            t = torch.zeros(shape, device=device, dtype=torch.float32)
            return t.contiguous(memory_format=torch.channels_last) if channels_last else t
        v = self.buf[self.off:self.off + n]
        self.off += n_al
        if channels_last:                      # logical [Cout,Cin,kh,kw], physical [Cout][kh][kw][Cin]
            co, ci, kh, kw = shape
            return v.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return v.view(shape)


_pool = _Pool()
reset = _pool.reset
zeros = _pool.zeros
