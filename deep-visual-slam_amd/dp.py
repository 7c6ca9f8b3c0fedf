"""Data-parallel training plumbing: one process per GPU, flat fp32 arenas, bucketed gradient
all-reduce over RCCL (torch.distributed backend "nccl" on ROCm), fused Adam on the arena.

The reference has no distributed code at all: `vo/train.py` builds one
`torch.optim.Adam` over both networks on one device.  This module keeps that optimiser's arithmetic
(dvs_adam_step) and adds the only exchange step the path has under data parallelism over frame
triplets: one sum all-reduce of the 26.8 M gradients per step, scaled by 1/world_size inside the
Adam pass.  BatchNorm statistics stay per rank (the reference has no SyncBN).

Layout: every tensor that can receive a gradient lives back to back in `FlatParams.params`
(16-byte aligned slots) with its gradient at the same offset of `FlatParams.grads`; nn.Parameter
`.data` / `.grad` are views, so autograd accumulates straight into the arena.  Slots are ordered by
expected backward completion (PoseNet decoder -> encoder, then DepthNet decoder -> encoder, i.e.
reverse construction order), so the buckets fill front to back and each bucket's all-reduce is
issued while the remaining backward still runs.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib, gradsink
from ._lib import check, ptr


def trainable_parameters(*modules):
    """Named parameters that receive gradients on this path: everything except torchvision's unused
    `fc` head, which the reference keeps in its state_dict but never runs."""
    out = []
    for mi, m in enumerate(modules):
        for n, p in m.named_parameters():
            if p.requires_grad and ".fc." not in n:
                out.append(("%d.%s" % (mi, n), p))
    return out


class FlatParams:
    def __init__(self, named_params, align=4, grad_sinks=True):
        """grad_sinks: let the weight-gradient / BatchNorm kernels accumulate straight into the gradient arena
        (gradsink.py).  Requires training with `loss.backward()`; `torch.autograd.grad` w.r.t. these parameters
        would see None."""
        named_params = list(named_params)[::-1]          # reverse construction order = backward order
        self.names = [n for n, _ in named_params]
        self.tensors = [p for _, p in named_params]
        dev = self.tensors[0].device
        self.offsets = []
        off = 0
        for p in self.tensors:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.params = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(self.tensors, self.offsets):
            n = p.numel()
            self._view(self.params, p, o).copy_(p.data)
            p.data = self._view(self.params, p, o)
            p.grad = self._view(self.grads, p, o)
            p._dvs_arena = True                  # MonodepthTrainer does not build an arena of its own over these
            if grad_sinks:
                gradsink.attach(p, p.grad)

    @staticmethod
    def _view(flat, p, o):
        """Slot view with the parameter's logical shape.  Convolution weights are stored [Cout][kh][kw][Cin]
        (torch channels_last), the [N][K] operand layout of the implicit-GEMM kernels."""
        n = p.numel()
        if p.dim() == 4:
            co, ci, kh, kw = p.shape
            return flat[o:o + n].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grads.zero_()

    def reattach(self):
        """Re-point .grad at the arena (e.g. after a caller ran zero_grad(set_to_none=True))."""
        for p, o in zip(self.tensors, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                p.grad = self._view(self.grads, p, o)
                if getattr(p, "_dvs_sink", None) is not None:
                    p._dvs_sink.grad = p.grad


class RcclComm:
    """Direct RCCL communicator for the gradient arena (include/dvslam_rccl.h: dvs_allreduce_{init,run,destroy}).

    The all-reduce runs on `self.stream`, a HIP stream this object owns -- so it has a hardware queue of its own next to
    the step's compute streams and the caller orders it with stream waits -- instead of the
    internal stream of torch.distributed's process group.  The 128-byte rendezvous token is created by rank 0 and
    broadcast over the torch.distributed group that already exists (any backend)."""

    def __init__(self, device, group=None):
        from . import _rccl
        self._rccl = _rccl
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        l = _rccl.lib()
        token = [None]
        if self.rank == 0:
            buf = (C.c_ubyte * _rccl.UNIQUE_ID_BYTES)()
            _rccl.check(l.dvs_allreduce_unique_id(buf), "dvs_allreduce_unique_id")
            token[0] = bytes(buf)
        if self.world > 1:
            # `src` is a GLOBAL rank: with a sub-group the group's rank 0 is not rank 0 of the world
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(token, src=src, group=group)
        self.device = torch.device(device)
        with torch.cuda.device(self.device):
            handle = C.c_void_p()
            idbuf = (C.c_ubyte * _rccl.UNIQUE_ID_BYTES).from_buffer_copy(token[0])
            _rccl.check(l.dvs_allreduce_init(C.byref(handle), idbuf, self.world, self.rank), "dvs_allreduce_init")
            self.handle = handle
            self.stream = torch.cuda.Stream(device=self.device)

    def all_reduce_(self, tensor):
        """In-place sum over the ranks, enqueued on self.stream AFTER everything the current stream has enqueued."""
        if not (tensor.is_cuda and tensor.is_contiguous() and tensor.dtype == torch.float32):
            raise _lib.DvsError("RcclComm.all_reduce_: contiguous fp32 GPU tensor expected")
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._rccl.check(self._rccl.lib().dvs_allreduce_run(self.handle, tensor.data_ptr(), tensor.numel(), self.stream.cuda_stream),
                         "dvs_allreduce_run")
        tensor.record_stream(self.stream)

    def all_reduce_ranges_(self, base, ranges):
        """In-place sum of several element ranges [(start, end), ...] of `base` as ONE RCCL group (one launch:
        dvs_allreduce_run_ranges), enqueued on self.stream after everything the current stream has enqueued."""
        if not (base.is_cuda and base.is_contiguous() and base.dtype == torch.float32):
            raise _lib.DvsError("RcclComm.all_reduce_ranges_: contiguous fp32 GPU tensor expected")
        n = len(ranges)
        if n == 0:
            return
        offs = (C.c_size_t * n)(*[int(s) for s, _ in ranges])
        cnts = (C.c_size_t * n)(*[int(e - s) for s, e in ranges])
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._rccl.check(self._rccl.lib().dvs_allreduce_run_ranges(self.handle, base.data_ptr(), offs, cnts, n, self.stream.cuda_stream),
                         "dvs_allreduce_run_ranges")
        base.record_stream(self.stream)

    def wait(self):
        """The current stream waits for every all-reduce enqueued so far."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def close(self):
        if getattr(self, "handle", None) is not None:
            self._rccl.check(self._rccl.lib().dvs_allreduce_destroy(self.handle), "dvs_allreduce_destroy")
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GradSync:
    """Bucketed sum all-reduce of FlatParams.grads, overlapped with backward.  `comm`: an RcclComm for the direct RCCL
    path (DVS_ALLREDUCE=rccl in bench.py); default = torch.distributed's all_reduce on the group's backend."""

    def __init__(self, flat, bucket_bytes=32 << 20, group=None, hook_streams=None, comm=None):
        self.comm = comm
        """hook_streams: optional {id(param): torch.cuda.Stream}.  Autograd creates a parameter's AccumulateGrad node
        -- and binds it to the then-current stream -- when the hook is registered; parameters whose backward runs on
        another stream (PoseNet's, see MonodepthTrainer.pose_stream) should be registered under that stream, otherwise
        every accumulation first synchronises the two streams."""
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = []          # (start, end) element ranges
        self.bucket_of = []
        cap = bucket_bytes // 4
        n = len(flat.tensors)
        ends = [flat.offsets[i + 1] if i + 1 < n else flat.numel for i in range(n)]
        # A bucket is reduced when its LAST gradient exists.  Slots are in backward order per network, and the networks
        # run their backward passes concurrently on their own streams (hook_streams), so (1) no bucket spans two
        # networks -- it would wait for both -- and (2) the tail of each network (its stem: the gradients that only exist
        # at the very end of the step) is a small bucket of its own, so that the all-reduce left exposed after backward
        # moves ~tail_bytes instead of a full bucket.
        def stream_of(i):
            return hook_streams.get(id(flat.tensors[i])) if hook_streams else None
        seg_last = [i for i in range(n) if i + 1 == n or stream_of(i + 1) is not stream_of(i)]
        tail = min(cap // 8, (2 << 20) // 4)
        cuts = set(seg_last)                               # tensor indices after which a bucket ends
        seg_first = 0
        for last in seg_last:
            start_el = flat.offsets[seg_first]
            acc = start_el
            for i in range(seg_first, last + 1):
                if ends[i] - acc >= cap:
                    cuts.add(i)
                    acc = ends[i]
            # split the segment's final tensors (<= tail elements) off the last bucket when that bucket is much larger
            j = last
            while j > seg_first and ends[last] - flat.offsets[j - 1] <= tail and (j - 1) not in cuts:
                j -= 1
            prev_cut = max([c for c in cuts if c < j] + [seg_first - 1])
            if j <= last and j - 1 > prev_cut and ends[j - 1] - (flat.offsets[prev_cut + 1]) >= 4 * tail:
                cuts.add(j - 1)
            seg_first = last + 1
        start = 0
        for i in range(n):
            self.bucket_of.append(len(self.buckets))
            if i in cuts:
                self.buckets.append((start, ends[i]))
                start = ends[i]
        self.sizes = [0] * len(self.buckets)
        for b in self.bucket_of:
            self.sizes[b] += 1
        # The tail bucket of each network (its stem: complete only when the backward pass is) is not reduced from its hook but
        # in finish(), together with the other network's tail: one RCCL group launch for the two instead of two launches that
        # would both sit exposed behind the backward pass.
        self.deferred = set()
        if hook_streams:
            for last in seg_last:
                b = self.bucket_of[last]
                s0, e0 = self.buckets[b]
                if (e0 - s0) <= 2 * tail and len(self.buckets) > len(seg_last):
                    self.deferred.add(b)
        self._ready = [0] * len(self.buckets)
        self._reduced = [False] * len(self.buckets)
        self._work = []
        self._members = [[] for _ in self.buckets]       # parameters of each bucket (for the stream fence)
        for i, p in enumerate(flat.tensors):
            self._members[self.bucket_of[i]].append(p)
        if self.world > 1:
            # fires once per backward, after the last use of the parameter -- also for sunk gradients, whose
            # Functions hand None to autograd (gradsink.py)
            for i, p in enumerate(flat.tensors):
                st = hook_streams.get(id(p)) if hook_streams else None
                if isinstance(st, torch.cuda.Stream):            # (any other key only groups the parameters into segments)
                    with torch.cuda.stream(st):
                        p.register_post_accumulate_grad_hook(self._make_hook(self.bucket_of[i]))
                else:
                    p.register_post_accumulate_grad_hook(self._make_hook(self.bucket_of[i]))

    def _make_hook(self, b):
        def hook(_param):
            self._ready[b] += 1
            if self._ready[b] == self.sizes[b] and not self._reduced[b] and b not in self.deferred:
                s, e = self.buckets[b]
                self._reduced[b] = True
                # the bucket's gradients were written on the compute and side streams of ITS network only: wait for
                # those, not for the other network's backward
                gradsink.fence_for(self._members[b])
                self._reduce(s, e)
        return hook

    def _reduce(self, s, e):
        if self.comm is not None:
            self.comm.all_reduce_(self.flat.grads[s:e])
        else:
            self._work.append(dist.all_reduce(self.flat.grads[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the step's all-reduces (any bucket whose hooks did not all fire is reduced now)."""
        gradsink.fence()                     # gradients accumulated on the side / per-network streams
        if self.world > 1:
            rest = [(s, e) for b, (s, e) in enumerate(self.buckets) if not self._reduced[b]]
            if self.comm is not None and len(rest) > 1 and hasattr(self.comm, "all_reduce_ranges_"):
                self.comm.all_reduce_ranges_(self.flat.grads, rest)        # the deferred tails (and anything whose hooks did not fire)
            else:
                for s, e in rest:
                    self._reduce(s, e)
            for w in self._work:
                w.wait()
            if self.comm is not None:
                self.comm.wait()
        self._work = []
        self._ready = [0] * len(self.buckets)
        self._reduced = [False] * len(self.buckets)

    @property
    def grad_scale(self):
        return 1.0 / self.world


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) semantics on the flat arena (dvs_adam_step): one pass over
    param / grad / exp_avg / exp_avg_sq instead of the reference's foreach Adam over 244 tensors (vo/train.py:114-117,192).

    It IS a torch.optim.Optimizer: LR schedulers (`PolynomialLR(self.optimizer, ...)`, vo/train.py:120-124) accept it and
    drive `param_groups[0]["lr"]`; `state_dict()` / `load_state_dict()` speak torch.optim.Adam's per-parameter layout, so
    the reference's `mono_optimizer_state_dict` checkpoints (vo/train.py:383-390) load here and checkpoints written here
    resume in the reference trainer.  `params` fixes the parameter numbering of that layout: pass what the reference passes
    to Adam, `list(depth_net.parameters()) + list(pose_net.parameters())` (the unused `fc` tensors have no arena slot and,
    as in torch, no state because they never receive a gradient); default = the arena's own order."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, params=None):
        self.flat = flat
        self._slot = {id(p): i for i, p in enumerate(flat.tensors)}
        plist = list(params) if params is not None else list(flat.tensors)
        missing = [i for i, p in enumerate(flat.tensors) if id(p) not in {id(q) for q in plist}]
        if missing:
            raise ValueError("FusedAdam: %d arena tensors are missing from `params`" % len(missing))
        super().__init__([{"params": plist}], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                                                  maximize=False, foreach=None, capturable=False, differentiable=False,
                                                  fused=None))
        self.exp_avg = torch.zeros_like(flat.params)
        self.exp_avg_sq = torch.zeros_like(flat.params)
        self.step_count = 0
        # data-gradient weight packs of the arena's convolutions, refreshed in one launch after every step
        from . import conv as _conv
        self.packed = _conv.PackedWeights(flat.tensors)
        self.packed.repack()

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, zero_grad=False):
        """One Adam update of every arena tensor.  grad_scale: 1/world_size after a sum all-reduce (GradSync.grad_scale);
        zero_grad=True clears the gradient arena in the same pass (the next step's `optimizer.zero_grad()` is then free)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_count += 1
        gradsink.join()                      # weight gradients accumulated on side streams
        g = self.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise _lib.DvsError("FusedAdam: weight_decay / amsgrad / maximize are not implemented (the reference trains "
                                "with plain Adam, vo/train.py:114-117)")
        f = self.flat
        f.reattach()                         # a caller's zero_grad(set_to_none=True) detaches .grad from the arena
        check(_lib.lib().dvs_adam_step(ptr(f.params), ptr(f.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                       f.numel, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], self.step_count,
                                       grad_scale, int(zero_grad), _lib.stream()), "dvs_adam_step")
        from . import nn_ops
        nn_ops.bump_generation()             # the weights changed behind torch's version counters: folded-BN caches are stale
        self.packed.repack()
        return loss

    def zero_grad(self, set_to_none=False):
        """Zero the gradient arena.  `set_to_none` is accepted for signature compatibility and ignored: the gradients are
        persistent views of one arena that the kernels accumulate into (gradient sinks), never freed."""
        gradsink.join()
        self.flat.reattach()
        self.flat.zero_grad()

    def __del__(self):
        try:
            self.packed.release()
        except Exception:
            pass

    # ---- torch.optim.Adam-compatible checkpoints -------------------------------------------------------------
    def state_dict(self):
        """{"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} exactly as torch.optim.Adam over
        `params` would produce (tensors cloned out of the arena in the parameters' logical shapes)."""
        group = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        plist = self.param_groups[0]["params"]
        group["params"] = list(range(len(plist)))
        state = {}
        if self.step_count > 0:
            f = self.flat
            for idx, p in enumerate(plist):
                i = self._slot.get(id(p))
                if i is None:
                    continue                 # e.g. encoder.fc.*: never receives a gradient, so torch keeps no state either
                o = f.offsets[i]
                state[idx] = {"step": torch.tensor(float(self.step_count)),
                              "exp_avg": f._view(self.exp_avg, p, o).detach().clone(memory_format=torch.contiguous_format),
                              "exp_avg_sq": f._view(self.exp_avg_sq, p, o).detach().clone(memory_format=torch.contiguous_format)}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's layout (a reference checkpoint's `mono_optimizer_state_dict`) and this class's
        round-1 flat layout ({"step", "exp_avg", "exp_avg_sq", "param_groups"})."""
        if "state" not in sd:                                    # flat layout
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            for k in ("lr", "betas", "eps"):
                if k in sd["param_groups"][0]:
                    self.param_groups[0][k] = sd["param_groups"][0][k]
            return
        groups = sd["param_groups"]
        if len(groups) != 1:
            raise ValueError("FusedAdam.load_state_dict: expected one parameter group, got %d" % len(groups))
        plist = self.param_groups[0]["params"]
        ids = list(groups[0]["params"])
        if len(ids) != len(plist):
            raise ValueError("FusedAdam.load_state_dict: checkpoint has %d parameters, optimiser has %d" % (len(ids), len(plist)))
        f = self.flat
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for pos, key in enumerate(ids):
            st = sd["state"].get(key)
            if st is None:
                continue
            p = plist[pos]
            i = self._slot.get(id(p))
            if i is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("FusedAdam.load_state_dict: state %s has shape %s, parameter has %s"
                                 % (key, tuple(st["exp_avg"].shape), tuple(p.shape)))
            o = f.offsets[i]
            f._view(self.exp_avg, p, o).copy_(st["exp_avg"])
            f._view(self.exp_avg_sq, p, o).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdam.load_state_dict: parameters with different step counts %s (one fused pass keeps one)" % sorted(steps))
        self.step_count = steps.pop() if steps else 0
        for k, v in groups[0].items():
            if k != "params":
                self.param_groups[0][k] = tuple(v) if k == "betas" else v


def profile_enable(on=True):
    check(_lib.lib().dvs_profile_enable(int(on)), "dvs_profile_enable")


def profile_read():
    """{kernel_name: (total_ms, launches, algorithmic_work)} for every slot that recorded launches."""
    l = _lib.lib()
    out = {}
    for s in range(l.dvs_profile_slots()):
        ms, n, w = C.c_double(), C.c_long(), C.c_double()
        check(l.dvs_profile_read(s, C.byref(ms), C.byref(n)), "dvs_profile_read")
        check(l.dvs_profile_work(s, C.byref(w)), "dvs_profile_work")
        if n.value:
            out[l.dvs_profile_slot_name(s).decode()] = (ms.value, n.value, w.value)
    return out
