"""Gradient sinks: parameter gradients accumulated in place by the kernel that produces them.

With a persistent `.grad` (dp.FlatParams keeps every gradient in one flat arena that the fused Adam pass
re-zeroes), autograd's AccumulateGrad is one zero-fill of a scratch tensor plus one `grad += scratch` launch
per parameter use: ~360 five-microsecond launches per training step (2.2 ms of GPU time, more on the
host).  The split-K weight-gradient kernels already accumulate with atomics and the BatchNorm backward
owns its channel sums, so they can add straight into `.grad` instead; the autograd Function then returns
None for that parameter.

A sink is opt-in per parameter (`attach`, done by dp.FlatParams) because it changes what
`torch.autograd.grad` sees: sunk gradients only ever reach `.grad`, i.e. train with `loss.backward()`.
"Gradient complete" notifications need nothing extra: autograd runs a parameter's AccumulateGrad node -- and
with it the post-accumulate-grad hooks dp.GradSync registers -- after the last Function that uses the parameter
has returned, also when every one of them returned None (tests/test_dp_cpu.py pins that behaviour).
"""


class Sink:
    __slots__ = ("grad", "streams")

    def __init__(self, grad):
        self.grad = grad
        self.streams = ()          # streams whose kernels accumulated into `grad` in the current backward (see note)


def note(param, *streams):
    """Backward: remember which streams wrote this parameter's sunk gradient, so that a data-parallel bucket only
    has to wait for the streams of its own parameters (fence_for) -- PoseNet's buckets must not wait for DepthNet's
    backward and vice versa."""
    s = getattr(param, "_dvs_sink", None) if param is not None else None
    if s is not None:
        s.streams = tuple(st for st in streams if st is not None)


def attach(param, grad_view):
    param._dvs_sink = Sink(grad_view)
    return param._dvs_sink


def target(param):
    """The tensor to accumulate into, or None when the parameter has no (live) sink."""
    if param is None:
        return None
    s = getattr(param, "_dvs_sink", None)
    if s is None or param.grad is None or param.grad.data_ptr() != s.grad.data_ptr():
        return None
    return s.grad


# ---------------------------------------------------------------------------------------------
# Side streams for sunk weight gradients.  A weight gradient is needed by nobody until the optimiser step, so
# when it is accumulated in place (no tensor handed back to autograd) its kernel can run on a side stream next
# to the data-gradient chain that the rest of backward is waiting for.  One side stream per compute stream.
#
# The streams, and the events of gradients that the loss chain hands to autograd before they are complete, belong to a
# StreamSet.  A trainer owns one (MonodepthTrainer.streams) and makes it the ACTIVE set of its thread while it builds the
# graph; every autograd Function of this package captures the active set in its forward and uses THAT set in its backward
# (which the engine runs on another thread), so two trainers in one process -- a training and an evaluation model, two
# models in a notebook -- never see each other's streams or pending events.  Code that runs the operators without a
# trainer gets the process-wide default set.  `join()` / `fence()` at module level cover every live set: call them before
# reading gradients (dp.FusedAdam.step and dp.GradSync.finish do).  The first use of a side stream inside a backward pass
# also queues a fence for the END of that pass (autograd's queue_callback: it runs on the thread and stream that called
# backward()), so whatever reads `.grad` next -- a stock torch optimiser, clip_grad_norm_ -- is ordered behind the side
# streams whichever loss produced the gradients (the fused loss chain, a supervised loss on the network's outputs, ...).
# ---------------------------------------------------------------------------------------------
import contextlib
import os
import threading
import weakref

import torch

_enabled = os.environ.get("DVS_WGRAD_STREAM", "1") != "0"
# DVS_WGRAD_STREAM=shared: both networks' weight gradients on ONE side stream (three streams per process instead of four)
_shared = os.environ.get("DVS_WGRAD_STREAM", "1") == "shared"
_PRIORITY_DEFAULT = {"side": "normal", "pose": "normal"}


def _priority(kind):
    """HIP stream priority for the weight-gradient side streams (DVS_SIDE_PRIORITY=low|normal) and the PoseNet stream
    (DVS_POSE_PRIORITY=high|normal): the data-gradient chains are the critical path of the step, the weight gradients only have to be
    done by the optimiser step -- with a lower queue priority their workgroups take the CUs the chains leave."""
    want = os.environ.get("DVS_SIDE_PRIORITY" if kind == "side" else "DVS_POSE_PRIORITY", _PRIORITY_DEFAULT[kind])
    try:
        least, greatest = torch.cuda.Stream.priority_range()
    except Exception:
        return 0
    return {"low": least, "high": greatest}.get(want, 0)


_sets = weakref.WeakSet()           # every live StreamSet (module-level join / fence / reset cover all of them)
_tls = threading.local()

# torch.cuda.current_stream() builds a Stream object through four layers of device-index helpers (~4 us, ~300 calls per training
# step from the backward Functions); the raw handle is a 0.2 us call, and torch's streams live in a pool that is never destroyed,
# so the object for a (device, handle) pair can be kept.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)
_stream_objs = {}


def cur_stream():
    """torch.cuda.current_stream(), cached per (device, raw handle)."""
    if _raw_stream is None or _get_device is None:
        return torch.cuda.current_stream()
    dev = _get_device()
    key = (dev, _raw_stream(dev))
    s = _stream_objs.get(key)
    if s is None:
        s = _stream_objs[key] = torch.cuda.current_stream()
    return s


class StreamSet:
    """Side streams (one per compute stream) and pending-gradient events of one owner."""

    def __init__(self):
        self.side = {}          # (device index, compute stream handle) -> (compute stream, its side stream)
        self.pending = {}       # data_ptr of a gradient tensor -> event recorded on the stream that produces it
        self._end_gid = -1      # graph task for whose end a fence has been queued
        _sets.add(self)

    # ---- side streams
    def side_stream(self, queue=True):
        """Side stream paired with the current stream, or None when disabled (or when the current stream is itself a side
        stream).  Called from backward code: the first call of a backward pass queues the end-of-pass fence; forward code that
        wants a branch to run beside the main chain (the BasicBlock's 1x1 downsample convolution) passes queue=False."""
        if not _enabled:
            return None
        cur = cur_stream()
        if any(cur.cuda_stream == sd.cuda_stream for _, sd in self.side.values()):
            return None                           # already ON a side stream (a forward branch that ran there): no side stream of a side stream
        key = (cur.device_index, cur.cuda_stream)
        pair = self.side.get(key)
        if pair is None:
            shared = None
            if _shared:                          # one side stream per device, whatever the compute stream
                for (dev, _), (_, side) in self.side.items():
                    if dev == cur.device_index:
                        shared = side
            pair = self.side[key] = (cur, shared if shared is not None
                                     else torch.cuda.Stream(device=cur.device, priority=_priority("side")))
        if queue:
            self.queue_end_fence()
        return pair[1]

    def queue_end_fence(self):
        """Inside an engine-driven backward pass: fence() once, when the pass ends, on the caller's thread and stream.
        (Keyed by the engine's graph-task id, so a pass that died with an exception does not mute the next one.)"""
        gid = torch._C._current_graph_task_id()
        if gid == -1 or gid == self._end_gid:     # not inside backward() (a direct call), or already queued for this pass
            return
        try:
            torch.autograd.Variable._execution_engine.queue_callback(self.fence)
            self._end_gid = gid
        except Exception:
            pass

    def join(self):
        """The current stream waits for every side stream of its device."""
        if self.side:
            cur = cur_stream()
            for (dev, _), (_, side) in self.side.items():
                if dev == cur.device_index:
                    cur.wait_stream(side)

    def fence(self):
        """The current stream waits for every stream that can hold gradient-producing work: the side streams AND their
        compute streams (BatchNorm / head gradients are accumulated on the compute stream, DepthNet and PoseNet use
        different ones)."""
        if self.side:
            cur = cur_stream()
            for (dev, handle), (comp, side) in self.side.items():
                if dev != cur.device_index:
                    continue
                if side.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(side)
                if handle != cur.cuda_stream:
                    cur.wait_stream(comp)

    def reset(self):
        self.join()
        self.side.clear()

    # ---- gradients handed to autograd before their producer stream has finished (loss-chain backward by scale)
    def set_pending(self, tensor, event):
        self.pending[tensor.data_ptr()] = event

    def wait_pending(self, tensor):
        """The current stream waits for the producer of `tensor`, if one was registered.  Fail safe: while gradients are
        pending and `tensor` is not one of them (autograd summed or copied it on the way -- a second consumer, a hook),
        wait for ALL pending producers rather than for none."""
        if self.pending and tensor is not None:
            ev = self.pending.pop(tensor.data_ptr(), None)
            cur = cur_stream()
            if ev is not None:
                cur.wait_event(ev)
            else:
                for e in self.pending.values():
                    cur.wait_event(e)
                self.pending.clear()

    def clear_pending(self):
        self.pending.clear()


_default = StreamSet()


def active():
    """The StreamSet a Function should capture in its forward: the one its trainer activated on this thread, else the
    process-wide default."""
    return getattr(_tls, "cur", None) or _default


@contextlib.contextmanager
def use(streams):
    prev = getattr(_tls, "cur", None)
    _tls.cur = streams
    try:
        yield streams
    finally:
        _tls.cur = prev


def of(ctx):
    """The set a Function captured in its forward (`ctx.gs = gradsink.active()`), or the default for a ctx without one."""
    return getattr(ctx, "gs", None) or _default


def enable_side_streams(on):
    """Turn the weight-gradient side streams on / off (off: every kernel runs on its compute stream, which is what
    per-kernel timing wants)."""
    global _enabled
    join()
    _enabled = bool(on)


def side_stream():
    return active().side_stream()


def set_pending(tensor, event):
    active().set_pending(tensor, event)


def wait_pending(tensor):
    active().wait_pending(tensor)


def clear_pending():
    active().clear_pending()


def reset_streams():
    """Drop the side streams of every set (after join()).  A process that builds a second trainer would otherwise keep the
    first one's compute / side streams alive, and HIP multiplexes all live streams onto a handful of hardware queues: the new
    trainer's four streams then share queues and lose part of their overlap (bench.py: batch-4 step 14.1 vs 12.9 ms)."""
    for s in list(_sets):
        s.reset()


def join():
    """The current stream waits for every side stream of its device, whichever set owns it."""
    for s in list(_sets):
        s.join()


def fence_for(params):
    """The current stream waits for the streams noted on `params` (all of them: fence(), when a parameter has no note,
    i.e. its gradient came through autograd on the current stream or from code that does not note)."""
    notes = [getattr(getattr(p, "_dvs_sink", None), "streams", ()) or () for p in params]
    if not any(notes):
        if any(s.side for s in _sets):
            fence()                # GPU run, nothing noted: be safe
        return                     # CPU tensors
    if not all(notes):
        fence()                    # a gradient of this bucket came through autograd (e.g. the stems): wait for everything
        return
    cur = cur_stream()
    seen = {cur.cuda_stream}
    for p in params:
        s = getattr(p, "_dvs_sink", None)
        for st in (s.streams if s is not None else ()):
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                cur.wait_stream(st)


def fence():
    """The current stream waits for every stream of every set that can hold gradient-producing work.  Used before a
    gradient bucket is handed to the all-reduce."""
    for s in list(_sets):
        s.fence()
