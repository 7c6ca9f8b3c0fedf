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
# `join()` makes the current stream wait for all of them: call it before reading the gradients (dp.FusedAdam.step
# and dp.GradSync.finish do) and before the scratch pool is recycled (MonodepthTrainer.process_batch does).
# ---------------------------------------------------------------------------------------------
import os

import torch

_side = {}          # (device index, compute stream handle) -> (compute stream, its side stream)
_enabled = os.environ.get("DVS_WGRAD_STREAM", "1") != "0"
# DVS_WGRAD_STREAM=shared: both networks' weight gradients on ONE side stream (three streams per process instead of four)
_shared = os.environ.get("DVS_WGRAD_STREAM", "1") == "shared"


def enable_side_streams(on):
    """Turn the weight-gradient side streams on / off (off: every kernel runs on its compute stream, which is what
    per-kernel timing wants)."""
    global _enabled
    join()
    _enabled = bool(on)


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


_PRIORITY_DEFAULT = {"side": "normal", "pose": "normal"}


def side_stream():
    """Side stream paired with the current stream, or None when disabled."""
    if not _enabled:
        return None
    cur = torch.cuda.current_stream()
    key = (cur.device_index, cur.cuda_stream)
    pair = _side.get(key)
    if pair is None:
        shared = None
        if _shared:                          # one side stream per device, whatever the compute stream
            for (dev, _), (_, side) in _side.items():
                if dev == cur.device_index:
                    shared = side
        pair = _side[key] = (cur, shared if shared is not None else torch.cuda.Stream(device=cur.device, priority=_priority("side")))
    return pair[1]


_pending = {}       # data_ptr of a gradient tensor -> event recorded on the stream that produces it


def set_pending(tensor, event):
    """`tensor` (a gradient handed to autograd) is being written on another stream; its consumer calls wait_pending."""
    _pending[tensor.data_ptr()] = event


def wait_pending(tensor):
    """The current stream waits for the producer of `tensor`, if one was registered (loss-chain backward by scale).
    Fail safe: while gradients are pending and `tensor` is not one of them (autograd summed or copied it on the way --
    a second consumer, a hook), wait for ALL pending producers rather than for none."""
    if _pending and tensor is not None:
        ev = _pending.pop(tensor.data_ptr(), None)
        cur = torch.cuda.current_stream()
        if ev is not None:
            cur.wait_event(ev)
        else:
            for e in _pending.values():
                cur.wait_event(e)
            _pending.clear()


def clear_pending():
    _pending.clear()


def reset_streams():
    """Drop the side streams (after join()).  A process that builds a second trainer would otherwise keep the first one's
    compute / side streams alive here, and HIP multiplexes all live streams onto a handful of hardware queues: the new
    trainer's four streams then share queues and lose part of their overlap (bench.py: batch-4 step 14.1 vs 12.9 ms)."""
    join()
    _side.clear()


def join():
    """The current stream waits for every side stream of its device."""
    if _side:
        cur = torch.cuda.current_stream()
        for (dev, _), (_, side) in _side.items():
            if dev == cur.device_index:
                cur.wait_stream(side)


def fence_for(params):
    """The current stream waits for the streams noted on `params` (all of them: fence(), when a parameter has no note,
    i.e. its gradient came through autograd on the current stream or from code that does not note)."""
    notes = [getattr(getattr(p, "_dvs_sink", None), "streams", ()) or () for p in params]
    if not any(notes):
        if _side:
            fence()                # GPU run, nothing noted: be safe
        return                     # CPU tensors
    if not all(notes):
        fence()                    # a gradient of this bucket came through autograd (e.g. the stems): wait for everything
        return
    cur = torch.cuda.current_stream()
    seen = {cur.cuda_stream}
    for p in params:
        s = getattr(p, "_dvs_sink", None)
        for st in (s.streams if s is not None else ()):
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                cur.wait_stream(st)


def fence():
    """The current stream waits for every stream that can hold gradient-producing work: the side streams AND their
    compute streams (BatchNorm / head gradients are accumulated on the compute stream, DepthNet and PoseNet use
    different ones).  Used before a gradient bucket is handed to the all-reduce."""
    if _side:
        cur = torch.cuda.current_stream()
        for (dev, handle), (comp, side) in _side.items():
            if dev != cur.device_index:
                continue
            if side.cuda_stream != cur.cuda_stream:
                cur.wait_stream(side)
            if handle != cur.cuda_stream:
                cur.wait_stream(comp)
