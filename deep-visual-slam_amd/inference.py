"""Inference path of the networks (callers vo/predict.py:20-42,63-86, vo/eval_traj.py,
vo/eval_redwood.py:325-350 and the ROS2 node): `.eval()` + `torch.no_grad()`, batch 1, `("disp", 0)` and the 6-DoF pose.

Nothing has to change in those callers: in that mode `nn_ops.conv_bn_act` folds every BatchNorm into the convolution in
front of it and lets the conv epilogue add the identity and apply the ReLU (one kernel per conv, `dvs_conv_fusion.residual`).
This module adds the two optional extras a real-time loop wants:

  * `depth_net.inference_scales = (0,)`   skip the three coarse disparity heads nobody reads at inference;
  * `Graphed(net, example)`               capture `net(example)` into a HIP graph and replay it per frame;
  * `FramePredictor(depth, pose, ...)`    the whole per-frame work with PoseNet and DepthNet on two streams (and, by
                                          default, as one graph with a fork / join): at batch 1 neither network fills
                                          the chip, so running them side by side is worth more than the graph.
"""
import torch

from . import _lib, nn_ops


def _fold_snapshot():
    """(generation, references to every folded-BatchNorm tensor currently cached).  A captured graph holds raw pointers to
    those tensors; nn_ops._fold_cache may replace an entry later (new weights, optimiser step) and the old tensors would
    be freed under the graph -- the holder keeps them alive, the generation tells a replay that they are stale."""
    return nn_ops.generation(), [(w, b) for (_, w, b) in nn_ops._fold_cache.values()]


class _WeightsWatch:
    """torch-visible modification state of every parameter / buffer of some networks (in-place torch updates,
    load_state_dict bump `_version`).  One attribute read per tensor and replay: ~25 us for both networks."""

    def __init__(self, *nets):
        self.tensors = [t for net in nets for t in list(net.parameters()) + list(net.buffers())]
        self.ptrs = tuple(t.data_ptr() for t in self.tensors)
        self.seen = self.versions()

    def versions(self):
        return sum(t._version for t in self.tensors)

    def changed(self):
        return self.versions() != self.seen


class Graphed:
    """net(x) for one fixed input shape, replayed from a captured HIP graph.

    The returned tensors are the graph's static outputs: consume (or clone) them before the next call.  The BatchNorm fold is part
    of the capture; the object keeps the folded tensors alive and re-captures by itself when the weights or buffers
    change (torch version counters, nn_ops.generation() for the library's raw-pointer writers)."""

    def __init__(self, net, example, warmup=3):
        if net.training:
            raise _lib.DvsError("Graphed: put the network in eval() mode first (training steps are issued eagerly)")
        if not example.is_cuda:
            raise _lib.DvsError("Graphed: GPU tensors only; this package has no CPU path")
        self.net = net
        self.static_in = example.detach().clone()
        self.warmup = warmup
        self.refresh()

    def refresh(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():        # warm-up off the capture: lazy initialisation, fold cache
            for _ in range(self.warmup):
                self.net(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self.net(self.static_in)
        self._gen, self._held = _fold_snapshot()
        self._watch = _WeightsWatch(self.net)

    def stale(self):
        """True when the weights or BatchNorm buffers changed since the capture (the fold is part of the graph)."""
        return self._gen != nn_ops.generation() or self._watch.changed()

    def __call__(self, x):
        if x.shape != self.static_in.shape:
            raise _lib.DvsError("Graphed: captured for input %s, got %s" % (tuple(self.static_in.shape), tuple(x.shape)))
        if self.stale():
            self.refresh()                   # re-fold and re-capture: never replay pointers to a replaced fold
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out


class FramePredictor:
    """The per-frame work of vo/predict.py:63-86 -- PoseNet on the frame pair + 4x4 pose matrix, DepthNet on the target
    frame + depth -- with the two networks on two HIP streams (they are independent, and at batch 1 neither fills the
    chip), optionally captured as ONE HIP graph with a fork / join.  Returns (T [B,4,4], depth [B,1,H,W], disp); with
    graph=True these are the graph's static outputs (consume or clone them before the next call)."""

    def __init__(self, depth_net, pose_net, target, pair, min_depth=0.1, max_depth=10.0, invert=False, graph=True,
                 warmup=3):
        if depth_net.training or pose_net.training:
            raise _lib.DvsError("FramePredictor: put both networks in eval() mode first")
        from .layers import disp_to_depth, transformation_from_parameters
        self._d2d, self._t = disp_to_depth, transformation_from_parameters
        self.depth_net, self.pose_net = depth_net, pose_net
        self.min_depth, self.max_depth, self.invert = min_depth, max_depth, invert
        self.side = torch.cuda.Stream(device=target.device)
        self.graph = None
        if graph:
            self.s_target, self.s_pair = target.detach().clone(), pair.detach().clone()
            warm = torch.cuda.Stream(device=target.device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm), torch.no_grad():
                for _ in range(warmup):
                    self._run(self.s_target, self.s_pair)
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            self._capture()

    def _capture(self):
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self._run(self.s_target, self.s_pair)
        self._gen, self._held = _fold_snapshot()
        self._watch = _WeightsWatch(self.depth_net, self.pose_net)

    def stale(self):
        return self.graph is not None and (self._gen != nn_ops.generation() or self._watch.changed())

    def _run(self, target, pair):
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            aa, t = self.pose_net(pair)
            T = self._t(aa[:, 0], t[:, 0], invert=self.invert)
        disp = self.depth_net(target)[("disp", 0)]
        _, depth = self._d2d(disp, self.min_depth, self.max_depth)
        main.wait_stream(self.side)
        return T, depth, disp

    def __call__(self, target, pair):
        if self.graph is None:
            with torch.no_grad():
                return self._run(target, pair)
        if self.stale():
            with torch.no_grad():
                self._run(self.s_target, self.s_pair)           # refold eagerly (fills the cache), then re-capture
            torch.cuda.synchronize()
            self._capture()
        self.s_target.copy_(target, non_blocking=True)
        self.s_pair.copy_(pair, non_blocking=True)
        self.graph.replay()
        return self.static_out


def prepare(depth_net, pose_net, scales=(0,)):
    """eval() both networks and restrict DepthNet to the disparity heads an inference caller reads."""
    depth_net.eval()
    pose_net.eval()
    depth_net.inference_scales = tuple(scales) if scales is not None else None
    return depth_net, pose_net
