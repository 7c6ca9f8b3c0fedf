"""MonodepthTrainer -- drop-in for the reference's vo/learner_new.py:15-258.

Same constructor, attributes and `process_batch(sample) -> (outputs, losses)` contract; the 4x2
scale/frame loop of `_generate_images_pred` and the loss loop of `_compute_losses` (about 600 eager
kernels per step in the reference) are one fused forward launch and one fused backward launch of
libdvslam_hip.so (ops.loss_chain).  The per-scale view-synthesis tensors of the reference's `outputs`
dict (("disp_up",s), ("depth",s), ("sample",f,s), ("color",f,s), "identity_selection/s") are only
consumed by the plotting code every `train_plot_interval` steps (vo/train.py:268), so they are
materialised lazily on first access (or eagerly with config["Train"]["materialize_outputs"]).
"""
from typing import Dict

import os

import torch
import torch.nn as nn

from . import _lib, gradsink, ops, zeropool
from .layers import SSIM, BackprojectDepth, Project3D


class LazyOutputs(dict):
    """`outputs` dict whose view-synthesis tensors are produced on first need.  Every read access a caller of the
    reference's dict could make -- `[]`, `in`, `.get()`, iteration, `.keys()/.items()/.values()`, `len()`, `==`, copying
    -- sees the full schema of vo/learner_new.py:132-172,241: a key that is already present is answered without
    materialising, anything else fills the lazy entries first."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._thunk = None

    def materialize(self):
        if self._thunk is not None:
            thunk, self._thunk = self._thunk, None
            thunk(self)
        return self

    def __missing__(self, key):
        if self._thunk is not None:
            self.materialize()
            if dict.__contains__(self, key):
                return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (self._thunk is not None and dict.__contains__(self.materialize(), key))

    def get(self, key, default=None):
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        return dict.get(self.materialize(), key, default)

    def __iter__(self):
        return dict.__iter__(self.materialize())

    def __len__(self):
        return dict.__len__(self.materialize())

    def keys(self):
        return dict.keys(self.materialize())

    def items(self):
        return dict.items(self.materialize())

    def values(self):
        return dict.values(self.materialize())

    def copy(self):
        return dict(self.materialize())

    def __eq__(self, other):
        return dict.__eq__(self.materialize(), other)

    __hash__ = None

    def __reduce__(self):
        return (dict, (dict(self.materialize()),))


class MonodepthTrainer:
    def __init__(self, depth_net: nn.Module, pose_net: nn.Module, config: dict, device: torch.device):
        self.depth_net = depth_net
        self.pose_net = pose_net
        self.config = config
        self.device = device

        tr = config["Train"]
        self.num_scales = 4
        self.num_source = tr["num_source"]
        self.batch_size = tr["batch_size"]
        self.image_shape = (tr["img_h"], tr["img_w"])
        self.smoothness_ratio = tr["smoothness_ratio"]
        self.auto_mask = tr["auto_mask"]
        self.ssim_ratio = tr["ssim_ratio"]
        self.min_depth = tr["min_depth"]
        self.max_depth = tr["max_depth"]
        self.use_compile = tr["use_compile"]   # accepted for compatibility; there is no tracing compiler here
        self.materialize_outputs = bool(tr.get("materialize_outputs", False))
        # Opt-in (Train.amp_bf16 / DVS_AMP_BF16=1; off: this package computes in fp32 whatever autocast says, which is what
        # every parity statement is made for): when the caller runs process_batch under torch.autocast -- the `use_amp` branch
        # of vo/train.py:177-185 -- the implicit-GEMM convolutions of the step (and of its backward) multiply on the bf16 matrix
        # cores (_lib.set_precision; fp32 tensors, fp32 accumulation, fp32 BatchNorm / loss chain / optimiser).
        self.amp_bf16 = bool(tr.get("amp_bf16", os.environ.get("DVS_AMP_BF16", "0") == "1"))
        self.noise_seed = int(tr.get("noise_seed", 0))
        self._step = 0
        # DepthNet and PoseNet are independent until the loss chain joins them: PoseNet's forward -- and, because
        # autograd replays a node on the stream of its forward, its backward too -- runs on a second HIP stream, so
        # the two networks' kernels fill each other's tails (a 450-workgroup conv on 256 CUs leaves 12 % of the chip
        # idle in its last round) and the many small launches of one hide behind the other's convolutions.
        # The two PoseNet passes stay in order on that one stream (they update the same BatchNorm running statistics).
        self.pose_pairs_batched = bool(tr.get("pose_pairs_batched", os.environ.get("DVS_POSE_BATCHED", "1") != "0"))
        use_stream = tr.get("pose_stream", os.environ.get("DVS_POSE_STREAM", "1") != "0")
        self.pose_stream = (torch.cuda.Stream(device=self.device, priority=gradsink._priority("pose"))
                            if use_stream and torch.device(self.device).type == "cuda" else None)
        # side streams of the weight gradients and the events of the loss chain's late gradients: owned by this trainer, active
        # while it builds a step's graph (gradsink.StreamSet); the loss chain's backward by scale runs its coarse scales on
        # the PoseNet stream (passed per call: ops.loss_chain(aux_stream=))
        self.streams = gradsink.StreamSet()
        self._noise = None   # test hook: inject the reference's torch.randn tie-break noise [S,B,2,H,W]
        # The unchanged caller (vo/train.py:114-117,173-199) builds a stock torch.optim.Adam over plain parameters and
        # calls zero_grad(set_to_none=True) every step.  Unless told otherwise the trainer moves the two networks'
        # parameters into one flat arena itself (dp.FlatParams: same Parameter objects, same values, `.data` / `.grad`
        # become views), so that the weight-gradient, head and BatchNorm kernels can accumulate straight into the gradient
        # arena on their side streams also for that caller; `process_batch` re-attaches and zeroes the gradient views after
        # a set_to_none, and the loss chain's backward queues a join of the side streams at the end of the backward pass,
        # so whoever reads `.grad` next (torch.optim, clip_grad_norm_) sees complete gradients.
        self.arena = self._arena_packs = None
        want_arena = tr.get("arena", os.environ.get("DVS_ARENA", "1") != "0")
        if want_arena and torch.device(self.device).type == "cuda" and depth_net is not None and pose_net is not None:
            from . import conv as _conv, dp
            named = dp.trainable_parameters(depth_net, pose_net)
            if named and all(p.is_cuda for _, p in named) and not any(getattr(p, "_dvs_arena", False) for _, p in named):
                self.arena = dp.FlatParams(named)
                self._arena_packs = _conv.PackedWeights(self.arena.tensors)
                self._arena_versions = None

        # standalone operators kept as public attributes like the reference (learner_new.py:44-57)
        self.ssim = SSIM().to(self.device)
        self.backproject_depth = BackprojectDepth(self.batch_size, self.image_shape[0], self.image_shape[1]).to(self.device)
        self.project_3d = Project3D(self.batch_size, self.image_shape[0], self.image_shape[1]).to(self.device)

    def _compute_reprojection_loss(self, pred: torch.Tensor, target: torch.Tensor):
        """learner_new.py:60-74 (standalone form; the training step uses the fused chain)."""
        l1_loss = torch.abs(target - pred).mean(1, True)
        ssim_loss = self.ssim(pred, target).mean(1, True)
        return self.ssim_ratio * ssim_loss + (1 - self.ssim_ratio) * l1_loss

    def process_batch(self, sample: Dict[str, torch.Tensor]):
        for key in sample:
            if isinstance(sample[key], torch.Tensor):
                sample[key] = sample[key].to(self.device, non_blocking=True)
        gradsink.join()                                          # side-stream kernels of the previous step (every set's)
        if self.amp_bf16:
            # the mode of this step AND of its backward (the caller runs that outside, after leaving autocast); a trainer without
            # the opt-in never touches the process-wide mode
            want = "bf16" if (self.amp_bf16 and torch.is_autocast_enabled("cuda")) else "fp32"
            if want != _lib.precision():
                _lib.set_precision(want)
        zeropool.reset(sample[("target_image", 0)].device)      # one memset for the step's zero-filled scratch
        if self.arena is not None and torch.is_grad_enabled():
            self._arena_prepare()
        with gradsink.use(self.streams):                          # the Functions of this step capture the trainer's stream set
            return self._process_batch(sample)

    def _process_batch(self, sample):
        if self.pose_stream is None:
            outputs = LazyOutputs(self.depth_net(sample[("target_image", 0)]))
            outputs.update(self._predict_poses(sample))
        else:
            main = torch.cuda.current_stream(self.device)
            self.pose_stream.wait_stream(main)                   # inputs, zeroed scratch, last step's weights
            with torch.cuda.stream(self.pose_stream):
                poses = self._predict_poses(sample)
            outputs = LazyOutputs(self.depth_net(sample[("target_image", 0)]))
            if getattr(self, "_timeline", False):                # tools/step_timeline.py
                self._marks = {"pose_done": torch.cuda.Event(enable_timing=True), "depth_done": torch.cuda.Event(enable_timing=True)}
                self._marks["pose_done"].record(self.pose_stream)
                self._marks["depth_done"].record(main)
            main.wait_stream(self.pose_stream)
            for t in poses.values():
                t.record_stream(main)                            # allocated on the pose stream, read by the chain
            outputs.update(poses)
        losses = self._fused_losses(sample, outputs)
        return outputs, losses

    def __del__(self):
        try:
            if self._arena_packs is not None:
                self._arena_packs.release()
        except Exception:
            pass

    def _arena_prepare(self):
        """Start of a training step on the trainer-owned arena: gradients that the caller set to None are re-attached to
        the (re-zeroed) gradient arena so the kernels can sink into them; the data-gradient weight packs are refreshed with
        one launch when an optimiser has changed the weights since the last step."""
        a = self.arena
        if any(p.grad is None for p in a.tensors):
            if all(p.grad is None for p in a.tensors):
                a.grads.zero_()
                a.reattach()
            # a mixture (the caller dropped some gradients and kept others) is left alone: plain autograd accumulation
        vers = sum(p._version for p in self._arena_packs.weights)
        if vers != self._arena_versions:
            self._arena_packs.repack()
            self._arena_versions = vers

    def _predict_poses(self, sample):
        """learner_new.py:107-129."""
        outputs = {}
        left, tgt, right = sample[("source_left", 0)], sample[("target_image", 0)], sample[("source_right", 0)]
        if self.pose_pairs_batched and getattr(self.pose_net, "supports_pairs", False):
            # both frame pairs in one PoseNet pass of batch 2B (per-pair BatchNorm statistics inside): half the launches,
            # twice the rows per convolution
            B = tgt.shape[0]
            pair = torch.empty((2 * B, left.shape[1] + tgt.shape[1]) + tuple(tgt.shape[2:]), device=tgt.device, dtype=tgt.dtype)
            torch.cat([left, tgt], dim=1, out=pair[:B])          # written in place: two copies instead of three, half the bytes
            torch.cat([tgt, right], dim=1, out=pair[B:])
            axisangle, translation = self.pose_net(pair, pairs=2)
            axisangle_left, axisangle_right = axisangle[:B], axisangle[B:]
            translation_left, translation_right = translation[:B], translation[B:]
        else:
            axisangle_left, translation_left = self.pose_net(torch.cat([left, tgt], dim=1))
            axisangle_right, translation_right = self.pose_net(torch.cat([tgt, right], dim=1))
        outputs[("axisangle", 0, -1)] = axisangle_left
        outputs[("translation", 0, -1)] = translation_left
        outputs[("axisangle", 0, 1)] = axisangle_right
        outputs[("translation", 0, 1)] = translation_right
        outputs[("cam_T_cam", 0, -1)] = ops.pose_to_mat(axisangle_left[:, 0], translation_left[:, 0], invert=True)
        outputs[("cam_T_cam", 0, 1)] = ops.pose_to_mat(axisangle_right[:, 0], translation_right[:, 0], invert=False)
        return outputs

    def _chain_args(self, sample, outputs):
        disps = [outputs[("disp", s)] for s in range(self.num_scales)]
        return (sample[("target_image", 0)], sample[("source_left", 0)], sample[("source_right", 0)],
                sample[("K", 0)], sample[("inv_K", 0)], outputs[("cam_T_cam", 0, -1)],
                outputs[("cam_T_cam", 0, 1)], disps)

    def _chain_kwargs(self, seed):
        return dict(noise=self._noise, seed=seed, auto_mask=self.auto_mask, min_depth=self.min_depth,
                    max_depth=self.max_depth, ssim_ratio=self.ssim_ratio, smoothness_ratio=self.smoothness_ratio,
                    aux_stream=self.pose_stream)          # follows the attribute (bench.py switches it off to time kernels)

    def _fused_losses(self, sample, outputs):
        """_generate_images_pred + _compute_losses (learner_new.py:132-258) as one fused launch."""
        seed = self.noise_seed + self._step
        self._step += 1
        args = self._chain_args(sample, outputs)
        loss_vec, sel, extras = ops.loss_chain(*args, materialize=self.materialize_outputs, **self._chain_kwargs(seed))
        losses = {"loss/{}".format(s): loss_vec[s] for s in range(self.num_scales)}
        losses["loss"] = loss_vec.sum() / self.num_scales
        srcs = {-1: sample[("source_left", 0)], 1: sample[("source_right", 0)]}

        def fill(out, sel=sel, extras=extras):
            if not extras:
                with torch.no_grad():
                    dargs = [a.detach() if isinstance(a, torch.Tensor) else [d.detach() for d in a] for a in args]
                    _, sel, extras = ops.loss_chain(*dargs, materialize=True, **self._chain_kwargs(seed))
            for s in range(self.num_scales):
                e = extras[s]
                out[("disp_up", s)], out[("depth", s)] = e["disp_up"], e["depth"]
                for i, f in enumerate((-1, 1)):
                    out[("sample", f, s)] = e["grid"][i]
                    out[("color", f, s)] = e["color"][i]
                    out[("color_identity", f, s)] = srcs[f]
                if self.auto_mask:
                    out["identity_selection/{}".format(s)] = (((sel >> (2 * s)) & 3) > 1).float().unsqueeze(1)

        if self.materialize_outputs:
            fill(outputs)
        else:
            outputs._thunk = fill
        return losses
