"""PoseNet -- drop-in for the reference's model/posenet_single.py:149-202.

`FlowPoseNet` exists as a name only because vo/train.py:18 imports it; the trainer never constructs
it (it needs the RAFT checkpoint that is absent from the reference tree)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import nn_ops
from .resnet_encoder import ResnetEncoder


class PoseNet(nn.Module):
    """Pose estimation network with ResNet encoder and pose decoder"""
    supports_pairs = True      # forward(x, pairs=2): two frame pairs in one pass (see forward)

    def __init__(self, num_layers=18, pretrained=True, num_input_images=2, stride=1):
        super().__init__()
        self.num_input_images = num_input_images
        self.stride = stride
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=pretrained,
                                     num_input_images=num_input_images)
        self.encoder.need_feature0 = False      # only feature[-1] is read below
        self.num_ch_enc = self.encoder.num_ch_enc
        self.convs = OrderedDict()
        self.convs[("squeeze")] = nn.Conv2d(self.num_ch_enc[-1], 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))
        # weights live as [Cout][kh][kw][Cin] in memory (same logical shapes / state_dict)
        self.to(memory_format=torch.channels_last)

    def forward(self, input_images, pairs=1):
        """pairs = 2 (an extension of the reference signature): `input_images` holds two batches back to back --
        MonodepthTrainer's (left, target) and (target, right) pairs -- evaluated in one pass; BatchNorm treats each
        half as its own batch, so outputs, gradients and running statistics equal two calls."""
        with nn_ops.batch_groups(pairs):
            feature = self.encoder(input_images)
        sq = self.convs["squeeze"]
        out = nn_ops.conv2d(feature[-1], sq.weight, sq.bias, 1, 0, act="relu")
        for i in range(3):
            c = self.convs[("pose", i)]
            out = nn_ops.conv2d(out, c.weight, c.bias, c.stride[0], c.padding[0], act="relu" if i != 2 else None)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, 1, 1, 6)
        return out[..., :3], out[..., 3:]


class FlowPoseNet(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("FlowPoseNet (RAFT-based alternate, model/posenet_single.py:91-147) is "
                                  "off the hot path: vo/train.py imports the name but constructs PoseNet")
