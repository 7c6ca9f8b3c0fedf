"""DepthNet -- drop-in for the reference's model/depthnet.py:22-90 (encoder + Monodepth2 decoder)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .layers import Conv3x3, ConvBlock
from .resnet_encoder import ResnetEncoder


class DepthNet(nn.Module):
    def __init__(self, num_layers=18, pretrained=True, num_input_images=1, scales=range(4),
                 num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=pretrained,
                                     num_input_images=num_input_images)
        self.num_ch_enc = self.encoder.num_ch_enc
        self.use_encoder = True
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])

        # decoder: same insertion order as model/depthnet.py:43-60 -> decoder.0 .. decoder.13
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
            num_ch_in = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                num_ch_in += self.num_ch_enc[i - 1]
            self.convs[("upconv", i, 1)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()
        # inference only (eval() + no_grad): compute just these disparity heads, e.g. (0,) for vo/predict.py:79-80 which
        # reads ("disp", 0) alone; None = every scale, as the reference does
        self.inference_scales = None
        # weights live as [Cout][kh][kw][Cin] in memory (same logical shapes / state_dict)
        self.to(memory_format=torch.channels_last)

    def _wanted(self, scale):
        if self.training or torch.is_grad_enabled() or self.inference_scales is None:
            return True
        return scale in self.inference_scales

    def forward(self, input_data) -> dict:
        input_features = self.encoder(input_data)
        self.outputs = {}
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            # upsample(x) ; cat skip ; ConvBlock -- one fused gather + conv (model/depthnet.py:80-85)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            x = self.convs[("upconv", i, 1)](x, skip=skip, upsample=True)
            if i in self.scales and self._wanted(i):
                if i > 0:
                    # x feeds this head and the next decoder level: it travels through the head's autograd node, whose
                    # data-gradient kernel then adds the next level's gradient (no autograd accumulation pass)
                    self.outputs[("disp", i)], x = self.convs[("dispconv", i)](x, act="sigmoid", passthrough=True)
                else:
                    self.outputs[("disp", i)] = self.convs[("dispconv", i)](x, act="sigmoid")
        return self.outputs
