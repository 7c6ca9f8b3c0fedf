"""ResnetEncoder -- drop-in for the reference's model/resnet_encoder.py:75-111.

The reference subclasses/instantiates torchvision.models.resnet (un-vendored third party); this file
restates the ResNet-18/34 BasicBlock recipe (conv3x3(s)-BN-ReLU-conv3x3-BN + identity or
conv1x1(s)-BN downsample, ReLU; stem conv7x7 s2 p3 no bias, BN, ReLU, maxpool 3 s2 p1) with the same
module tree so that state_dict keys/shapes match SURVEY.md Appendix A (`encoder.conv1.weight`,
`encoder.layer2.0.downsample.0.weight`, the unused `encoder.fc.*`, ...).  nn.Conv2d/nn.BatchNorm2d
are used as parameter containers only; the arithmetic goes through nn_ops (GPU only).
"""
import numpy as np
import torch
import torch.nn as nn

from . import nn_ops


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        return self.forward_with_input(x)[0]

    def forward_with_input(self, x):
        """(block(x), x').  On the training path x' is x after it has travelled through the autograd nodes of its consumers
        inside the block (conv1, then the downsample convolution): one more consumer of the block input -- DepthNet's skip
        connection -- reads x', and every consumer's gradient is added in the next one's data-gradient kernel instead of by
        autograd accumulation passes."""
        # identity / downsample input taken from conv1's autograd node (nn_ops.conv_bn_relu_with_identity)
        out, x = nn_ops.conv_bn_relu_with_identity(x, self.conv1.weight, self.bn1, self.stride, 1)
        if self.downsample is None:
            return nn_ops.conv_bn_act(out, self.conv2.weight, self.bn2, 1, 1, relu=True, residual=x), x
        res = (self.downsample[0].weight, self.downsample[1], self.stride)
        return nn_ops.conv_bn_act(out, self.conv2.weight, self.bn2, 1, 1, relu=True, residual=x, res=res, res_passthrough=True)


class ResNet(nn.Module):
    """Module tree of torchvision.models.ResNet for BasicBlock nets (18/34 layers)."""

    def __init__(self, layers, num_input_images=1, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)   # never used by forward; kept for checkpoint compatibility
        # same initialisation scheme as torchvision / model/resnet_encoder.py:35-40
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)


_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}


class ResnetEncoder(nn.Module):
    """Pytorch module for a resnet encoder (model/resnet_encoder.py:75-111)."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers not in (18, 34, 50, 101, 152):
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if num_layers not in _BLOCKS:
            raise NotImplementedError("only BasicBlock ResNets (18/34) are on the MI355X hot path; "
                                      "the reference trainer uses 18 (vo/train.py:67-93)")
        self.encoder = ResNet(_BLOCKS[num_layers], num_input_images=num_input_images)
        # False: the caller never reads features[0] (PoseNet uses the last feature only), so the training forward does not
        # materialise relu(bn1(conv1(x))) and hands out None in its place
        self.need_feature0 = True
        if pretrained:
            self._load_imagenet(num_layers, num_input_images)

    def _load_imagenet(self, num_layers, num_input_images):
        """The reference downloads torchvision's IMAGENET1K_V1 weights here
        (model/resnet_encoder.py:55-70,95).  Offline, a local copy of that state_dict can be supplied
        through DVS_IMAGENET_RESNET{18,34}; otherwise the seeded default init is kept (the reference
        trainer overwrites these weights from its own checkpoint anyway, vo/train.py:83-98)."""
        import os
        import warnings
        path = os.environ.get("DVS_IMAGENET_RESNET%d" % num_layers)
        if not path:
            warnings.warn("pretrained=True: ImageNet weights are not available offline; keeping the default "
                          "initialisation (set DVS_IMAGENET_RESNET%d to a torchvision state_dict)" % num_layers)
            return
        loaded = torch.load(path, map_location="cpu", weights_only=True)
        if num_input_images > 1:   # model/resnet_encoder.py:65-67
            loaded["conv1.weight"] = torch.cat([loaded["conv1.weight"]] * num_input_images, 1) / num_input_images
        self.encoder.load_state_dict(loaded)

    def forward(self, input_image):
        e = self.encoder
        self.features = []
        # (x - 0.45) / 0.225 is folded into conv1's gather of the planar image (resnet_encoder.py:102-103)
        nch = input_image.shape[1]
        key = (input_image.device, nch)
        if getattr(self, "_norm_key", None) != key:     # constants: built once, not two fill launches per forward
            self._norm_key = key
            self._norm = (torch.full((nch,), 1.0 / 0.225, device=input_image.device),
                          torch.full((nch,), -0.45 / 0.225, device=input_image.device))
        scale, shift = self._norm
        # bn1 + relu + maxpool in one pass on the training path; features[0] is None when nobody reads it (need_feature0)
        z, x = nn_ops.stem_conv_bn_relu_pool(input_image, e.conv1.weight, e.bn1, (scale, shift), need_z=self.need_feature0)
        self.features.append(z)
        # a feature map has up to three consumers (the next layer's conv1 and downsample branch, DepthNet's skip connection):
        # the tensor handed out as the feature is the one that has passed through the encoder-side consumers' autograd nodes
        for layer in (e.layer1, e.layer2, e.layer3, e.layer4):
            for i, block in enumerate(layer):
                if i == 0 and block.downsample is not None:
                    x, self.features[-1] = block.forward_with_input(x)
                else:
                    x = block(x)
            self.features.append(x)
        return self.features
