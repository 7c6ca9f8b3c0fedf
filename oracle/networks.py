"""ORACLE (test infrastructure only) -- CPU restatement of the reference networks.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
Restates, with plain torch-CPU functional ops, the forward of

  * model/resnet_encoder.py:75-111   (ResnetEncoder over torchvision's ResNet-18 BasicBlock recipe)
  * model/depthnet.py:22-90          (DepthNet encoder + decoder)
  * model/posenet_single.py:149-202  (PoseNet)
  * model/layers.py:106-136,196-199  (ConvBlock, Conv3x3, upsample)

The functions take a ``state_dict`` in the reference's key layout, so the same
weights drive the product's GPU modules and this CPU path.  PARITY UNPINNED by reference-run vectors:
the reference's own network files import torchvision, which is absent from the container
(SURVEY.md section 8c); the restatement is pinned by the state_dict key set / shapes, parameter totals
and feature shapes derived from the reference source (tests/test_networks_cpu.py).
"""
import torch
import torch.nn.functional as F


def _bn(x, sd, prefix, train, momentum=0.1, eps=1e-5, update=None):
    """nn.BatchNorm2d: batch statistics in training mode (biased var for normalisation, unbiased for
    the running estimate)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        if update is not None:
            n = x.numel() / x.shape[1]
            update[prefix + ".running_mean"] = (1 - momentum) * rm + momentum * mean.detach()
            update[prefix + ".running_var"] = (1 - momentum) * rv + momentum * var.detach() * n / (n - 1)
            update[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = rm, rv
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def _basic_block(x, sd, p, stride, train, update):
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)
    out = F.relu(_bn(out, sd, p + ".bn1", train, update=update))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1)
    out = _bn(out, sd, p + ".bn2", train, update=update)
    idn = x
    if (p + ".downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
        idn = _bn(idn, sd, p + ".downsample.1", train, update=update)
    return F.relu(out + idn)


def resnet_encoder(x, sd, prefix="encoder.encoder", train=True, update=None, blocks=(2, 2, 2, 2)):
    """model/resnet_encoder.py:100-111 -> list of 5 feature maps."""
    feats = []
    x = (x - 0.45) / 0.225
    x = F.conv2d(x, sd[prefix + ".conv1.weight"], None, 2, 3)
    x = F.relu(_bn(x, sd, prefix + ".bn1", train, update=update))
    feats.append(x)
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nb in enumerate(blocks, start=1):
        for bi in range(nb):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = _basic_block(x, sd, "%s.layer%d.%d" % (prefix, li, bi), stride, train, update)
        feats.append(x)
    return feats


def _conv3x3_refl(x, sd, p):
    """Conv3x3 with ReflectionPad2d(1), model/layers.py:121-136."""
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), sd[p + ".weight"], sd[p + ".bias"])


def depthnet(x, sd, train=True, update=None):
    """model/depthnet.py:64-90 -> {("disp", s): [B,1,H/2^s,W/2^s]}."""
    feats = resnet_encoder(x, sd, "encoder.encoder", train, update)
    outputs = {}
    x = feats[-1]
    idx = 0
    for i in range(4, -1, -1):
        x = F.elu(_conv3x3_refl(x, sd, "decoder.%d.conv.conv" % idx))
        idx += 1
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i > 0:
            x = torch.cat([x, feats[i - 1]], 1)
        x = F.elu(_conv3x3_refl(x, sd, "decoder.%d.conv.conv" % idx))
        idx += 1
        if i in range(4):
            outputs[("disp", i)] = torch.sigmoid(_conv3x3_refl(x, sd, "decoder.%d.conv" % (10 + i)))
    return outputs


def posenet(x, sd, train=True, update=None):
    """model/posenet_single.py:174-202 -> (axisangle [B,1,1,3], translation [B,1,1,3])."""
    f = resnet_encoder(x, sd, "encoder.encoder", train, update)[-1]
    out = F.relu(F.conv2d(f, sd["net.0.weight"], sd["net.0.bias"]))
    out = F.relu(F.conv2d(out, sd["net.1.weight"], sd["net.1.bias"], 1, 1))
    out = F.relu(F.conv2d(out, sd["net.2.weight"], sd["net.2.bias"], 1, 1))
    out = F.conv2d(out, sd["net.3.weight"], sd["net.3.bias"])
    out = out.mean(3).mean(2)
    out = 0.01 * out.view(-1, 1, 1, 6)
    return out[..., :3], out[..., 3:]
