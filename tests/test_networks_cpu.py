"""CPU checks of the network drop-ins: state_dict layout (derived from the
reference source), parameter totals, output shapes of the oracle restatement, loud failure off-GPU."""
import pytest
import torch

from deep_visual_slam_amd.depthnet import DepthNet
from deep_visual_slam_amd.posenet_single import PoseNet
from oracle import networks as ON


@pytest.fixture(scope="module")
def nets():
    torch.manual_seed(0)
    return DepthNet(18, pretrained=False), PoseNet(18, pretrained=False, num_input_images=2)


def test_depthnet_state_dict_layout(nets):
    sd = nets[0].state_dict()
    assert sd["encoder.encoder.conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["encoder.encoder.fc.weight"].shape == (1000, 512)
    assert sd["encoder.encoder.layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert "encoder.encoder.layer1.0.downsample.0.weight" not in sd
    assert "encoder.encoder.bn1.num_batches_tracked" in sd
    shapes = [(256, 512, 3, 3), (256, 512, 3, 3), (128, 256, 3, 3), (128, 256, 3, 3), (64, 128, 3, 3),
              (64, 128, 3, 3), (32, 64, 3, 3), (32, 96, 3, 3), (16, 32, 3, 3), (16, 16, 3, 3)]
    for i, sh in enumerate(shapes):
        assert sd["decoder.%d.conv.conv.weight" % i].shape == sh
        assert sd["decoder.%d.conv.conv.bias" % i].shape == (sh[0],)
    for s, cin in enumerate((16, 32, 64, 128)):
        assert sd["decoder.%d.conv.weight" % (10 + s)].shape == (1, cin, 3, 3)
    assert not any(k.startswith("convs") for k in sd)          # self.convs is a plain OrderedDict
    assert sum(p.numel() for p in nets[0].parameters()) == 14842236


def test_posenet_state_dict_layout(nets):
    sd = nets[1].state_dict()
    assert sd["encoder.encoder.conv1.weight"].shape == (64, 6, 7, 7)
    assert sd["net.0.weight"].shape == (256, 512, 1, 1)
    assert sd["net.1.weight"].shape == (256, 256, 3, 3) and sd["net.2.weight"].shape == (256, 256, 3, 3)
    assert sd["net.3.weight"].shape == (6, 256, 1, 1)
    assert sum(p.numel() for p in nets[1].parameters()) == 13011950
    assert type(nets[1]).__module__.endswith("posenet_single")


def test_flowposenet_is_a_name_only():
    from deep_visual_slam_amd.posenet_single import FlowPoseNet
    with pytest.raises(NotImplementedError):
        FlowPoseNet()


def test_invalid_layer_count():
    from deep_visual_slam_amd.resnet_encoder import ResnetEncoder
    with pytest.raises(ValueError):
        ResnetEncoder(19, False)


def test_oracle_network_shapes(nets):
    torch.manual_seed(1)
    x = torch.rand(2, 3, 64, 96)
    out = ON.depthnet(x, nets[0].state_dict(), train=True)
    for s in range(4):
        assert out[("disp", s)].shape == (2, 1, 64 >> s, 96 >> s)
        assert float(out[("disp", s)].min()) > 0 and float(out[("disp", s)].max()) < 1
    aa, t = ON.posenet(torch.rand(2, 6, 64, 96), nets[1].state_dict(), train=True)
    assert aa.shape == (2, 1, 1, 3) and t.shape == (2, 1, 1, 3)


def test_oracle_bn_matches_torch_module():
    """The oracle's hand-written BatchNorm equals nn.BatchNorm2d incl. the running-stat update."""
    torch.manual_seed(2)
    bn = torch.nn.BatchNorm2d(5)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-0.5, 0.5)
    x = torch.randn(3, 5, 7, 9)
    sd = {"p." + k: v.clone() for k, v in bn.state_dict().items()}
    upd = {}
    y = ON._bn(x, sd, "p", True, update=upd)
    y_ref = bn(x)
    assert torch.allclose(y, y_ref, atol=1e-5)
    assert torch.allclose(upd["p.running_mean"], bn.running_mean, atol=1e-6)
    assert torch.allclose(upd["p.running_var"], bn.running_var, atol=1e-6)


def test_forward_on_cpu_fails_loudly(nets):
    from deep_visual_slam_amd._lib import DvsError
    with pytest.raises(DvsError):
        nets[0](torch.rand(1, 3, 32, 32))
