"""The reference's import statements resolve to the MI355X modules when dropin/ is put on sys.path
(vo/train.py:13-19, vo/learner_new.py:6-13, vo/predict.py:7-16, vo/eval_redwood.py:16-19)."""
import os
import subprocess
import sys

from conftest import ROOT

DROPIN = os.path.join(ROOT, "deep-visual-slam_amd", "dropin")

SCRIPT = r"""
import sys
from model.depthnet import DepthNet
from model.posenet_single import PoseNet, FlowPoseNet
from model.resnet_encoder import ResnetEncoder
from model.layers import disp_to_depth, transformation_from_parameters, BackprojectDepth, Project3D, SSIM, get_smooth_loss, ConvBlock, Conv3x3, upsample
from vo.learner_new import MonodepthTrainer
from vo.learner_func import transformation_from_parameters as t2
from learner_func import disp_to_depth as d2, BackprojectDepth as B2, Project3D as P2, get_smooth_loss as g2, SSIM as S2
import learner_new
import torch
assert DepthNet.__module__.startswith("deep_visual_slam_amd"), DepthNet.__module__
assert MonodepthTrainer.__module__.startswith("deep_visual_slam_amd")
net = DepthNet(num_layers=18, pretrained=False)
pose = PoseNet(num_layers=18, pretrained=False, num_input_images=2)
assert len(net.decoder) == 14 and list(net.num_ch_dec) == [16, 32, 64, 128, 256]
cfg = {"Train": dict(num_source=1, batch_size=2, img_h=64, img_w=96, smoothness_ratio=0.001, auto_mask=True,
                     ssim_ratio=0.85, min_depth=0.1, max_depth=10.0, use_compile=False)}
tr = MonodepthTrainer(net, pose, cfg, torch.device("cpu"))
assert tr.num_scales == 4 and hasattr(tr, "ssim") and hasattr(tr, "backproject_depth") and hasattr(tr, "project_3d")
print("DROPIN-OK")
"""


def test_reference_imports_resolve_to_this_package():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([DROPIN, os.path.join(DROPIN, "vo")])
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)
    assert "DROPIN-OK" in r.stdout, r.stdout + r.stderr
