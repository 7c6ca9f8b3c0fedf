"""DepthLearner -- drop-in for the reference's supervised depth learner, depth/depth_learner.py:6-147
(SURVEY.md section 8(f) rank 4).

Same constructor, attributes and methods (`disp_to_depth`, `compute_gradients`, `get_smooth_loss`, `silog_loss`,
`multi_scale_loss`, `forward_step`); the loop of `multi_scale_loss` -- per scale F.interpolate + mean-normalised
edge-aware smoothness + SILog over the valid pixels, ~40 eager kernels per scale in the reference -- is one fused
forward launch and one fused backward launch of libdvslam_hip.so (ops.depth_multiscale_losses, csrc/depth_loss.hip).
"""
from typing import Any, Dict, List, Tuple

import torch
import torch.nn as nn

from . import ops


class DepthLearner:
    def __init__(self, model: nn.Module, config: Dict[str, Any], device: torch.device) -> None:
        self.model = model
        self.min_depth = config['Train']['min_depth']
        self.max_depth = config['Train']['max_depth']
        self.num_scales = 4
        self.device = torch.device(device if torch.cuda.is_available() else 'cpu')
        self.alphas = [1.0, 0.5, 0.25, 0.125]                     # depth_learner.py:25
        self.smooth_weight = config['Train'].get('smooth_weight', 0.1)
        self.silog_weight = config['Train'].get('silog_weight', 1.0)

    def disp_to_depth(self, disp: torch.Tensor) -> torch.Tensor:
        """depth_learner.py:32-38."""
        min_disp = 1.0 / self.max_depth
        max_disp = 1.0 / self.min_depth
        scaled_disp = min_disp + (max_disp - min_disp) * disp
        depth = 1.0 / scaled_disp
        return depth.float()

    @staticmethod
    def compute_gradients(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """depth_learner.py:40-49 (plain tensor slicing; the fused loss does not go through it)."""
        dx = torch.abs(x[:, :, :, 1:] - x[:, :, :, :-1])
        dy = torch.abs(x[:, :, 1:, :] - x[:, :, :-1, :])
        return dx, dy

    def _one_scale(self, pred, gt, rgb, mask, variance_focus=0.85):
        silog, smooth = ops.depth_multiscale_losses([pred], gt, rgb, mask, variance_focus)
        return silog[0], smooth[0]

    def get_smooth_loss(self, disp: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
        """depth_learner.py:51-73: edge-aware smoothness of disp / mean(disp).clamp(1e-7).  (The silog half of the fused
        kernel runs on an all-valid mask against the map itself and is discarded.)"""
        mask = torch.ones(disp.shape, dtype=torch.uint8, device=disp.device)
        return self._one_scale(disp, disp.detach().abs() + 1.0, img, mask)[1]

    def silog_loss(self, prediction: torch.Tensor, target: torch.Tensor, valid_mask: torch.Tensor,
                   variance_focus: float = 0.85) -> torch.Tensor:
        """depth_learner.py:75-95."""
        B, _, H, W = prediction.shape
        rgb = torch.zeros(B, 3, H, W, device=prediction.device, dtype=torch.float32)
        return self._one_scale(prediction, target, rgb, valid_mask, variance_focus)[0]

    def multi_scale_loss(self, pred_depths: List[torch.Tensor], gt_depth: torch.Tensor, rgb: torch.Tensor,
                         valid_mask: torch.Tensor):
        """depth_learner.py:97-117: (total, total_silog, total_smooth)."""
        n = len(self.alphas)
        silog, smooth = ops.depth_multiscale_losses(list(pred_depths[:n]), gt_depth, rgb, valid_mask)
        alphas = torch.tensor(self.alphas, device=silog.device, dtype=silog.dtype)
        total_silog = (alphas * silog).sum()
        total_smooth = (alphas * smooth).sum()
        total_loss = self.silog_weight * total_silog + self.smooth_weight * total_smooth
        return total_loss, total_silog, total_smooth

    def forward_step(self, sample: Dict[str, torch.Tensor]):
        """depth_learner.py:119-147."""
        rgb = sample['image'].to(self.device)
        depth = sample['depth'].to(self.device)
        valid_mask = sample['valid_mask'].to(self.device)
        if depth.dim() == 3:
            depth = depth.unsqueeze(1)
        outputs = self.model(rgb)
        pred_depths = [self.disp_to_depth(outputs[("disp", scale)]) for scale in range(self.num_scales)]
        total_loss, total_silog, total_smooth = self.multi_scale_loss(pred_depths, depth, rgb, valid_mask)
        return total_loss, total_silog, total_smooth, pred_depths
