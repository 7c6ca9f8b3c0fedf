"""`Networks` -- the network adapter of the reference's SLAM front end, slam/network.py:19-74, on the MI355X inference path
(SURVEY.md section 8(f) rank 4).

`slam/MonoVO.py:9-12,19-27` constructs `Networks(depth_weight_path, pose_weight_path, image_shape)` and calls
`.depth(frame) -> [H,W] numpy depth clipped to [0.1, 10]` and `.pose(img1, img2, depth) -> 4x4 numpy`; frames are the
BGR uint8 images of the OpenCV capture loop.  The reference file is a stale TensorFlow implementation (it imports `tf`,
`model.depth_net.DispNet` and `vo.utils.d3vo_projection_utils`, none of which exist in the tree); this class keeps its
contract and backs it with the torch-checkpoint DepthNet / PoseNet of this package: BGR -> RGB, / 255 and HWC -> CHW in
one HIP kernel (dvs_u8_to_f32_planar), folded-BatchNorm inference convolutions, PoseNet and DepthNet side by side on two
streams (inference.FramePredictor's structure), `transformation_from_parameters(invert=True)` for the pose matrix
(slam/network.py:71; slam/export_model.py:101-127 spells the same R^T . T(-t))."""
import numpy as np
import torch

from . import _lib, inference
from .depthnet import DepthNet
from .input_pipeline import u8_to_f32_planar
from .layers import disp_to_depth, transformation_from_parameters
from .posenet_single import PoseNet


def _load(net, path):
    if path is None:
        return
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    sd = {(k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k): v for k, v in sd.items()}    # vo/train.py:28-36
    net.load_state_dict(sd)


class Networks:
    def __init__(self, depth_weight_path: str = None, pose_weight_path: str = None, image_shape: tuple = (480, 640),
                 device="cuda:0", graph=True):
        if not torch.cuda.is_available():
            raise _lib.DvsError("Networks: needs the GPU (this package has no CPU path)")
        self.image_shape = tuple(image_shape)
        self.batch_size = 1
        self.device = torch.device(device)
        self.depth_net = DepthNet(num_layers=18, pretrained=False)
        self.pose_net = PoseNet(num_layers=18, pretrained=False, num_input_images=2)
        _load(self.depth_net, depth_weight_path)
        _load(self.pose_net, pose_weight_path)
        self.depth_net.to(self.device)
        self.pose_net.to(self.device)
        inference.prepare(self.depth_net, self.pose_net, scales=(0,))
        for p in list(self.depth_net.parameters()) + list(self.pose_net.parameters()):
            p.requires_grad_(False)                                # `.trainable = False` of slam/network.py:33,40
        H, W = self.image_shape
        self._graph = bool(graph)
        self._depth_fn = self._pose_fn = None
        self._u8 = torch.empty(2, H, W, 3, dtype=torch.uint8).pin_memory()

    def _preprocess(self, *images, is_bgr=True):
        """uint8 HWC frames (numpy) -> fp32 [N,3,H,W] in [0,1], RGB (slam/network.py:42-50)."""
        H, W = self.image_shape
        n = len(images)
        for i, im in enumerate(images):
            im = np.asarray(im)
            if im.shape != (H, W, 3) or im.dtype != np.uint8:
                raise _lib.DvsError("Networks: frames must be uint8 [%d,%d,3] (got %s %s)" % (H, W, im.dtype, im.shape))
            self._u8[i].copy_(torch.from_numpy(np.ascontiguousarray(im)))
        dev = self._u8[:n].to(self.device, non_blocking=True)
        return u8_to_f32_planar(dev, bgr=is_bgr)

    def depth(self, image):
        """[H,W] float32 numpy depth of one BGR frame, clipped to [0.1, 10] (slam/network.py:52-60)."""
        x = self._preprocess(image)
        with torch.no_grad():
            if self._graph:
                if self._depth_fn is None:
                    self._depth_fn = inference.Graphed(self.depth_net, x)
                disp = self._depth_fn(x)[("disp", 0)]
            else:
                disp = self.depth_net(x)[("disp", 0)]
            _, depth = disp_to_depth(disp, 0.1, 10.0)
            depth = depth[0, 0].clamp(0.1, 10.0)
        return depth.cpu().numpy()

    def pose(self, img1, img2, depth=None, translation_scale=5.6):
        """4x4 float32 numpy camera motion from (img1, img2), inverted as the reference does (slam/network.py:62-74;
        `depth` and `translation_scale` are accepted and, as in the reference body, not used by the arithmetic)."""
        x = self._preprocess(img1, img2)
        pair = torch.cat([x[0:1], x[1:2]], 1)
        with torch.no_grad():
            if self._graph:
                if self._pose_fn is None:
                    self._pose_fn = inference.Graphed(self.pose_net, pair)
                aa, t = self._pose_fn(pair)
            else:
                aa, t = self.pose_net(pair)
            T = transformation_from_parameters(aa[:, 0], t[:, 0], invert=True)
        return T[0].cpu().numpy()
