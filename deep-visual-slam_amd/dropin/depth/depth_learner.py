"""Reference import path `depth.depth_learner` (depth/train.py:14) -> MI355X DepthLearner."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from deep_visual_slam_amd.depth_learner import DepthLearner  # noqa: F401,E402
