"""Reference import path `vo.learner_new` / bare `learner_new` (vo/train.py:131-136) -> fused MI355X learner."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from deep_visual_slam_amd.learner_new import LazyOutputs, MonodepthTrainer  # noqa: F401,E402
