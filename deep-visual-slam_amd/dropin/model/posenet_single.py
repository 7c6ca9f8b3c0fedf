"""Reference import path `model.posenet_single` -> MI355X implementation (deep-visual-slam_amd/posenet_single.py)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from deep_visual_slam_amd.posenet_single import *  # noqa: F401,F403,E402
from deep_visual_slam_amd import posenet_single as _impl  # noqa: E402

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
