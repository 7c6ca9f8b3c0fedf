"""Reference import path `network` (slam/MonoVO.py:3 `from network import Networks`, run with slam/ on sys.path) ->
the MI355X adapter.  Put deep-visual-slam_amd/dropin/slam ahead of the reference's slam/ on sys.path."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from deep_visual_slam_amd.slam_network import Networks  # noqa: F401,E402
