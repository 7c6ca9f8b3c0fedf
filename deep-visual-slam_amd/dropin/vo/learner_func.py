"""Reference import path `vo.learner_func` / bare `learner_func` (vo/learner_new.py:6-13, vo/eval_traj.py,
vo/eval_redwood.py:19) -> MI355X operator surface (deep-visual-slam_amd/layers.py holds the same names as
the reference's vo/learner_func.py:16-207)."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from deep_visual_slam_amd.layers import (BackprojectDepth, Project3D, SSIM, disp_to_depth, get_smooth_loss,  # noqa: F401,E402
                                         get_translation_matrix, rot_from_axisangle, transformation_from_parameters)
