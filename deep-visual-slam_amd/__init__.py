"""MI355X-native Monodepth2 VO training/inference hot path (drop-in for the reference's
model/layers.py, model/depthnet.py, model/posenet_single.py, vo/learner_func.py, vo/learner_new.py).

The compute lives in ``csrc/`` (hand-written HIP for gfx950 behind the C-ABI declared in
``include/dvslam.h``); the Python modules here mirror the reference's operator surface and call the
library through ctypes.  There is no CPU fallback: every operator raises if the HIP library is not
built or its inputs are not on a GPU.
"""
__version__ = "0.1.0"
