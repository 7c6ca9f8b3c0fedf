import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def golden_chain_inputs(rec):
    """Rebuild (sample, disps, poses, noise) torch inputs from a chain_* fixture."""
    t = lambda k: torch.from_numpy(rec[k])
    sample = {("source_left", 0): t("in/source_left"), ("target_image", 0): t("in/target_image"),
              ("source_right", 0): t("in/source_right"), ("K", 0): t("in/K"), ("inv_K", 0): t("in/inv_K")}
    ns = int(rec["meta/num_scales"])
    disps = [t("in/disp%d" % s) for s in range(4)]
    poses = [t("in/" + n) for n in ("aa_left", "t_left", "aa_right", "t_right")]
    noise = [t("in/noise%d" % s) for s in range(ns)]
    return sample, disps, poses, noise, ns


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
