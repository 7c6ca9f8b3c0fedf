"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/dvslam.h declares.  No compute is called (there is no GPU here)."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from deep_visual_slam_amd import _lib
    return _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dvslam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dvs_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(built):
    assert header_symbols() == built.exported_symbols()


def test_binding_arity_matches_the_header(built):
    """Every ctypes signature passes exactly as many arguments as the prototype in include/dvslam.h declares (a miscounted
    binding only fails at its first call, on the GPU box)."""
    src = open(os.path.join(ROOT, "include", "dvslam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, (_, args) in built._SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_library_exports_every_symbol(built):
    l = built.lib()
    for name in header_symbols():
        assert hasattr(l, name), name
    assert l.dvs_abi_version() == built.ABI_VERSION
    assert l.dvs_arch() == b"gfx950"


def test_bad_arguments_fail_loudly(built):
    import ctypes as C
    l = built.lib()
    assert l.dvs_pose_to_mat_fwd(None, None, 0, None, 4, None) < 0
    assert b"null" in l.dvs_last_error()
    cfg = built.ChainCfg()
    cfg.B, cfg.H, cfg.W, cfg.num_scales = 1, 48, 64, 9
    assert l.dvs_chain_workspace(C.byref(cfg), None, None, None, None) < 0
    assert b"num_scales" in l.dvs_last_error()


def test_cpu_tensors_are_rejected(built):
    import torch
    from deep_visual_slam_amd import ops
    with pytest.raises(built.DvsError):
        ops.pose_to_mat(torch.zeros(2, 1, 3), torch.zeros(2, 1, 3))


def test_code_object_is_gfx950_only(built):
    """The shipped fat binary carries exactly one device target: gfx950."""
    data = open(built.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data))
    assert targets == {b"gfx950"}, targets


def test_rccl_library_exports_every_symbol_of_its_header(built):
    """include/dvslam_rccl.h <-> deep_visual_slam_amd/_rccl.py <-> libdvslam_rccl.so (no collective is called)."""
    from deep_visual_slam_amd import _rccl
    src = open(os.path.join(ROOT, "include", "dvslam_rccl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(dvs_[a-z0-9_]+)\s*\(", src)))
    assert names == _rccl.exported_symbols()
    l = _rccl.lib()
    for n in names:
        assert hasattr(l, n), n
    assert l.dvs_allreduce_run(None, None, 0, None) < 0 and b"null" in l.dvs_rccl_last_error()
