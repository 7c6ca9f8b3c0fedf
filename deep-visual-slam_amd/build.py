"""Build recipe for libdvslam_hip.so (hipcc, gfx950 only, in-tree so the .so travels with gpurun).

    python -m deep_visual_slam_amd.build          # incremental
    python -m deep_visual_slam_amd.build --force
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdvslam_hip.so")
RCCL_LIB = os.path.join(HERE, "libdvslam_rccl.so")          # include/dvslam_rccl.h: kept apart so single-GPU users never load RCCL
RCCL_SRC = os.path.join(CSRC, "rccl", "allreduce.cpp")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -munsafe-fp-atomics: float atomicAdd lowers to one global_atomic_add_f32 / ds_add_f32 (no CAS loop)
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "dvslam.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, dep_mtime, verbose):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= dep_mtime):
        return obj, False
    cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip() and verbose:
        print(r.stderr, file=sys.stderr)
    return obj, True


def build(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link libdvslam_hip.so.  Returns the library path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    dep_mtime = _deps_mtime()
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, dep_mtime, verbose), srcs))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or force or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        _check_loads(LIB)
    build_rccl(force, verbose)
    return LIB


def _check_loads(path):
    """dlopen the fresh library in a child process: an undefined symbol (e.g. a kernel launch stub that hipcc's host pass dropped)
    must fail the BUILD, not the first import on the GPU box."""
    r = subprocess.run([sys.executable, "-c", "import ctypes, sys; ctypes.CDLL(sys.argv[1])", path], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s does not load:\n%s" % (path, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else r.returncode))


def build_rccl(force=False, verbose=True):
    """libdvslam_rccl.so: the RCCL all-reduce entry points of include/dvslam_rccl.h (host code only, links librccl)."""
    hdr = os.path.join(os.path.dirname(HERE), "include", "dvslam_rccl.h")
    if (not force and os.path.exists(RCCL_LIB) and os.path.getmtime(RCCL_LIB) >= os.path.getmtime(RCCL_SRC)
            and os.path.getmtime(RCCL_LIB) >= os.path.getmtime(hdr)):
        return RCCL_LIB
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(rocm, "include"), RCCL_SRC, "-o", RCCL_LIB,
           "-L" + os.path.join(rocm, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (RCCL_SRC, r.stdout, r.stderr))
    return RCCL_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
