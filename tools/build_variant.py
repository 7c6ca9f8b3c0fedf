"""A/B builds of libdvslam_hip.so: tools/build_variant.py <name> [--flag=-fno-slp-vectorize ...] [--src file.hip=/path/to/other.hip ...]
writes deep-visual-slam_amd/csrc/build/variant_<name>.so (git-ignored, travels with gpurun); run with DVS_LIB=<that path>."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_visual_slam_amd import build as B

name = sys.argv[1]
flags = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--flag=")]
only = [a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("--only=")]          # flags apply to these files only
over = dict(a.split("=", 1)[1].split("=", 1) for a in sys.argv[2:] if a.startswith("--src="))
B.build(verbose=False)
objs = []
for src in B.sources():
    base = os.path.basename(src)
    obj = os.path.join(B.OBJ_DIR, base[:-4] + ".o")
    special = base in over or (flags and (not only or base in only))
    if special:
        obj = os.path.join(B.OBJ_DIR, "%s__%s.o" % (base[:-4], name))
        cmd = [B.HIPCC] + B.CFLAGS + (flags if (not only or base in only) else []) + ["-I" + B.CSRC, "-I" + os.path.join(os.path.dirname(B.HERE), "include"),
                                                                                      "-c", over.get(base, src), "-o", obj]
        subprocess.run(cmd, check=True)
    objs.append(obj)
out = os.path.join(B.OBJ_DIR, "variant_%s.so" % name)
subprocess.run([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
