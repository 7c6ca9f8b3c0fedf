"""Per-launch view of one training step: every launch the library times (DVS_PROFILE_LOG), in launch order per slot, averaged
over the profiled single-stream steps -- which layer of a slot the time goes to.  tools/per_launch.py [batch] [scales] [steps]"""
import collections
import os
import sys
import tempfile

sys.path.insert(0, ".")
log = os.path.join(tempfile.gettempdir(), "dvs_profile_%d.log" % os.getpid())
os.environ["DVS_PROFILE_LOG"] = log
import torch
import bench
from deep_visual_slam_amd import _lib, dp, gradsink
if os.environ.get("DVS_PRECISION"):
    _lib.set_precision(os.environ["DVS_PRECISION"])      # per-launch view of the bf16 mode: DVS_PRECISION=bf16

dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 12
scales = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
trainer, flat, sync, opt, sample = bench.build_gpu(batch, scales, dev, 0)
torch.cuda.synchronize()
gradsink.enable_side_streams(False)     # one stream: an event pair then measures what a kernel needs, not what it shared
trainer.pose_stream = None
for _ in range(3):
    bench.gpu_step(trainer, sync, opt, sample)
torch.cuda.synchronize()
dp.profile_enable(True)
for _ in range(steps):
    bench.gpu_step(trainer, sync, opt, sample)
torch.cuda.synchronize()
dp.profile_read()
dp.profile_enable(False)

rows = collections.defaultdict(list)
for line in open(log):
    name, work, ms = line.split()
    rows[name].append((float(work), float(ms)))
os.remove(log)
for name, r in rows.items():
    n = len(r) // steps
    if n * steps != len(r):
        print("%s: %d launches do not divide into %d steps" % (name, len(r), steps))
        continue
    tot = sum(ms for _, ms in r) / steps
    print("== %s: %d launches/step, %.3f ms/step" % (name, n, tot))
    for i in range(n):
        work = r[i][0]
        ms = sum(r[i + k * n][1] for k in range(steps)) / steps
        rate = ("%7.1f T/s" % (work / ms / 1e9)) if work > 0 else ""
        print("  %3d  work %8.3f G  %8.1f us  %s" % (i, work / 1e9, ms * 1e3, rate))
