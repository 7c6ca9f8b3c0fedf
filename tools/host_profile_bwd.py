"""Host time per autograd Function (forward and backward separately; the backward runs on the engine's thread, which cProfile on
the calling thread does not see).  Usage: python tools/host_profile_bwd.py [batch]"""
import sys, time, collections, inspect
import torch
sys.path.insert(0, ".")
import bench
import deep_visual_slam_amd as pkg
import importlib, pkgutil

acc = collections.defaultdict(lambda: [0, 0.0])


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        e = acc[name]
        e[0] += 1
        e[1] += time.perf_counter() - t
        return r
    return staticmethod(w)


seen = set()
for m in pkgutil.iter_modules(pkg.__path__):
    try:
        mod = importlib.import_module(pkg.__name__ + "." + m.name)
    except Exception:
        continue
    for n, c in inspect.getmembers(mod, inspect.isclass):
        if issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function and c not in seen and c.__module__.startswith(pkg.__name__):
            seen.add(c)
            c.forward = timed(c.__name__ + ".fwd", c.__dict__["forward"].__func__ if isinstance(c.__dict__["forward"], staticmethod) else c.forward)
            c.backward = timed(c.__name__ + ".bwd", c.__dict__["backward"].__func__ if isinstance(c.__dict__["backward"], staticmethod) else c.backward)

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tr, flat, sync, opt, sample = bench.build_gpu(B, 1 if B == 4 else 4, dev, 0)
for _ in range(5):
    bench.gpu_step(tr, sync, opt, sample)
torch.cuda.synchronize()
acc.clear()
N = 20
phase = collections.defaultdict(float)
t0 = time.perf_counter()
for _ in range(N):
    t = time.perf_counter()
    _, losses = tr.process_batch(sample)
    t1 = time.perf_counter()
    losses["loss"].backward()
    t2 = time.perf_counter()
    sync.finish()
    opt.step(grad_scale=sync.grad_scale, zero_grad=True)
    t3 = time.perf_counter()
    phase["forward"] += t1 - t
    phase["backward"] += t2 - t1
    phase["finish+adam"] += t3 - t2
torch.cuda.synchronize()
print("step %.2f ms (host loop incl. final sync)" % ((time.perf_counter() - t0) / N * 1e3))
for k, v in phase.items():
    print("  %-12s %.2f ms/step" % (k, v / N * 1e3))
tot = 0.0
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %6.1f calls/step %7.1f us/call %7.3f ms/step" % (k, n / N, s / n * 1e6, s / N * 1e3))
    tot += s
print("sum inside Functions: %.2f ms/step" % (tot / N * 1e3))
