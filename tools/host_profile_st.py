"""cProfile of whole steps with autograd's backward on the CALLING thread (torch.autograd.set_multithreading_enabled(False)), so
that the Python cost of the backward Functions is visible line by line.  Usage: python tools/host_profile_st.py [batch] [sort]"""
import cProfile, pstats, sys, io, time
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sort = sys.argv[2] if len(sys.argv) > 2 else "tottime"
tr, flat, sync, opt, sample = bench.build_gpu(B, 1 if B == 4 else 4, dev, 0)
for mt in (True, False):
    torch.autograd.set_multithreading_enabled(mt)
    for _ in range(5):
        bench.gpu_step(tr, sync, opt, sample)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        bench.gpu_step(tr, sync, opt, sample)
    torch.cuda.synchronize()
    print("multithreading %s: %.2f ms/step" % (mt, (time.perf_counter() - t) / 30 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.gpu_step(tr, sync, opt, sample)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(sort).print_stats(45)
print(s.getvalue()[:9000])
