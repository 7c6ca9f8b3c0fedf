"""Concurrency summary of a rocprofv3 --kernel-trace of bench.py: per step, wall span, union of kernel-busy time,
sum of kernel durations (= serial time), and the same per stream/queue."""
import collections
import csv
import glob
import sys

root, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(root + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
sel = rows[adam[-nsteps - 1] + 1: adam[-1] + 1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = max(e for _, e in iv) - iv[0][0]
serial = sum(e - s for s, e in iv)
print("per step: span %.2f ms, GPU busy (union) %.2f ms, sum of kernel durations %.2f ms, idle %.2f ms" %
      (span / nsteps / 1e6, union / nsteps / 1e6, serial / nsteps / 1e6, (span - union) / nsteps / 1e6))
per = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = (r.get("Queue_Id"), r.get("Stream_Id"))
    per[k][0] += 1
    per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("queue/stream %s: %6.1f launches/step %7.2f ms/step" % (k, v[0] / nsteps, v[1] / nsteps / 1e6))
