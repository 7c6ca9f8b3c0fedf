"""Condense tools/profile_bench.sh output: per kernel family, launches/step, avg duration, ms/step and the
PMC HBM traffic per launch (FETCH_SIZE doubled: on gfx950 it tallies 128-B requests at 64 B, see
MI355X_MICROARCH.md; WRITE_SIZE as reported).  Only the last N steps (delimited by adam_kernel) are used.
Writes <out>/traffic.json for bench.py's roofline.traffic."""
import collections
import csv
import glob
import json
import sys

root, nsteps = sys.argv[1], int(sys.argv[2])


_phase = ["fwd"]


def family(n):
    """Kernel name -> the profile slot bench.py counts it in.  Stateful: rows must be fed in launch order -- the Winograd kernel
    serves the forward convolution and (on dY, with the rotated filter) the data gradient, told apart by whether the loss chain's
    backward kernel of the step has run."""
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    base = n.split("(")[0]
    if "chain_bwd_kernel" in n:
        _phase[0] = "bwd"
    elif "adam_kernel" in n:
        _phase[0] = "fwd"
    if "wino_wgrad" in n:
        return "conv_wgrad_kernel"
    if "wino_fwd_kernel" in n:
        return "conv_dgrad_kernel" if _phase[0] == "bwd" else "conv_fwd_kernel"
    if "conv_fwd_kernel" in n or "conv_dma_kernel" in n:
        targs = base.split("<")[1].split(",") if "<" in base else []
        mode = targs[4].strip() if len(targs) > 4 else "?"
        return "conv_dgrad_kernel" if mode.startswith("3") or "IN_DGRAD" in mode else "conv_fwd_kernel"
    if "conv_wgrad" in n or "stem_wgrad" in n or "thin_wgrad" in n:
        return "conv_wgrad_kernel"
    if "thin_fwd" in n:
        return "conv_fwd_kernel"
    if "thin_dgrad" in n or "reflect_fold" in n or "act_bwd" in n:
        return "conv_dgrad_kernel"
    if "bn_bwd" in n or "bn_pool_bwd" in n:
        return "bn_bwd_kernel"
    if "bn_apply_fwd" in n or "bn_finalize" in n or "bn_fwd_fused" in n or "bn_relu_maxpool_fwd" in n:
        return "bn_fwd_kernel"
    return base.split("<")[0][-48:]


def last_steps(rows, key):
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[key]]
    return rows[adam[-nsteps - 1] + 1: adam[-1] + 1]


trace = glob.glob(root + "/stats/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
sel = last_steps(rows, "Kernel_Name")
dur = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    f = family(r["Kernel_Name"])
    dur[f][0] += 1
    dur[f][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
busy = sum(v[1] for v in dur.values()) / nsteps
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / nsteps


def pmc(which, counter):
    fs = glob.glob(root + "/%s/*/*_counter_collection.csv" % which)
    if not fs:
        return {}
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    sel = last_steps(rows, "Kernel_Name")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        f = family(r["Kernel_Name"])
        acc[f][0] += 1
        acc[f][1] += float(r["Counter_Value"])
    return acc


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
print("GPU busy %.2f ms/step, first-to-last kernel span %.2f ms/step (last %d steps)" % (busy, span, nsteps))
print("%-40s %8s %10s %9s %12s %12s" % ("kernel family", "n/step", "avg us", "ms/step", "rd MB/launch", "wr MB/launch"))
out = {}
for f, (n, ms) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    # FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3
    rd = 2.0 * fetch[f][1] * 1024 / fetch[f][0] if f in fetch and fetch[f][0] else None
    wr = write[f][1] * 1024 / write[f][0] if f in write and write[f][0] else None
    out[f] = {"launches_per_step": n / nsteps, "avg_us": 1e3 * ms / n, "ms_per_step": ms / nsteps,
              "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": (rd or 0) + (wr or 0) if rd is not None or wr is not None else None}
    print("%-40s %8.1f %10.1f %9.3f %12s %12s" % (f, n / nsteps, 1e3 * ms / n, ms / nsteps,
                                                   "%.2f" % (rd / 1e6) if rd is not None else "-",
                                                   "%.2f" % (wr / 1e6) if wr is not None else "-"))
json.dump({"busy_ms_per_step": busy, "span_ms_per_step": span, "kernels": out},
          open(root + "/traffic.json", "w"), indent=1)
