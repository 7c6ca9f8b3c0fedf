"""Per-kernel GPU time per training step from a rocprofv3 --kernel-trace CSV of bench.py (last N steps)."""
import csv, glob, collections, sys
root = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(root + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
sel = rows[adam[-nsteps - 1] + 1: adam[-1] + 1]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    key = n.split('(')[0][:64]
    if 'conv_fwd_kernel' in n: key = 'conv_fwd_kernel<' + ('dgrad' if ', 3, ' in n else 'fwd') + '>'
    if 'conv_wgrad_kernel' in n: key = 'conv_wgrad_kernel'
    acc[key][0] += 1; acc[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
tot = sum(v[1] for v in acc.values())
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6
print("GPU busy ms/step: %.2f   wall ms/step: %.2f" % (tot / nsteps, span / nsteps))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-66s n/step %6.1f  %7.3f ms/step" % (k, v[0] / nsteps, v[1] / nsteps))
