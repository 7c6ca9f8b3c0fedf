import csv, glob, collections, sys
root = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_'
vals = {}
for f in sorted(glob.glob(root + '/*/*/*_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items(): vals[c] = sum(v) / len(v)
d = []
for f in glob.glob(root + '/sq1/*/*_kernel_trace.csv'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
print("%s: dur us %.1f" % (root, sum(d) / len(d)))
wc = vals['SQ_WAVE_CYCLES']; cyc = vals['GRBM_GUI_ACTIVE'] / 8; nw = vals['SQ_WAVES']
print("  waves %d  kernel cycles %.3g (%.2f GHz)  MFMA busy / SIMD-cycle %.2f  avg resident waves/SIMD %.2f" % (
    nw, cyc, cyc / (sum(d) / len(d)) / 1e3, vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), wc * 4 / (cyc * 1024)))
for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS'):
    print("  %-22s %.1f%% of wave cycles" % (k, 100 * vals[k] / wc))
print("  per wave: VALU %d SALU %d LDS %d VMEM_RD %d MFMA %d | LDS bank-conflict cycles %d | FETCH_SIZE %.1f MB (x2 corr) | L2 hit %.2f" % (
    vals['SQ_INSTS_VALU'] / nw, vals['SQ_INSTS_SALU'] / nw, vals['SQ_INSTS_LDS'] / nw, vals['SQ_INSTS_VMEM_RD'] / nw,
    vals['SQ_INSTS_MFMA'] / nw, vals['SQ_LDS_BANK_CONFLICT'], vals['FETCH_SIZE'] * 2 / 1024,
    vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum'])))
