"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]
        if pat and pat not in r["Kernel_Name"]:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print("%-12s %-28s %-26s n=%3d mean=%.4g" % (f.split("/")[-3], k, c, len(v), sum(v) / len(v)))
