"""Sums rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "k_"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row["Kernel_Name"].split("(")[0]
        if sub not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k].add(row["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(calls[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.4g per dispatch" % (c, v / len(calls[k])))
