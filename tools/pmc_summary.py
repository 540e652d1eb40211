"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter for kernels matching a pattern."""
import csv, collections, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
