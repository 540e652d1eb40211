"""Condense a rocprofv3 --kernel-trace --stats output directory into a small CSV for profiles/."""
import csv, glob, sys, re
src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(src + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        name = re.sub(r"\(.*", "", name)[:90]
        rows.append([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
rows.sort(key=lambda r: -float(r[2]))
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
    w.writerows(rows)
print(open(dst).read())
