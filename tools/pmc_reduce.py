"""Mean PMC value per kernel over the dispatches of a rocprofv3 --pmc output tree: python tools/pmc_reduce.py DIR [substr]"""
import csv, glob, os, sys
acc = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            a = acc.setdefault((row["Kernel_Name"], row["Counter_Name"]), [0.0, 0])
            a[0] += float(row["Counter_Value"]); a[1] += 1
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for (k, c), (s, n) in sorted(acc.items()):
    if sub in k:
        print("%-60s %-24s n=%-4d mean=%.1f" % (k[:60], c, n, s / n))
