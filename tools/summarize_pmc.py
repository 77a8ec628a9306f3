"""Condense a rocprofv3 --pmc counter_collection.csv into per-kernel totals.
usage: python tools/summarize_pmc.py <counter_collection.csv> > profiles/<name>.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"], r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum", "Per_Dispatch"])
for (k, c), (n, v) in agg.items():
    w.writerow([k, c, n, v, v / n])
