#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel name and counter."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row.get("Kernel_Name", "")[:90], row.get("Counter_Name", ""))
        acc[k][0] += float(row.get("Counter_Value", 0) or 0)
        acc[k][1] += 1
        acc[k][2] = max(acc[k][2], float(row.get("Counter_Value", 0) or 0))
print("kernel,counter,mean_value,dispatches,max_value")
for (k, c), (s, n, mx) in sorted(acc.items()):
    if k.startswith("bj::") or "bj::" in k:
        print('"%s",%s,%.1f,%d,%.1f' % (k, c, s / n, n, mx))
