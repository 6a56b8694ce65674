#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel name and counter."""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = (row.get("Kernel_Name", "")[:90], row.get("Counter_Name", ""))
        acc[k][0] += float(row.get("Counter_Value", 0) or 0)
        acc[k][1] += 1
print("kernel,counter,mean_value,dispatches")
for (k, c), (s, n) in sorted(acc.items()):
    if k.startswith("bj::") or "bj::" in k:
        print('"%s",%s,%.1f,%d' % (k, c, s / n, n))
