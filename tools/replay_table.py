#!/usr/bin/env python3
"""Per-proof kernel table of ONE replayed rank out of a rocprofv3 kernel trace of tools/replay_rank.py (the trace also holds the
recording pass, in which the W ranks share the GPU and every duration is inflated): the launches between two consecutive
`deep_accumulate_multi_kernel` launches are one proof's worth of kernels (the same multiset for every proof, shifted by a constant);
the last `steps - 1` such periods are the replayed rank's timed proofs.  Prints kernel, launches per proof, ms per proof, share.
    python tools/replay_table.py <kernel_trace.csv> <steps>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
name = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "deep_accumulate_multi_kernel" in r[name]]
assert len(marks) > steps, "fewer proofs in the trace than asked for"
lo, hi = marks[-steps], marks[-1]
periods = steps - 1
acc = {}
for r in rows[lo:hi]:
    k = r[name].split("(")[0].replace("void ", "")
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in acc.values())
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / periods
print("kernel,launches_per_proof,ms_per_proof,percent_of_kernel_time")
for k, (c, ms) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%s,%.2f,%.4f,%.2f" % (k, c / periods, ms / periods, 100.0 * ms / tot))
print("TOTAL kernel time,%.2f,%.4f,100.00" % (sum(v[0] for v in acc.values()) / periods, tot / periods))
print("wall per proof under the profiler (deep launch to deep launch),,%.4f," % span)
