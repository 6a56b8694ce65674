#!/usr/bin/env python3
"""HBM traffic of the witness-tree Poseidon2 leaf kernel from two rocprofv3 --pmc passes over bench.py.

Per MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced streaming reads, so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes.  WRITE_SIZE was
calibrated on this kernel: it equals leaves * 32 B (the digests) exactly.
In dispatch order the leaf kernel runs once for the setup tree, then three times per proof (witness, stage 2,
quotient); the witness-tree launches are the dispatches 1, 4, 7, ... of that kernel."""
import csv
import json
import sys


def leaf_dispatches(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "poseidon2_leaves_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                rows.append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


fetch = leaf_dispatches(sys.argv[1], "FETCH_SIZE")
write = leaf_dispatches(sys.argv[2], "WRITE_SIZE")
wit_f, wit_w = fetch[1::3], write[1::3]
mean = lambda x: sum(x) / len(x)
valu = {}
if len(sys.argv) > 3:   # third pass: SQ / GRBM counters -> how busy the integer VALU is during the same launches
    insts = leaf_dispatches(sys.argv[3], "SQ_INSTS_VALU")[1::3]
    gui = leaf_dispatches(sys.argv[3], "GRBM_GUI_ACTIVE")[1::3]
    if insts and gui:
        cyc = mean(gui) / 8.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        valu = {"SQ_INSTS_VALU_mean": mean(insts), "kernel_cycles": cyc,
                "valu_cycles_per_wave_instruction_per_simd": cyc * 1024 / mean(insts),
                "note": "1024 SIMDs; the instruction mix of this kernel costs ~3.1 issue cycles per wave64 instruction "
                        "(tools/microbench_ops.hip), so ~3.4 measured = the integer VALU is ~90 % busy"}
out = {
    "kernel": "bj::poseidon2_leaves_kernel, witness-tree launches",
    "launches": len(wit_f),
    "FETCH_SIZE_KiB_mean": mean(wit_f), "WRITE_SIZE_KiB_mean": mean(wit_w),
    "traffic_bytes_per_launch": (2 * mean(wit_f) + mean(wit_w)) * 1024,
    "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024  (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)",
    "all_leaf_dispatch_fetch_KiB": fetch[:10], "all_leaf_dispatch_write_KiB": write[:10],
    "valu": valu,
}
print(json.dumps(out, indent=1))
