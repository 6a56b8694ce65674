#!/usr/bin/env python3
"""cfg2 (forward NTT 2^20 x 256 columns, coset 7): one JSON summary of the rocprofv3 passes tools/prof_cfg2.sh leaves in
gpurun_out/cfg2_<tag>_*.csv — per kernel: launch time, VALU wave-instructions, cycles per VALU instruction per SIMD, HBM traffic
((2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md) against the algorithmic bytes."""
import csv, json, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
def pmc(name):
    d = {}
    try:
        for r in csv.DictReader(open(os.path.join(root, "cfg2_%s_pmc_%s.csv" % (tag, name)))):
            d.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean_value"]), int(r["dispatches"]))
    except OSError:
        pass
    return d
sq, tcc, tccw = pmc("sq1"), pmc("tcc"), pmc("tccw")
stats = {}
for r in csv.DictReader(open(os.path.join(root, "cfg2_%s_kernel_stats.csv" % tag))):
    stats[r["Name"][:90]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
ALG = 16.0 * (1 << 20) * 256          # bytes one pass reads + writes
out = {"workload": "cfg2: forward NTT 2^20 x 256 columns, coset 7, 12 launches (tools/cfg2_ntt.py --cfg2-only)", "kernels": {}}
total_ms = 0.0
for k, (calls, ms) in stats.items():
    if "ntt_" not in k:
        continue
    e = {"launches": calls, "avg_ms": round(ms, 4), "algorithmic_bytes": ALG}
    total_ms += ms
    if k in sq and "SQ_INSTS_VALU" in sq[k]:
        valu, busy = sq[k]["SQ_INSTS_VALU"][0], sq[k].get("SQ_BUSY_CYCLES", (0, 0))[0]
        e["SQ_INSTS_VALU"] = valu
        e["SQ_BUSY_CYCLES"] = busy
        # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): /32 = the kernel's duration in shader clocks.
        # (GRBM_GUI_ACTIVE carries a per-dispatch overhead of the counter pass and is not used.)
        cyc = busy / 32.0
        e["kernel_cycles"] = cyc
        if valu and cyc:
            e["cycles_per_valu_instruction_per_simd"] = round(cyc * 1024 / valu, 3)
            e["implied_clock_GHz"] = round(cyc / (ms * 1e-3) / 1e9, 3)
            # issue cost of the butterfly mix (15 multiply-add / carry class at ~4.3 cycles + 6 plain at ~2.6 per 21): ~3.8
            e["valu_busy_estimate"] = round(3.8 / (cyc * 1024 / valu), 3)
    if k in tcc and k in tccw:
        f, w = tcc[k]["FETCH_SIZE"][0], tccw[k]["WRITE_SIZE"][0]
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = f, w
        e["traffic_bytes"] = (2 * f + w) * 1024
        e["traffic_over_algorithmic"] = round(e["traffic_bytes"] / ALG, 3)
    out["kernels"][k] = e
out["sum_of_passes_ms"] = round(total_ms, 4)
out["algorithmic_GBps"] = round(ALG / total_ms / 1e6, 1) if total_ms else None
out["frac_of_8TBps"] = round(ALG / total_ms / 1e6 / 8000.0, 4) if total_ms else None
print(json.dumps(out, indent=1))
