#!/usr/bin/env python3
"""Predicted time of ONE 2^22-row proof on W GPUs from a single-GPU rocprofv3 kernel table: every kernel of
profiles/r<NN>_prover_2p22_kernel_stats.csv is put in one of three classes —
  replicated   main-domain work every rank repeats (inverse transforms to monomials, the stage-2 polynomials, weights, small trees),
  sharded      work on a rank's own cosets / leaves (LDE passes, leaf and node hashing, quotient terms, DEEP, first FRI fold),
  per-proof    host round trips and launch gaps = measured wall time - sum of kernel time (does not shrink),
and T(W) = replicated + sharded / W + per-proof + comm(W), comm from the bytes the library reports and a link rate.
This is arithmetic on measured single-GPU numbers, not a measurement: no multi-GPU box has run the sharded path yet.  Round 5
measured the COMPUTE side of it on one GPU (bench.py's scale_replay: one rank alone, peers replayed): 143.8 / 85.7 / 55.5 ms at
W = 2 / 4 / 8 before link time, i.e. 3.6 ms more per rank than the kernel table alone predicts — work that does not shrink with W
and is not in the table (launch gaps of ~330 launches, the latency-bound tails of small layers, host transcript round trips): the
per-proof constant is 6.6 ms since then (it was 3.0), which puts the model within 1 % of the three replayed times + its own link term.
    python tools/scale_model.py [profiles/r04_prover_2p22_kernel_stats.csv] [--proofs 7] [--wall-ms 261.2]"""
import argparse
import csv
import json

# kernels whose work is the same on every rank.  Inverse transforms are told from forward ones by their template argument
# (<false, ...> = no coset scaling = the main-domain inverse transform; the quotient's own inverse transform is sharded since round 4
# and small).  The first LDE pass reads its monomials once for all cosets: its read half (1/9 of its bytes) is charged as replicated.
REPLICATED = ("ntt_first4_kernel<false", "ntt_strided8_kernel<false", "ntt_local12_kernel<false", "ntt_strided4_kernel<false",
              "bitrev_scale", "copy_perm_rational", "chunk_prefix", "scan_", "lookup_polys", "barycentric_weights",
              "poseidon2_nodes_lanepar", "poseidon2_leaves_chunked_lanepar", "twiddle_kernel", "round_scale", "merkle_paths",
              "gather_", "inv_x_minus_one", "__amd_rocclr")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats", nargs="?", default="profiles/r04_prover_2p22_kernel_stats.csv")
    ap.add_argument("--proofs", type=int, default=7, help="proofs in the profiled run (steps + warmup)")
    ap.add_argument("--wall-ms", type=float, default=261.2, help="un-profiled single-GPU wall time per proof (bench.py)")
    ap.add_argument("--host-ms", type=float, default=6.6, help="per-proof time that neither shards nor shows in the kernel table "
                    "(fitted to the replayed ranks of round 5: bench.py scale_replay; 3.0 in round 4)")
    ap.add_argument("--link-gbps", type=float, default=150.0, help="sustained rate of one xGMI link")
    ap.add_argument("--mb-per-rank-w8", type=float, default=350.0, help="bytes arriving per rank per proof at W = 8")
    ap.add_argument("--replay", type=str, default=None, help="a bench.py JSON line (file) with scale_replay: printed next to the model")
    a = ap.parse_args()
    rep = sh = 0.0
    rows = list(csv.DictReader(open(a.stats)))
    for r in rows:
        ms = float(r["TotalDurationNs"]) / 1e6 / a.proofs
        name = r["Name"]
        if any(k in name for k in REPLICATED):
            rep += ms
        elif "ntt_first4_kernel<true" in name:      # coset-expanding pass: reads 1 word, writes L = 8
            rep += ms / 9.0
            sh += ms * 8.0 / 9.0
        else:
            sh += ms
    kernel_ms = rep + sh
    # the profiler slows the clocks: scale the kernel classes to the un-profiled wall time, keeping a fixed per-proof host part
    host_ms = a.host_ms
    scale = (a.wall_ms - host_ms) / kernel_ms
    rep, sh = rep * scale, sh * scale
    out = {"profiled_kernel_ms_per_proof": round(kernel_ms, 1), "scale_to_unprofiled": round(scale, 3), "replicated_ms": round(rep, 1),
           "sharded_ms": round(sh, 1), "host_ms": host_ms, "T": {}}
    for W in (1, 2, 4, 8):
        mb = 0.0 if W == 1 else a.mb_per_rank_w8 * (W - 1) / W * 8.0 / 7.0          # (W-1)/W of every all-gathered buffer arrives
        # a ring all-gather is bound by ONE link per rank (xGMI is point to point); a full-mesh exchange could use W - 1 of them
        ring = 0.0 if W == 1 else mb / a.link_gbps + 11 * 0.03
        mesh = 0.0 if W == 1 else mb / (a.link_gbps * min(W - 1, 7)) + 11 * 0.03
        t = rep + sh / W + host_ms + ring
        out["T"][str(W)] = {"ms": round(t, 1), "speedup": round(a.wall_ms / t, 2), "efficiency": round(a.wall_ms / t / W, 2),
                            "comm_ms_ring_bound": round(ring, 1), "comm_ms_full_mesh_bound": round(mesh, 1),
                            "mb_arriving_per_rank": round(mb, 0)}
    if a.replay:
        line = [ln for ln in open(a.replay) if ln.startswith("{")][-1]
        sr = json.loads(line).get("scale_replay", {}).get("worlds", {})
        for w, v in sr.items():
            if w in out["T"] and "max_ms" in v:
                t = out["T"][w]
                t["replayed_compute_ms"] = v["max_ms"]
                t["replayed_plus_model_link_ms"] = round(v["max_ms"] + t["comm_ms_ring_bound"], 1)
                t["model_over_measured"] = round(t["ms"] / t["replayed_plus_model_link_ms"], 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
