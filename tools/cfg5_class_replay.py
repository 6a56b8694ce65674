#!/usr/bin/env python3
"""BASELINE config 5's circuit CLASS on the sharded path: the recursion-class circuit (155 variable columns, the golden proof's eleven
evaluators over general-purpose columns incl. the 118-term Poseidon2 flattened gate, a gate over a specialized column, 8 x width-3
lookups, quotient degree 8; era_boojum_amd/synthetic.py::recursion_like_circuit) proved on one GPU and as one rank of W alone behind
recorded peers (era_boojum_amd/scale_replay.py).  Prints one JSON object.
    python tools/cfg5_class_replay.py [--log-n 22] [--world 8] [--steps 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import era_boojum_amd as E  # noqa: E402
from era_boojum_amd import proof_format, scale_replay, synthetic as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--world", type=str, default="8")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--security", type=int, default=100)
    a = ap.parse_args()
    t0 = time.perf_counter()
    c = S.recursion_like_circuit(a.log_n, seed=7)
    t_gen = time.perf_counter() - t0
    ctx = E.Context(0)
    setup = E.ProverSetup(ctx, c, 8, 16, a.security)
    d_vars = torch.from_numpy(np.ascontiguousarray(np.concatenate([c.variables, c.witness], axis=0) if getattr(c, "num_witness_cols", 0) else c.variables).view(np.int64)).cuda()
    d_mult = torch.from_numpy(c.multiplicities.view(np.int64)).cuda()
    buf, _ = setup.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        buf, stages = setup.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t1) / a.steps * 1e3
    out = {"circuit": "recursion-class: %d variable columns, %d constant columns, quotient degree %d, %d x width-%d lookups, 2^%d rows; LDE 8, cap 16, security %d"
                      % (c.num_vars, c.num_constant_cols, c.quotient_degree, c.lookup_reps, c.lookup_width, a.log_n, a.security),
           "synthesis_s": round(t_gen, 1), "single_gpu_ms": round(single_ms, 3), "single_gpu_stages_ms": {k: round(v, 3) for k, v in stages.items()},
           "single_gpu_hbm_gb": {"setup": round(setup.device_bytes() / 1e9, 2), "workspace_high_water": round(setup.last_workspace["high_water_bytes"] / 1e9, 2)},
           "rows_per_s": round((1 << a.log_n) / single_ms * 1e3, 1)}
    from oracle import verifier as OV
    out["verified"] = bool(OV.verify(OV.VerificationKey(c, setup.cap(), 8, 16), proof_format.parse(buf, security_level=a.security)))
    cap = setup.cap()
    setup.close()
    ctx.release_workspace()
    out["worlds"] = {}
    for w in [int(x) for x in a.world.split(",") if x]:
        r = scale_replay.measure(c, w, 8, 16, a.security, "poseidon2", steps=a.steps, warmup=1, device=0, reference_proof=buf,
                                 d_vars=d_vars, d_mult=d_mult, setup_cap=cap)
        out["worlds"][str(w)] = {"max_ms": round(r["max_ms"], 3), "min_ms": round(r["min_ms"], 3), "speedup_compute_only": round(single_ms / r["max_ms"], 3),
                                 "collectives_per_proof": r["collectives_per_proof"], "mb_gathered_per_proof": round(r["mb_gathered_per_proof"], 2),
                                 "hbm_per_rank_gb": {"setup": round(max(v["setup_bytes"] for v in r["ranks"].values()) / 1e9, 2),
                                                     "workspace_high_water": round(max(v["workspace"]["high_water_bytes"] for v in r["ranks"].values()) / 1e9, 2)},
                                 "what": "rank r of W alone on this GPU behind recorded peers, max over ranks; every replayed proof = the single-GPU bytes"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
