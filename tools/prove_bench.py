#!/usr/bin/env python3
"""Full-prover timing on one GPU, bj_prove_dev with the witness resident in HBM.
    python tools/prove_bench.py --log-n 16 [--reps 3] [--verify]              random circuit of the SHA bench's geometry
    python tools/prove_bench.py --message-bytes 8192 --transcript blake2s    the reference's own test: SHA-256 of 8 KiB
                                                                              (2^16 rows), non-recursive configuration"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import era_boojum_amd as E
from era_boojum_amd import synthetic as S, proof_format


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fri-lde", type=int, default=8)
    ap.add_argument("--cap", type=int, default=16)
    ap.add_argument("--security", type=int, default=100)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--message-bytes", type=int, default=0, help="prove the real SHA-256 circuit of this many random bytes")
    ap.add_argument("--transcript", default="poseidon2", choices=["poseidon2", "poseidon", "blake2s", "keccak256"])
    a = ap.parse_args()
    t0 = time.time()
    if a.message_bytes:
        from era_boojum_amd import sha256_circuit as SHA
        c = SHA.sha256_circuit(SHA.bench_message(a.message_bytes))
        a.log_n = c.log_n
    else:
        c = S.sha_shaped_circuit(a.log_n, seed=42, table_bits=4 if a.log_n >= 14 else 2)
    t_gen = time.time() - t0
    ctx = E.Context(0)
    t0 = time.time()
    setup = E.ProverSetup(ctx, c, a.fri_lde, a.cap, a.security, transcript=a.transcript)
    ctx.sync()
    t_setup = time.time() - t0
    d_vars = ctx.upload(c.variables)
    d_mult = ctx.upload(c.multiplicities)
    times, stages = [], None
    for r in range(a.reps + 1):
        ctx.sync()
        t0 = time.time()
        buf, stage_ms = setup.prove_dev(d_vars, d_mult)
        dt = time.time() - t0
        if r > 0:
            times.append(dt)
            stages = stage_ms
    best = min(times)
    out = {"log_n": a.log_n, "gen_s": round(t_gen, 2), "setup_s": round(t_setup, 3), "prove_ms_best": round(best * 1e3, 2),
           "prove_ms_all": [round(x * 1e3, 2) for x in times], "rows_per_s": round((1 << a.log_n) / best, 1),
           "proof_bytes": int(buf.size * 8), "stages_ms": {k: round(v, 2) for k, v in stages.items()}}
    if a.verify:
        from oracle import verifier as OV
        pg = proof_format.parse(buf, security_level=a.security)
        t0 = time.time()
        out["verifier_accepts"] = bool(OV.verify(OV.VerificationKey(c, setup.cap(), a.fri_lde, a.cap), pg, verbose=True, transcript_kind=setup.transcript_kind))
        out["verify_s"] = round(time.time() - t0, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
