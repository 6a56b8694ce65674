#!/usr/bin/env python3
"""ONE proof of the synthetic bench-geometry circuit at 2^log_n (no warm-up), stage times and verification."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import era_boojum_amd as E
from era_boojum_amd import synthetic as S, proof_format
log_n = int(sys.argv[1])
c = S.sha_shaped_circuit(log_n, seed=42, table_bits=4)
ctx = E.Context(0)
setup = E.ProverSetup(ctx, c, 8, 16, 100)
d_vars, d_mult = ctx.upload(c.variables), ctx.upload(c.multiplicities)
t = time.time(); buf, st = setup.prove_dev(d_vars, d_mult); print("prove %.1f ms" % ((time.time() - t) * 1e3), {k: round(v, 2) for k, v in st.items()}, flush=True)
if "--verify" in sys.argv:
    from oracle import verifier as OV
    print("verifier accepts:", OV.verify(OV.VerificationKey(c, setup.cap(), 8, 16), proof_format.parse(buf, security_level=100), verbose=True))
