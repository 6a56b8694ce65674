#!/usr/bin/env python3
"""A/B of the monomial layout at 2^22 rows: the same circuit proved with tiled monomials (default) and with natural-order ones
(BJ_MONO_TILED=0: inverse transforms end in bitrev_scale_tiled) — proofs must be byte-identical; prints ms per proof of both.
    python tools/tiled_ab.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import era_boojum_amd as E  # noqa: E402
from era_boojum_amd import synthetic as S  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = E.Context(0)
lib = E.load_library()
c = S.sha_shaped_circuit(22, seed=42, table_bits=4)
d_vars, d_mult = ctx.upload(c.variables), ctx.upload(c.multiplicities)
out = {}
for name, env in (("tiled", None), ("natural", "0"), ("tiled_again", None)):
    if env is None:
        os.environ.pop("BJ_MONO_TILED", None)
    else:
        os.environ["BJ_MONO_TILED"] = env
    lib.bj_env_reload()
    setup = E.ProverSetup(ctx, c, 8, 16, 100)
    buf, _ = setup.prove_dev(d_vars, d_mult)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        buf, st = setup.prove_dev(d_vars, d_mult)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    out[name] = (buf.copy(), ms, setup.cap().copy())
    print("%-12s %.3f ms per proof   stages %s" % (name, ms, {k: round(v, 2) for k, v in st.items()}), flush=True)
    setup.close()
same = np.array_equal(out["tiled"][0], out["natural"][0]) and np.array_equal(out["tiled"][2], out["natural"][2]) and np.array_equal(out["tiled"][0], out["tiled_again"][0])
print("proofs and setup caps identical:", same)
sys.exit(0 if same else 1)
