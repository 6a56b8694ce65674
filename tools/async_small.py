#!/usr/bin/env python3
"""Serial bj_prove against the pipelined bj_prove_async / bj_proof_wait at small trace lengths (latency-bound proofs): ms per proof.
    python tools/async_small.py [log_n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import era_boojum_amd as E
from era_boojum_amd import synthetic as S

ctx = E.Context(0)
for log_n in [int(x) for x in sys.argv[1:]] or [14, 16, 18, 20]:
    c = S.sha_shaped_circuit(log_n, seed=5, table_bits=4 if log_n >= 14 else 2)
    setup = E.ProverSetup(ctx, c, 8, 16, 100)
    hv = torch.from_numpy(c.variables.view(np.int64)).pin_memory().numpy().view(np.uint64)
    hm = torch.from_numpy(c.multiplicities.view(np.int64)).pin_memory().numpy().view(np.uint64)
    ref, _ = setup.prove(variables=hv, multiplicities=hm)
    k = int(os.environ.get("BJ_ASYNC_K", "40" if log_n <= 18 else "16"))
    t0 = time.perf_counter()
    for _ in range(k):
        setup.prove(variables=hv, multiplicities=hm)
    serial = (time.perf_counter() - t0) / k * 1e3
    setup.wait(setup.prove_async(variables=hv, multiplicities=hm))          # warm-up of both lanes, the second one overlapped
    tw = setup.prove_async(variables=hv, multiplicities=hm)
    setup.wait(setup.prove_async(variables=hv, multiplicities=hm))
    setup.wait(tw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = setup.prove_async(variables=hv, multiplicities=hm)
    ok, first = True, None
    for _ in range(k - 1):
        cur = setup.prove_async(variables=hv, multiplicities=hm)
        got, stages = setup.wait(prev)
        ok = ok and np.array_equal(got, ref)
        if first is None:
            first, first_stages = (time.perf_counter() - t0) * 1e3, {k2: round(v, 1) for k2, v in stages.items()}
        prev = cur
    ok = ok and np.array_equal(setup.wait(prev)[0], ref)
    piped = (time.perf_counter() - t0) / k * 1e3
    print("2^%d rows: serial bj_prove %.3f ms, pipelined %.3f ms per proof (x%.2f), first completion %.0f ms, identical: %s"
          % (log_n, serial, piped, serial / piped, first, ok), flush=True)
    if first > 3.0 * serial:
        print("   slow first proof, its stages:", first_stages, flush=True)
    setup.close()
    ctx.release_workspace()
