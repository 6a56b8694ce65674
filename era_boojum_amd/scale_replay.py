"""One rank's critical path of a W-GPU sharded proof, measured on ONE GPU (bench.py --replay-world).

An 8-GPU node is not available to a build round, but everything a rank of the sharded prover does between two collectives can
be timed on one device, because every rank of a sharded proof receives the same gathered bytes and the proof of one witness is
deterministic:

  pass 1  the W ranks of ONE proof run as W threads of this process on the one GPU (binding.ThreadGroup: own context, stream and
          arena per rank, the all-gather a rendezvous + device copies); rank 0 keeps a copy of every gathered buffer.  Timing
          meaningless (the ranks time-slice the device); every rank's proof must be the single-GPU proof.
  pass 2  rank r ALONE: its setup's transport is replaced by the recorded-peer one (bj_comm_replay_create through
          bj_setup_set_comm) — each all-gather becomes a device-to-device copy of the recorded buffer on the proof's stream.
          First one verifying proof (every contribution of the rank compared with the recording, proof bytes compared), then
          the timed ones.

What the number contains: the rank's kernels, launches, host round trips and transcript work, i.e. T(W) without link time and
without waiting for slower peers; what it lacks is exactly `comm` of DESIGN.md §6 (cap fragments, quotient residues, first FRI
layer, DEEP numerator, query openings over xGMI).  It is hardware evidence for the compute side of the scaling model, not a
scaling curve."""
import threading
import time

import numpy as np


def measure(circuit, world, fri_lde=8, cap=16, security=100, transcript="poseidon2", steps=3, warmup=1, device=0, ranks=None,
            reference_proof=None, d_vars=None, d_mult=None):
    """Returns {"world", "ranks": {r: {"ms_per_step", "stages_ms", "ms_in_replayed_copies"}}, "max_ms", "slowest_rank", "min_ms",
    "collectives_per_proof", "mb_gathered_per_proof", "record_pass_s"}.  Raises if any rank's proof differs from `reference_proof`
    (the single-GPU bytes) or a replayed contribution differs from the recording."""
    import torch
    import era_boojum_amd as E
    dev = torch.device("cuda", device)
    own = d_vars is None
    if own:
        d_vars = torch.from_numpy(circuit.variables.view(np.int64)).to(dev)
        d_mult = torch.from_numpy(circuit.multiplicities.view(np.int64)).to(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
    ctxs = [E.Context(device, stream=s.cuda_stream) for s in streams]
    group = E.ThreadGroup(world, record=True)
    setups, proofs, errors = [None] * world, [None] * world, []
    sync = threading.Barrier(world)

    def rank_main(r):
        try:
            comm = group.comm(ctxs[r], r)
            setups[r] = E.ProverSetup(ctxs[r], circuit, fri_lde, cap, security, comm=comm, transcript=transcript)
            sync.wait()
            if r == 0:
                group.mark()                     # what was gathered so far belongs to bj_setup_create_sharded
            sync.wait()
            proofs[r], _ = setups[r].prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
        except Exception as e:                   # noqa: BLE001 — reported below, after the other threads have been released
            errors.append((r, e))
            for b in (sync, group._barrier):
                try:
                    b.abort()
                except Exception:
                    pass

    t0 = time.perf_counter()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    record_s = time.perf_counter() - t0
    try:
        if errors or group.error is not None:
            raise RuntimeError("recording pass failed: %r" % (errors or group.error))
        for r in range(1, world):
            if not np.array_equal(proofs[r], proofs[0]):
                raise RuntimeError("rank %d of the recording pass ended with another proof than rank 0" % r)
        if reference_proof is not None and not np.array_equal(proofs[0], reference_proof):
            raise RuntimeError("the sharded proof of the recording pass differs from the single-GPU proof")
        per_proof = group.recorded[group.marks[0]:]
        recorded = [(t.data_ptr(), t.numel()) for t in per_proof]
        out = {"world": world, "ranks": {}, "collectives_per_proof": len(recorded),
               "mb_gathered_per_proof": round(sum(b for _, b in recorded) / 1e6, 2), "record_pass_s": round(record_s, 2)}
        for r in (range(world) if ranks is None else ranks):
            st = setups[r]
            check = E.ReplayComm(ctxs[r], r, world, recorded, verify=True)
            st.set_comm(check)
            got, _ = st.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
            calls, _, bad = check.stats()
            if bad or calls != len(recorded):
                raise RuntimeError("rank %d alone: %d of %d contributions differ from the recording" % (r, bad, calls))
            if not np.array_equal(got, proofs[0]):
                raise RuntimeError("rank %d alone, peers replayed: another proof than the %d-rank run" % (r, world))
            fast = E.ReplayComm(ctxs[r], r, world, recorded, verify=False)
            st.set_comm(fast)
            for _ in range(warmup):
                st.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
            torch.cuda.synchronize(dev)
            acc, copies = {}, 0.0
            t0 = time.perf_counter()
            for _ in range(steps):
                _, stages = st.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
                copies += st.last_comm["ms_in_collectives"]
                for k, v in stages.items():
                    acc[k] = acc.get(k, 0.0) + v
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / steps * 1e3
            out["ranks"][r] = {"ms_per_step": round(ms, 3), "stages_ms": {k: round(v / steps, 3) for k, v in acc.items()},
                               "ms_in_replayed_copies": round(copies / steps, 3)}
            st.set_comm(check)                   # keep a live transport in the setup until it is closed
            fast.close()
            setups[r]._replay_keepalive = check
        times = {r: v["ms_per_step"] for r, v in out["ranks"].items()}
        slow = max(times, key=times.get)
        out.update(max_ms=times[slow], slowest_rank=slow, min_ms=min(times.values()))
        return out
    finally:
        for st in setups:
            if st is not None:
                st.close()
        for c in ctxs:
            try:
                c.release_workspace()
            except Exception:
                pass
            c.close()
        group.recorded.clear()
        if own:
            del d_vars, d_mult
        torch.cuda.empty_cache()
