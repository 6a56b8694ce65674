"""One rank's critical path of a W-GPU sharded proof, measured on ONE GPU (bench.py --replay-world).

An 8-GPU node is not available to a build round, but everything a rank of the sharded prover does between two collectives can
be timed on one device, because every rank of a sharded proof receives the same gathered bytes and the proof of one witness is
deterministic:

  pass 1  the recording, ONE rank on the device at a time (round 6; until round 5 the W ranks ran side by side as W threads with W
          workspaces, which a 288 GB device cannot hold at 2^23 rows: every rank keeps the replicated monomials).
          Collective k of the proof is recorded when collectives 0 .. k-1 are: rank r proves behind a replay transport that serves
          those and CAPTURES its contribution to collective k, which ends that proof (bj_comm_replay_capture); the W contributions,
          rank-major, are the gathered buffer of collective k.  K + 1 sweeps over the ranks (K collectives: 11 per proof), each
          proof cut short at the collective it is after; in the last sweep every rank runs to the end and its proof must be the
          single-GPU proof.  The W setups stay resident, the workspace is one context's.
  pass 2  rank r ALONE with the complete recording — each all-gather a device-to-device copy of the recorded buffer on the proof's
          stream.  First one verifying proof (every contribution of the rank compared with the recording, proof bytes compared),
          then the timed ones.

What the number contains: the rank's kernels, launches, host round trips and transcript work, i.e. T(W) without link time and
without waiting for slower peers; what it lacks is exactly `comm` of DESIGN.md §6 (cap fragments, quotient residues, first FRI
layer, DEEP numerator, query openings over xGMI).  It is hardware evidence for the compute side of the scaling model, not a
scaling curve."""
import time

import numpy as np


def record(ctx, setups, world, d_vars, d_mult, capture_bytes, device):
    """Pass 1: the gathered buffer of every collective of one proof, by re-execution (see the module text).  Returns (list of torch
    uint8 tensors, the ranks' proofs)."""
    import torch
    import era_boojum_amd as E
    dev = torch.device("cuda", device)
    cap_buf = torch.empty(capture_bytes, dtype=torch.uint8, device=dev)
    gathered, proofs = [], [None] * world
    while True:
        parts, finished = [], 0
        rec = [(t.data_ptr(), t.numel()) for t in gathered]
        for r in range(world):
            comm = E.ReplayComm(ctx, r, world, rec, keepalive=gathered, capture=(cap_buf.data_ptr(), capture_bytes))
            setups[r].set_comm(comm)
            try:
                proofs[r], _ = setups[r].prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
                finished += 1
            except E.BoojumHipError:
                nb = comm.captured
                if not nb:
                    raise
                parts.append(cap_buf[:nb].clone())
            finally:
                setups[r]._replay_keepalive = comm      # the setup points at this transport until the next one is installed
        if finished == world:
            return gathered, proofs
        if finished or len(set(p.numel() for p in parts)) != 1:
            raise RuntimeError("recording: the ranks disagree on collective %d" % len(gathered))
        gathered.append(torch.cat(parts))
        if len(gathered) > 64:
            raise RuntimeError("recording: more than 64 collectives in one proof")


def measure(circuit, world, fri_lde=8, cap=16, security=100, transcript="poseidon2", steps=3, warmup=1, device=0, ranks=None,
            reference_proof=None, d_vars=None, d_mult=None, setup_cap=None):
    """Returns {"world", "ranks": {r: {"ms_per_step", "stages_ms", "ms_in_replayed_copies", "setup_bytes", "workspace"}}, "max_ms",
    "slowest_rank", "min_ms", "collectives_per_proof", "mb_gathered_per_proof", "record_pass_s"}.  Raises if any rank's proof differs
    from `reference_proof` (the single-GPU bytes) or a replayed contribution differs from the recording.  setup_cap: the setup
    oracle's cap (4 * cap u64, e.g. ProverSetup.cap() of a single-GPU setup): what the one collective of bj_setup_create_sharded
    gathers; computed through a single-GPU setup when None."""
    import torch
    import era_boojum_amd as E
    dev = torch.device("cuda", device)
    own = d_vars is None
    if own:
        d_vars = torch.from_numpy(circuit.variables.view(np.int64)).to(dev)
        d_mult = torch.from_numpy(circuit.multiplicities.view(np.int64)).to(dev)
    stream = torch.cuda.Stream(device=dev)
    ctx = E.Context(device, stream=stream.cuda_stream)
    setups = [None] * world
    try:
        if setup_cap is None:
            s1 = E.ProverSetup(ctx, circuit, fri_lde, cap, security, transcript=transcript)
            setup_cap = s1.cap()
            s1.close()
        d_cap = torch.from_numpy(np.ascontiguousarray(setup_cap, dtype=np.uint64).view(np.uint8).copy()).to(dev)
        t0 = time.perf_counter()
        for r in range(world):
            comm = E.ReplayComm(ctx, r, world, [(d_cap.data_ptr(), d_cap.numel())], n_setup=1, keepalive=d_cap)
            setups[r] = E.ProverSetup(ctx, circuit, fri_lde, cap, security, comm=comm, transcript=transcript)
            setups[r]._replay_keepalive = comm
            if not np.array_equal(np.asarray(setups[r].cap()).reshape(-1), np.asarray(setup_cap, dtype=np.uint64).reshape(-1)):
                raise RuntimeError("rank %d: the sharded setup's cap differs from the single-GPU one" % r)
        n = 1 << circuit.log_n
        capture_bytes = 2 * circuit.quotient_degree * n * 8 // world + (1 << 22)
        recorded_t, proofs = record(ctx, setups, world, d_vars, d_mult, capture_bytes, device)
        record_s = time.perf_counter() - t0
        for r in range(1, world):
            if not np.array_equal(proofs[r], proofs[0]):
                raise RuntimeError("rank %d of the recording pass ended with another proof than rank 0" % r)
        if reference_proof is not None and not np.array_equal(proofs[0], reference_proof):
            raise RuntimeError("the sharded proof of the recording pass differs from the single-GPU proof")
        recorded = [(t.data_ptr(), t.numel()) for t in recorded_t]
        out = {"world": world, "ranks": {}, "collectives_per_proof": len(recorded),
               "mb_gathered_per_proof": round(sum(b for _, b in recorded) / 1e6, 2), "record_pass_s": round(record_s, 2),
               "recording": "one rank on the device at a time: %d sweeps of partial proofs" % (len(recorded) + 1)}
        for r in (range(world) if ranks is None else ranks):
            st = setups[r]
            check = E.ReplayComm(ctx, r, world, recorded, verify=True, keepalive=recorded_t)
            st.set_comm(check)
            got, _ = st.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())
            calls, _, bad = check.stats()
            if bad or calls != len(recorded):
                raise RuntimeError("rank %d alone: %d of %d contributions differ from the recording" % (r, bad, calls))
            if not np.array_equal(got, proofs[0]):
                raise RuntimeError("rank %d alone, peers replayed: another proof than the %d-rank run" % (r, world))
            fast = E.ReplayComm(ctx, r, world, recorded, verify=False, keepalive=recorded_t)
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
            if fast.mismatches:
                raise RuntimeError("rank %d alone: the first collective of a timed proof differs from the recording" % r)
            out["ranks"][r] = {"ms_per_step": round(ms, 3), "stages_ms": {k: round(v / steps, 3) for k, v in acc.items()},
                               "ms_in_replayed_copies": round(copies / steps, 3), "setup_bytes": st.device_bytes(),
                               "workspace": dict(st.last_workspace)}
            st.set_comm(check)                   # keep a live transport in the setup until it is closed
            fast.close()
            st._replay_keepalive = check
        times = {r: v["ms_per_step"] for r, v in out["ranks"].items()}
        slow = max(times, key=times.get)
        out.update(max_ms=times[slow], slowest_rank=slow, min_ms=min(times.values()))
        return out
    finally:
        for st in setups:
            if st is not None:
                st.close()
        try:
            ctx.release_workspace()
        except Exception:
            pass
        ctx.close()
        if own:
            del d_vars, d_mult
        torch.cuda.empty_cache()
