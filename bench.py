#!/usr/bin/env python3
"""bench.py — headline measurement of the MI355X-native Boojum proving hot path.

Workload (the one BASELINE.json's metric is quoted on, configs[3] "cfg4"): FULL PROVE of a SHA-256-shaped circuit with
2^22 rows — witness LDE + Poseidon2 Merkle tree, copy-permutation / lookup stage, quotient, openings, DEEP, FRI, queries —
with the reference bench's parameters (60 + 32 variable columns, 8 x width-4 lookups, LDE 8, cap 16, security 100, PoW off,
Poseidon2 tree hasher; src/gadgets/sha256/mod.rs:296-375).  The circuit is the SHA-shaped satisfiable synthetic circuit of
era_boojum_amd/synthetic.py (real SHA-256 synthesis needs the reference's Rust CS); prover cost is
data-independent.  One "step" = one proof; the witness is resident in HBM when the timed region starts (the span the
reference times is `prove_cpu_basic` only, sha256/mod.rs:514-527), the serialised proof is on the host when it ends.
`value` = constraints/sec := trace rows proved / wall seconds.  It fits one GPU, so N = 1 proves it on one MI355X; with
N > 1 the SAME proof is sharded over the N GPUs by LDE cosets (bj_setup_create_sharded: GPU g owns cosets
[g*8/N, (g+1)*8/N) = a contiguous range of Merkle leaves; cap fragments, the quotient evaluations, the first folded FRI
layer and the query openings are all-gathered over RCCL) and every rank ends with the identical proof: STRONG scaling
(total work fixed).  `--mode replicas` instead lets every rank prove its own instance (weak scaling, no collective);
`--log-n 20` is BASELINE's cfg3.

Extra objects on the JSON line:
  roofline      the dominant kernel of a proof, the Poseidon2 leaf hashing of the witness tree: algorithmic bytes
                (8*W + 32 per leaf) / its duration measured with HIP events on the launch stream inside the
                timed proofs.  It is integer-VALU-bound by construction (~472 field multiplications per 64 absorbed bytes).
  ntt           BASELINE configs[1] ("cfg2"): 2^20 x 256-column forward NTT, algorithmic 16 B/element vs the HBM peak.
  stages_ms     per-round wall time, named after the reference's log lines.
  cpu_baseline  the C/Python oracle prover (restated reference CPU algorithm) on this box's host cores, on a smaller
                instance of the same circuit (bounded to ~10-30 s), in rows/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--mode", choices=["auto", "sharded", "replicas"], default="auto")
    ap.add_argument("--transcript", choices=["poseidon", "poseidon2", "blake2s", "keccak256"], default="poseidon",
                    help="poseidon = the bench script's GoldilocksPoisedonTranscript (v1 permutation, no KAT in the reference); "
                         "poseidon2 = the golden proof's transcript (pinned)")
    ap.add_argument("--circuit", choices=["sha256", "synthetic"], default="sha256",
                    help="sha256: the reference bench's real SHA-256 circuit over as many random message bytes as fit 2^log_n "
                         "rows (era_boojum_amd/sha256_circuit.py); synthetic: random satisfiable circuit of the same geometry")
    ap.add_argument("--fri-lde", type=int, default=8)
    ap.add_argument("--cap", type=int, default=16)
    ap.add_argument("--security", type=int, default=100)
    ap.add_argument("--cpu-log-n", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import era_boojum_amd as E
    from era_boojum_amd import synthetic as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    backend = os.environ.get("BJ_BENCH_BACKEND", "nccl")     # "gloo": several ranks sharing one GPU (functional check only)
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    log_n = args.log_n
    n = 1 << log_n
    table_bits = 4 if log_n >= 14 else 2
    sharded = world > 1 and args.mode != "replicas"
    seed = 42 if sharded else 42 + rank
    t_gen = time.perf_counter()
    if args.circuit == "sha256" and log_n >= 14:
        from era_boojum_amd import sha256_circuit as SHA
        msg_len = SHA.message_len_for_log_n(log_n)
        circuit = SHA.sha256_circuit(SHA.bench_message(msg_len, seed=seed))
        assert circuit.log_n == log_n
        circuit_name = "SHA-256 of %d random bytes (seed %d), synthesised like the reference bench (sha256/mod.rs:296-470)" % (msg_len, seed)
    else:
        circuit = S.sha_shaped_circuit(log_n, seed=seed, table_bits=table_bits)
        circuit_name = "SHA-shaped satisfiable synthetic (seed %d)" % seed
    t_gen = time.perf_counter() - t_gen
    ctx = E.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    comm = E.TorchComm(ctx) if sharded else None
    setup = E.ProverSetup(ctx, circuit, args.fri_lde, args.cap, args.security, comm=comm, transcript=args.transcript)
    # witness resident in HBM (torch owns the allocations)
    d_vars = torch.from_numpy(circuit.variables.view(np.int64)).to(dev)
    d_mult = torch.from_numpy(circuit.multiplicities.view(np.int64)).to(dev)

    def step():
        return setup.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())

    for _ in range(args.warmup):
        step()
    barrier()
    comm0 = (comm.calls, comm.bytes) if sharded else (0, 0)
    t0 = time.perf_counter()
    leaf_ms, stage_acc, proof_buf = [], {}, None
    for _ in range(args.steps):
        proof_buf, stages = step()
        if os.environ.get('BJ_BENCH_DEBUG'):
            print({k: round(v, 1) for k, v in stages.items()}, file=sys.stderr)
        leaf_ms.append(stages.pop("witness_tree_leaf_kernel"))
        for k, v in stages.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
    comm_calls = (comm.calls - comm0[0]) // args.steps if sharded else 0
    comm_mb = (comm.bytes - comm0[1]) / 1e6 / args.steps if sharded else 0.0

    rows_total = float(n) * (1 if sharded else world) * args.steps
    value = rows_total / elapsed
    W = circuit.num_vars + 1                                  # witness leaf width (variables + multiplicities)
    leaves = n * args.fri_lde // (world if sharded else 1)    # leaves hashed by ONE launch of the leaf kernel on this rank
    leaf_bytes = float(leaves) * (8 * W + 32)
    leaf_s = float(np.mean(leaf_ms)) / 1e3
    achieved = leaf_bytes / leaf_s / 1e9
    # HBM traffic of that kernel: PMC counters cannot be collected from inside this process; the committed summary of the
    # separate rocprofv3 --pmc passes over this same command (tools/pmc_bench.sh) is used when it matches the launch shape
    traffic, traffic_src, valu = None, None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_bench_2p22_leaf_traffic.json")))
        if abs(pm["WRITE_SIZE_KiB_mean"] * 1024 - leaves * 32.0) < 1.0 and W == 93:   # same leaves per launch, same width
            traffic = pm["traffic_bytes_per_launch"]
            traffic_src = "profiles/r01_pmc_bench_2p22_leaf_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024, rocprofv3 --pmc passes over this command"
            valu = pm.get("valu") or None
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "prover_constraints_per_sec",
        "value": round(value, 1),
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak" if (args.mode == "replicas") else "strong",   # --gpus N shards the SAME 2^22-row proof: total work fixed
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "%s: full prove of the SHA-256 circuit, 2^%d rows (92 variable + 1 multiplicity "
                               "columns, 8x4 lookups, LDE %d, cap %d, security %d, %s, PoW off)"
                               % ({22: "cfg4", 20: "cfg3"}.get(log_n, "custom"), log_n, args.fri_lde, args.cap, args.security,
                                  {"poseidon": "Poseidon2 tree hasher + Poseidon (v1) transcript", "poseidon2": "Poseidon2 tree hasher + Poseidon2 transcript",
                                   "blake2s": "Blake2s tree hasher + Blake2s transcript (the non-recursive configuration)",
                                   "keccak256": "Keccak256 tree hasher + Keccak256 transcript"}[args.transcript]),
                   "log_n": log_n, "rows": n, "circuit": circuit_name, "circuit_synthesis_s": round(t_gen, 1),
                   "sharding": ("one proof sharded by LDE cosets over %d GPUs (%d cosets = %d Merkle leaves each), all-gather of "
                                "caps / quotient / first FRI layer / query openings, %d collectives and %.1f MB received per "
                                "rank per proof" % (world, args.fri_lde // world, leaves, comm_calls, comm_mb))
                               if sharded else ("one GPU" if world == 1 else "one independent proof per rank (replicas), no data-path collective"),
                   "proof_bytes": int(proof_buf.size * 8)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src, "valu_pmc": valu,
                     "kernel": "bj::%s_leaves_kernel (witness tree: %d leaves x %d elements per launch)" % ({"blake2s": "blake2s", "keccak256": "keccak"}.get(args.transcript, "poseidon2"), leaves, W),
                     "kernel_ms": round(leaf_s * 1e3, 3), "algorithmic_bytes_per_launch": leaf_bytes,
                     "note": "integer-VALU-bound: ~472 Goldilocks multiplications per 64 absorbed bytes (DESIGN.md §4)"},
        "stages_ms": {k: round(v / args.steps, 3) for k, v in stage_acc.items()},
    }

    # ---- secondary leg: cfg2 NTT (2^20 x 256 columns), the "NTT GB/s vs HBM peak" half of the metric
    if not args.no_ntt:
        nlog, ncols = 20, 256
        src = torch.randint(0, 1 << 62, (ncols, 1 << nlog), dtype=torch.int64, device=dev)
        dst = torch.empty_like(src)
        for _ in range(2):
            ctx.ntt_forward_batch(src.data_ptr(), dst.data_ptr(), nlog, ncols, coset=7)
        reps = 10
        ctx.timer_start()
        for _ in range(reps):
            ctx.ntt_forward_batch(src.data_ptr(), dst.data_ptr(), nlog, ncols, coset=7)
        ms = ctx.timer_stop_ms() / reps
        nb = 16.0 * (1 << nlog) * ncols
        out["ntt"] = {"workload": "cfg2: forward NTT 2^20 x 256 columns, natural->bit-reversed, coset 7", "ms": round(ms, 4),
                      "achieved": round(nb / ms / 1e6, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": round(nb / ms / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": nb,
                      "kernels": "bj::ntt_strided8_kernel + bj::ntt_local12_kernel (HIP events on the launch stream)"}
        del src, dst

    if rank == 0:
        # parity spot check of what was just timed: the oracle's verifier restatement must accept the timed proof
        from era_boojum_amd import proof_format
        from oracle import verifier as OV
        pg = proof_format.parse(proof_buf, security_level=args.security)
        if not OV.verify(OV.VerificationKey(circuit, setup.cap(), args.fri_lde, args.cap), pg, transcript_kind=setup.transcript_kind):
            raise SystemExit("parity failure: the verifier restatement rejects the HIP proof")
        out["config"]["verified"] = "oracle/verifier.py accepts the last timed proof"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import prover as OP
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = min(threads, 64)
        if args.circuit == "sha256" and args.cpu_log_n >= 14:
            csmall = SHA.sha256_circuit(SHA.bench_message(SHA.message_len_for_log_n(args.cpu_log_n), seed=42))
        else:
            csmall = S.sha_shaped_circuit(args.cpu_log_n, seed=42, table_bits=4 if args.cpu_log_n >= 14 else 2)
        osetup = OP.Setup(csmall, args.fri_lde, args.cap, threads=threads)
        c0 = time.perf_counter()
        OP.prove(csmall, osetup, args.fri_lde, args.cap, security_level=args.security, threads=threads)
        t_cpu = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": round((1 << args.cpu_log_n) / t_cpu, 1), "unit": "rows/s", "cores": threads, "kind": "port",
                               "sample": "one proof of the same kind of circuit at 2^%d rows by the oracle prover (C bulk ops "
                                         "+ python orchestration, OpenMP, -O3 -march=x86-64-v3), %.1f s" % (args.cpu_log_n, t_cpu)}
    if rank == 0:
        print(json.dumps(out))
    barrier()           # rank 0 may still have been verifying / timing the CPU baseline: leave the group together
    setup.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
