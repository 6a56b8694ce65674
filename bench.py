#!/usr/bin/env python3
"""bench.py — headline measurement of the MI355X-native Boojum proving hot path.

Workload (the one BASELINE.json's metric is quoted on, configs[3] "cfg4"): FULL PROVE of the SHA-256 circuit with 2^22 rows —
witness LDE + Poseidon2 Merkle tree, copy-permutation / lookup stage, quotient, openings, DEEP, FRI, queries — with the
reference bench's parameters (60 + 32 variable columns, 8 x width-4 lookups, LDE 8, cap 16, security 100, PoW off, Poseidon2
tree hasher; src/gadgets/sha256/mod.rs:296-375).  The circuit is the REAL SHA-256 circuit (era_boojum_amd/sha256_circuit.py
restates the reference's synthesis; the digest wired out of it equals hashlib's) over a seeded random message of as many bytes as
fit the rows — the input data are synthetic, the circuit is not (`--circuit synthetic` keeps the random satisfiable circuit of
the same geometry).  The default transcript is Poseidon2, the one the reference's golden proof pins bit for bit
(`"transcript_parity": "golden"`); `--transcript poseidon` is the bench script's Poseidon-v1 pairing, for which the
reference holds no known-answer vector ("unpinned"; same kernels, ~0.1 ms of host work apart).  One "step" = one proof; the
witness is resident in HBM when the timed region starts (the span the reference times is `prove_cpu_basic` only,
sha256/mod.rs:514-527), the serialised proof is on the host when it ends; `host_witness` reports the same proofs through
bj_prove, i.e. with the witness in (pinned) host memory and its PCIe transfer inside the timed region.
`value` = constraints/sec := trace rows proved / wall seconds.  It fits one GPU, so N = 1 proves it on one MI355X; with
N > 1 the SAME proof is sharded over the N GPUs by LDE cosets (bj_setup_create_sharded: GPU g owns cosets
[g*8/N, (g+1)*8/N) = a contiguous range of Merkle leaves; cap fragments, the quotient evaluations, the first folded FRI
layer and the query openings are all-gathered over RCCL — by the library itself on the proof's stream (comm_rccl.hip) after a
self-test of that transport that all ranks agree on; otherwise through the torch.distributed callback) and every rank ends with
the identical proof: STRONG scaling (total work fixed).  `--mode replicas` instead lets every rank prove its own instance (weak scaling, no collective);
`--log-n 20` is BASELINE's cfg3.

Extra objects on the JSON line:
  roofline      the dominant kernel of a proof, the Poseidon2 leaf hashing of the witness tree: algorithmic bytes
                (8*W + 32 per leaf) / its duration measured with HIP events on the launch stream inside the
                timed proofs.  It is integer-VALU-bound by construction (~472 field multiplications per 64 absorbed bytes).
  ntt           BASELINE configs[1] ("cfg2"): 2^20 x 256-column forward NTT, algorithmic 16 B/element vs the HBM peak.
  stages_ms     per-round wall time, named after the reference's log lines.
  kernels       gate evaluation, copy-permutation quotient, barycentric evaluation, DEEP and the first FRI fold: first launch of
                each inside every timed proof (HIP events on the launch stream), algorithmic bytes by SURVEY §8d vs the HBM peak.
  scale_replay  (N = 1) one rank's critical path of the SAME proof sharded over W = 2, 4, 8 GPUs, measured on this one GPU: the W ranks
                run once as threads sharing the device while every all-gathered buffer is recorded, then each rank runs ALONE with
                the recorded-peer transport (bj_comm_replay_create) serving its collectives as device copies.  ms per proof per
                rank, max over ranks; link time and waiting for peers EXCLUDED (DESIGN.md §6 adds ~2-3.5 ms for them); every
                replayed proof equals the single-GPU bytes.  Next to it the model of tools/scale_model.py.
  cpu_baseline  the C/Python oracle prover (restated reference CPU algorithm) on this box's host cores, on a smaller
                instance of the same circuit (2^20 rows = BASELINE cfg3's size, ~40 s; 2^18 as `micro.proof_2p18`), in rows/s; `micro` times the oracle's C primitives on all
                cores at the bench's own sizes (NTT 2^20 x 256, Poseidon2 tree 2^23 x 93: benches/benchmarks.rs:479-520, 73-79);
                `cores` = threads started, `cgroup_quota_cores` / `busy_cores_measured` = what the container actually gave them.
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


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one per GPU, through torch.distributed.run
    (the command the driver uses for N > 1) on a free local port, and hand its exit code back.  Fewer than N visible devices is
    an error — never a silent single-GPU run — unless BJ_BENCH_BACKEND=gloo asks for the functional mode in which ranks share
    GPUs."""
    import socket
    import subprocess
    if os.environ.get("BJ_BENCH_BACKEND", "nccl") != "gloo":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_ranks:
            print("bench.py --gpus %d: only %d GPU(s) visible; refusing to run on fewer devices than asked "
                  "(BJ_BENCH_BACKEND=gloo runs the ranks on shared GPUs as a functional check)" % (n_ranks, have), file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    if os.environ.get("BJ_BENCH_FAULTHANDLER_S"):       # debugging aid: dump every thread's stack if the run is still going after that long
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BJ_BENCH_FAULTHANDLER_S"]), repeat=False, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--mode", choices=["auto", "sharded", "replicas"], default="auto")
    ap.add_argument("--transcript", choices=["poseidon", "poseidon2", "blake2s", "keccak256"], default="poseidon2",
                    help="poseidon2 = the golden proof's transcript (pinned by the reference's proof.json); poseidon = the bench "
                         "script's GoldilocksPoisedonTranscript (v1 permutation, no KAT in the reference: parity unpinned)")
    ap.add_argument("--circuit", choices=["sha256", "synthetic"], default="sha256",
                    help="sha256: the reference bench's real SHA-256 circuit over as many random message bytes as fit 2^log_n "
                         "rows (era_boojum_amd/sha256_circuit.py); synthetic: random satisfiable circuit of the same geometry")
    ap.add_argument("--fri-lde", type=int, default=8)
    ap.add_argument("--cap", type=int, default=16)
    ap.add_argument("--security", type=int, default=100)
    ap.add_argument("--cpu-log-n", type=int, default=20)
    ap.add_argument("--cpu-micro-log-n", type=int, default=18, help="second, smaller oracle proof reported under cpu_baseline.micro")
    ap.add_argument("--bulk-transport", choices=["base", "peer"], default=os.environ.get("BJ_COMM_BULK", "base"),
                    help="N > 1: how the bulk exchanges (quotient residues, first folded FRI layer, DEEP numerator slices) of the TIMED proofs "
                         "travel: base = the transport chosen above (RCCL ring all-gather); peer = full-mesh peer copies (bj_comm_peer_create). "
                         "The other one is timed on a few extra proofs after the headline either way (comm.other_bulk_transport)")
    ap.add_argument("--test-die-rank", type=int, default=-1,
                    help="fault injection for tests/test_bench_launcher.py only: this rank exits right after the rendezvous")
    ap.add_argument("--no-other-bulk", action="store_true", help="N > 1: skip the extra proofs on the other bulk transport")
    ap.add_argument("--no-host-witness", action="store_true", help="skip the bj_prove (host witness, PCIe inclusive) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-two-in-flight", action="store_true", help="skip the secondary throughput leg with two proofs in flight on one GPU")
    ap.add_argument("--no-scale-replay", action="store_true", help="skip the one-rank-alone measurement of the sharded proof at W = 2, 4, 8")
    ap.add_argument("--replay-world", "--scale-replay", dest="replay_world", type=str, default="2,4,8",
                    help="world sizes of the scale_replay leg (era_boojum_amd/scale_replay.py): rank r of a W-rank proof alone on this "
                         "GPU with its peers' all-gathered buffers replayed from a recording")
    ap.add_argument("--replay-steps", type=int, default=3)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ and ("TORCHELASTIC_RUN_ID" in os.environ or int(os.environ["WORLD_SIZE"]) > 1)
    if args.gpus > 1 and not launched:
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import era_boojum_amd as E
    from era_boojum_amd import synthetic as S
    from era_boojum_amd import sha256_circuit as SHA

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and args.test_die_rank >= 0:
        # tests/test_bench_launcher.py (runs without a GPU): one rank dies right after the rendezvous, the others are left in a
        # collective — the launcher must notice and end the whole command with an error instead of hanging
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        tdist.init_process_group("gloo", rank=rank, world_size=world)
        if args.test_die_rank == rank:
            os._exit(3)
        tdist.barrier()
        raise SystemExit("the rank that was told to die is still alive")
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

    def synthesise():
        if args.circuit == "sha256" and log_n >= 14:
            msg_len = SHA.message_len_for_log_n(log_n)
            c = SHA.sha256_circuit(SHA.bench_message(msg_len, seed=seed))
            assert c.log_n == log_n
            return c, "SHA-256 of %d random bytes (seed %d), synthesised like the reference bench (sha256/mod.rs:296-470)" % (msg_len, seed)
        return S.sha_shaped_circuit(log_n, seed=seed, table_bits=table_bits), "SHA-shaped satisfiable synthetic (seed %d)" % seed

    circuit, circuit_name, circuit_from, cache = None, None, "synthesised by this rank", None
    if sharded:
        # every rank of a sharded proof needs the SAME circuit: rank 0 synthesises it once and leaves the big arrays in a directory
        # of plain .npy files (tmpfs when there is one); the other ranks map them read-only — one copy of the ~7 GB in host memory
        # and one synthesis instead of N of each.  Anything that goes wrong falls back to local synthesis.
        ok = [None]
        if rank == 0:
            try:
                import shutil
                circuit, circuit_name = synthesise()
                need = int(S.circuit_bytes(circuit) * 1.1) + (64 << 20)
                roots = [os.environ["BJ_BENCH_CIRCUIT_CACHE"]] if os.environ.get("BJ_BENCH_CIRCUIT_CACHE") else ["/dev/shm", "/tmp"]
                root = next((r for r in roots if os.path.isdir(r) and shutil.disk_usage(r).free > need), None)   # a container's /dev/shm is often 64 MB
                if root is None:
                    raise OSError("no directory with %.1f GB free among %s" % (need / 1e9, roots))
                cache = os.path.join(root, "bj_bench_circuit_%s_%d_%d_%d" % (args.circuit, log_n, seed, os.getuid()))
                S.save_circuit(circuit, cache, note=circuit_name)
                ok[0] = cache
            except Exception as e:                # noqa: BLE001
                print("rank 0: circuit cache not written (%r); every rank synthesises its own" % (e,), file=sys.stderr)
        dist.broadcast_object_list(ok, src=0)
        cache = ok[0]
        if rank != 0 and cache:
            try:
                circuit, circuit_name = S.load_circuit(cache)
                circuit_from = "mapped from rank 0's copy in %s" % cache
            except Exception as e:                # noqa: BLE001
                print("rank %d: circuit cache unreadable (%r); synthesising" % (rank, e), file=sys.stderr)
                circuit = None
    if circuit is None:
        circuit, circuit_name = synthesise()
    t_gen = time.perf_counter() - t_gen
    ctx = E.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    comm, transport = None, None
    if sharded:
        # the library's own transport: ncclAllGather on the proof stream, straight on the prover's buffers (comm_rccl.hip); the
        # 128-byte unique id travels through torch.distributed.  BJ_BENCH_TRANSPORT=torch keeps the host-callback transport
        # (also the automatic choice for gloo, where several ranks may share a GPU, which RCCL refuses).
        want = os.environ.get("BJ_BENCH_TRANSPORT", "rccl" if backend == "nccl" else "torch")
        if want == "rccl":
            # the unique id travels through torch.distributed; then a self-test before trusting the transport with a proof:
            # every rank contributes 1 KiB of its own pattern and checks the gathered block.  Whatever happens on a rank, ALL
            # ranks meet in one all-reduce (MIN) and keep the transport only if every one of them succeeded.
            ok_local, err = 0, None
            # bj_comm_rccl_create is itself a collective (ncclCommInitRank): no rank may enter it while another cannot, so the
            # ranks first agree that librccl loads everywhere
            avail = torch.tensor([E.load_library().bj_rccl_available()], dtype=torch.int32, device=dev)
            dist.all_reduce(avail, op=dist.ReduceOp.MIN)
            box = [None]
            if int(avail.item()) == 1:
                try:
                    box = [E.binding.rccl_unique_id() if rank == 0 else None]
                except Exception as e:
                    box, err = [None], e
                dist.broadcast_object_list(box, src=0)
            else:
                err = "librccl could not be loaded on every rank"
            if box[0] is not None:
                # ncclCommInitRank and the first collective are where a fabric / driver problem shows as a HANG, on hardware this
                # path has never met: they run in a helper thread under a watchdog.  A rank whose helper has not come back in time
                # votes "failed" in the all-reduce below (torch's own process group) and the run goes on over the callback
                # transport; the stuck helper is a daemon thread and dies with the process.
                import threading
                res = {}

                def bring_up():
                    try:
                        torch.cuda.set_device(local_rank)
                        c = E.RcclComm(ctx, box[0], rank, world)
                        mine = torch.full((128,), 0x0101010101010101 * (rank + 1), dtype=torch.int64, device=dev)
                        got = torch.zeros((world, 128), dtype=torch.int64, device=dev)
                        torch.cuda.synchronize()
                        c.all_gather(mine.data_ptr(), got.data_ptr(), 1024, stream=torch.cuda.current_stream().cuda_stream)
                        torch.cuda.synchronize()
                        want_blk = (torch.arange(1, world + 1, dtype=torch.int64, device=dev) * 0x0101010101010101).view(world, 1)
                        res["ok"] = bool((got == want_blk).all())
                        res["comm"] = c
                    except Exception as e:            # noqa: BLE001
                        res["err"] = e

                th = threading.Thread(target=bring_up, daemon=True)
                th.start()
                th.join(float(os.environ.get("BJ_BENCH_RCCL_TIMEOUT_S", "120")))
                if th.is_alive():
                    err = "in-library RCCL bring-up did not return within the watchdog"
                elif "err" in res:
                    err = res["err"]
                elif not res.get("ok"):
                    err = "self-test of the in-library all-gather returned wrong data"
                else:
                    comm, ok_local = res["comm"], 1
            flag = torch.tensor([ok_local], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                transport = "in-library RCCL all-gather on the proof stream"
            else:   # never lose the run to the transport: fall back to the host-callback one
                print("rank %d: in-library RCCL transport not used (%s); using torch.distributed" % (rank, err), file=sys.stderr)
                comm = None
        if comm is None:
            comm, transport = E.TorchComm(ctx), "torch.distributed all_gather_into_tensor through the bj_comm host callback"
        base_comm, peer_comm, peer_err = comm, None, None
        try:     # no collective of its own (the control channel rides on the default group); every rank reports, all agree
            peer_comm = E.PeerComm(ctx, base_comm)
            okp = 1
        except Exception as e:                    # noqa: BLE001
            peer_err, okp = repr(e)[:200], 0
        flagp = torch.tensor([okp], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flagp, op=dist.ReduceOp.MIN)
        if int(flagp.item()) != 1:
            peer_comm = None
        if args.bulk_transport == "peer" and peer_comm is not None:
            comm, transport = peer_comm, transport + " + full-mesh peer copies (HIP IPC) for the exchanges of >= 1 MiB per rank"
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
    leaf_ms, stage_acc, proof_buf, comm_ms_acc, kern_acc = [], {}, None, 0.0, {}
    for _ in range(args.steps):
        proof_buf, stages = step()
        if os.environ.get('BJ_BENCH_DEBUG'):
            print({k: round(v, 1) for k, v in stages.items()}, file=sys.stderr)
        leaf_ms.append(stages.pop("witness_tree_leaf_kernel"))
        comm_ms_acc += setup.last_comm["ms_in_collectives"]
        for kname, (kms, kbytes) in setup.last_kernels.items():
            acc = kern_acc.setdefault(kname, [0.0, kbytes])
            acc[0] += kms
        for k, v in stages.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        mine = {"rank": rank, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "stages_ms": {k: round(v / args.steps, 3) for k, v in stage_acc.items()},
                "leaf_kernel_ms": round(float(np.mean(leaf_ms)), 3), "circuit": circuit_from}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
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
    for prof in ("r06_pmc_bench_2p22_leaf_traffic.json", "r05_pmc_bench_2p22_leaf_traffic.json", "r04_pmc_bench_2p22_leaf_traffic.json", "r03_pmc_bench_2p22_leaf_traffic.json", "r02_pmc_bench_2p22_leaf_traffic.json"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", prof)))
            if abs(pm["WRITE_SIZE_KiB_mean"] * 1024 - leaves * 32.0) < 1.0 and W == 93:   # same leaves per launch, same width
                traffic = pm["traffic_bytes_per_launch"]
                traffic_src = ("NOT measured in this run: read from the committed profile profiles/%s = (2*FETCH_SIZE + WRITE_SIZE)*1024 of "
                               "separate rocprofv3 --pmc passes over this command (tools/pmc_bench.sh)" % prof)
                valu = pm.get("valu") or None
                break
        except (OSError, KeyError, ValueError):
            pass
    out = {
        "metric": "prover_constraints_per_sec",
        "value": round(value, 1),
        "unit": "rows/s",
        "n_gpus": dist.get_world_size() if dist is not None else 1,      # what the process group reports, not what was asked
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak" if (args.mode == "replicas") else "strong",   # --gpus N shards the SAME 2^22-row proof: total work fixed
        "vs_baseline": None,
        "dtype": "u64",
        "hbm_gb": {"what": "HBM this rank holds for the proof: its setup (bj_setup_device_bytes: natural columns, monomials, LDE shard, tree), the "
                           "proof workspace (bj_proof_workspace_bytes: reserved, high-water mark; overflow slabs must be 0) and the resident witness",
                   "setup": round(setup.device_bytes() / 1e9, 2),
                   "workspace_reserved": round(setup.last_workspace["reserved_bytes"] / 1e9, 2),
                   "workspace_high_water": round(setup.last_workspace["high_water_bytes"] / 1e9, 2),
                   "workspace_overflow_slabs": setup.last_workspace["overflow_slabs"],
                   "witness": round((d_vars.numel() + d_mult.numel()) * 8 / 1e9, 2)},
        "data": "synthetic input (seeded random message bytes) through the real SHA-256 circuit" if circuit_name.startswith("SHA-256")
                else "synthetic (random satisfiable circuit of the bench geometry)",
        "config": {"workload": "%s: full prove of the SHA-256 circuit, 2^%d rows (92 variable + 1 multiplicity "
                               "columns, 8x4 lookups, LDE %d, cap %d, security %d, %s, PoW off)"
                               % ({22: "cfg4", 20: "cfg3"}.get(log_n, "custom"), log_n, args.fri_lde, args.cap, args.security,
                                  {"poseidon": "Poseidon2 tree hasher + Poseidon (v1) transcript", "poseidon2": "Poseidon2 tree hasher + Poseidon2 transcript",
                                   "blake2s": "Blake2s tree hasher + Blake2s transcript (the non-recursive configuration)",
                                   "keccak256": "Keccak256 tree hasher + Keccak256 transcript"}[args.transcript]),
                   "transcript_parity": {"poseidon2": "golden (replays the reference's proof.json)", "poseidon": "unpinned (no KAT in the reference)",
                                         "blake2s": "hashlib.blake2s", "keccak256": "hashlib.sha3_256 (domain byte switched)"}[args.transcript],
                   "log_n": log_n, "rows": n, "circuit": circuit_name, "circuit_synthesis_s": round(t_gen, 1),
                   "sharding": ("one proof sharded by LDE cosets over %d GPUs (%d cosets = %d Merkle leaves each), all-gather of "
                                "caps / quotient / first FRI layer / query openings, %d collectives and %.1f MB received per "
                                "rank per proof; transport: %s" % (world, args.fri_lde // world, leaves, comm_calls, comm_mb, transport))
                               if sharded else ("one GPU" if world == 1 else "one independent proof per rank (replicas), no data-path collective"),
                   "proof_bytes": int(proof_buf.size * 8)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src, "valu_pmc_from_committed_profile": valu,
                     "kernel": "bj::%s_leaves_kernel (witness tree: %d leaves x %d elements per launch)" % ({"blake2s": "blake2s", "keccak256": "keccak"}.get(args.transcript, "poseidon2"), leaves, W),
                     "kernel_ms": round(leaf_s * 1e3, 3), "algorithmic_bytes_per_launch": leaf_bytes,
                     "note": "integer-VALU-bound: ~472 Goldilocks multiplications per 64 absorbed bytes (DESIGN.md §4)"},
        "stages_ms": {k: round(v / args.steps, 3) for k, v in stage_acc.items()},
    }
    # the other kernels north_star names, each against the HBM roofline: first launch of each inside every timed proof, HIP
    # events on the launch stream (bj_proof_kernel_stats), algorithmic bytes by SURVEY §8d; the VALU-bound ones say so
    KNOTE = {"quotient_gates": "bj::quotient_gates_windowed_kernel: 8 (gp columns + constants) qn + 16 qn",
             "quotient_copy_perm": "bj::quotient_copy_perm_kernel: 8 (2V + z + partial products + 1/(x-1)) qn + 32 qn",
             "barycentric_eval": "bj::barycentric_partial_kernel + final: 8 n per base column + 16 n weights (the set at z)",
             "deep_accumulate_multi": "bj::deep_accumulate_multi_kernel: 8 (#base columns) Ln + 16 Ln",
             "fri_fold_first": "bj::fri_fold_fused_kernel<k> of the first FRI oracle: 16 m (1 + 2^-k)"}
    out["kernels"] = {k: {"ms": round(v[0] / args.steps, 4), "algorithmic_bytes": v[1], "achieved": round(v[1] / (v[0] / args.steps) / 1e6, 1),
                          "unit": "GB/s", "frac": round(v[1] / (v[0] / args.steps) / 1e6 / HBM_PEAK_GBPS, 4), "what": KNOTE.get(k, k)}
                      for k, v in kern_acc.items() if v[0] > 0}
    if per_rank is not None:      # which rank set the time, and how far apart the ranks are, stage by stage
        slow = max(per_rank, key=lambda r: r["ms_per_step"])
        stages = sorted(per_rank[0]["stages_ms"])
        out["ranks"] = {"slowest_rank": slow["rank"], "ms_per_step_min": min(r["ms_per_step"] for r in per_rank),
                        "ms_per_step_max": slow["ms_per_step"],
                        "stages_ms_min": {k: min(r["stages_ms"].get(k, 0.0) for r in per_rank) for k in stages},
                        "stages_ms_max": {k: max(r["stages_ms"].get(k, 0.0) for r in per_rank) for k in stages},
                        "leaf_kernel_ms_min_max": [min(r["leaf_kernel_ms"] for r in per_rank), max(r["leaf_kernel_ms"] for r in per_rank)],
                        "circuit_source": sorted({r["circuit"].split(" in ")[0] for r in per_rank}),
                        "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or "all",
                        "note": "LOCAL_RANK indexes the devices this process SEES (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES are applied by "
                                "the runtime before torch and libboojum_hip enumerate them)"}
    if sharded:      # what a scaling record needs to explain itself: per proof on rank 0, and the slowest rank's time in collectives
        cm = torch.tensor([comm_ms_acc / args.steps], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        cmax = cm.clone()
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        out["comm"] = {"world": int(comm.world), "calls_per_proof": int(comm_calls), "mb_received_per_rank_per_proof": round(comm_mb, 2),
                        "ms_in_collectives_rank0": round(float(cm[0]), 3), "ms_in_collectives_max_rank": round(float(cmax[0]), 3),
                        "note": "time between the start and the end of each all-gather on the proof's stream (transfer + waiting for "
                                "the slowest peer), summed over a proof; replicated work is ~25 ms per rank, measured by the replayed ranks of scale_replay (DESIGN.md §6)",
                        "transport": transport}

    if sharded and peer_comm is not None and not args.no_other_bulk:
        # the OTHER bulk transport on a few extra proofs (after the headline's timed region, under a watchdog: a hang here must not cost the line)
        import threading
        other = base_comm if comm is peer_comm else peer_comm
        other_name = "base (ring all-gather)" if comm is peer_comm else "full-mesh peer copies (bj_comm_peer_create)"
        res2 = {}

        def other_leg():
            try:
                torch.cuda.set_device(local_rank)
                setup.set_comm(other)
                buf2, _ = step()
                k3, acc, t3 = 3, 0.0, time.perf_counter()
                for _ in range(k3):
                    buf2, _ = step()
                    acc += setup.last_comm["ms_in_collectives"]
                torch.cuda.synchronize()
                res2.update(ms_per_step=(time.perf_counter() - t3) / k3 * 1e3, ms_in_collectives=acc / k3, same=bool(np.array_equal(buf2, proof_buf)))
                setup.set_comm(comm)
            except Exception as e:                # noqa: BLE001
                res2["error"] = repr(e)[:300]

        th2 = threading.Thread(target=other_leg, daemon=True)
        th2.start()
        th2.join(float(os.environ.get("BJ_BENCH_OTHER_BULK_TIMEOUT_S", "60")))
        if th2.is_alive():
            res2 = {"error": "did not finish within the watchdog"}
        if "error" not in res2 and not res2.get("same"):
            res2["error"] = "proof differs from the timed one"
        out["comm"]["other_bulk_transport"] = dict(res2, transport=other_name, proofs=3,
                                                   peer_stats=peer_comm.stats() if "error" not in res2 else None,
                                                   note="rank 0's wall time and time in collectives per proof with the bulk exchanges on the other transport; "
                                                        "the timed headline used: " + transport)
        other_bulk_stuck = th2.is_alive()
        ran_other_bulk = True
    # ---- the drop-in call with a host witness (bj_prove): PCIe transfer of the 93 columns inside the timed region
    if world == 1 and not args.no_host_witness:
        hv = torch.from_numpy(circuit.variables.view(np.int64)).pin_memory()
        hm = torch.from_numpy(circuit.multiplicities.view(np.int64)).pin_memory()
        hvn, hmn = hv.numpy().view(np.uint64), hm.numpy().view(np.uint64)
        setup.prove(variables=hvn, multiplicities=hmn)
        torch.cuda.synchronize()
        hsteps = max(2, min(args.steps, 5))
        h0 = time.perf_counter()
        for _ in range(hsteps):
            hbuf, _ = setup.prove(variables=hvn, multiplicities=hmn)
        torch.cuda.synchronize()
        hms = (time.perf_counter() - h0) / hsteps * 1e3
        assert np.array_equal(hbuf, proof_buf), "bj_prove and bj_prove_dev disagree"
        out["host_witness"] = {"entry_point": "bj_prove (witness in pinned host memory, %.2f GB over PCIe per proof)" % (hvn.nbytes / 1e9 + hmn.nbytes / 1e9),
                               "ms_per_step": round(hms, 3), "value": round(n / hms * 1e3, 1), "unit": "rows/s", "steps": hsteps,
                               "overhead_vs_resident": round(hms / (elapsed / args.steps * 1e3) - 1.0, 4)}
        # the same drop-in call pipelined from this ONE host thread (bj_prove_async / bj_proof_wait, csrc/prove_async.hip): proof
        # k + 1's PCIe transfer, inverse transforms and first absorptions run under proof k's latency-bound tail
        # (in a helper thread under a watchdog like the other secondary legs: a lane that never came back must not cost the headline line)
        import threading
        pipe_stuck = False

        def pipelined_leg():
            try:
                torch.cuda.set_device(local_rank)
                psteps = max(6, min(4 * args.steps, 24))
                pbuf, _ = setup.wait(setup.prove_async(variables=hvn, multiplicities=hmn))      # warm-up of both lanes
                t_prev = setup.prove_async(variables=hvn, multiplicities=hmn)
                pbuf2, _ = setup.wait(setup.prove_async(variables=hvn, multiplicities=hmn))
                setup.wait(t_prev)
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                t_prev = setup.prove_async(variables=hvn, multiplicities=hmn)
                same, done_at = True, []
                for _ in range(psteps - 1):
                    t_cur = setup.prove_async(variables=hvn, multiplicities=hmn)
                    pb, _ = setup.wait(t_prev)
                    done_at.append(time.perf_counter())
                    same = same and np.array_equal(pb, proof_buf)
                    t_prev = t_cur
                pb, _ = setup.wait(t_prev)
                done_at.append(time.perf_counter())
                pms = (done_at[-1] - p0) / psteps * 1e3
                # proofs leave the pipeline at the steady rate from the first one on until the last but one; the last one has the device to
                # itself (drain) and the first one started on an idle device (fill): the interval between them is the rate a long-running host sees
                steady = (done_at[-2] - done_at[0]) / (psteps - 2) * 1e3
                assert same and np.array_equal(pb, proof_buf) and np.array_equal(pbuf, proof_buf) and np.array_equal(pbuf2, proof_buf), \
                    "a pipelined proof differs from the serial one"
                out["host_witness_pipelined"] = {
                    "entry_point": "bj_prove_async / bj_proof_wait from one host thread, two proofs in flight (witness in pinned host memory)",
                    "ms_per_proof_aggregate": round(pms, 3), "value": round(n / pms * 1e3, 1), "unit": "rows/s", "proofs": psteps,
                    "vs_resident_single_proof": round((n / pms * 1e3) / value, 4),
                    "steady_state_ms_per_proof": round(steady, 3), "steady_state_value": round(n / steady * 1e3, 1),
                    "steady_state_vs_resident_single_proof": round((n / steady * 1e3) / value, 4),
                    "first_completion_ms": round((done_at[0] - p0) * 1e3, 1), "last_interval_ms": round((done_at[-1] - done_at[-2]) * 1e3, 1),
                    "longest_interval_ms": round(max(b - a for a, b in zip(done_at, done_at[1:])) * 1e3, 1),
                    "what": "every proof equals the serial one byte for byte; `value` includes the fill and the drain of the two-deep pipeline "
                            "(first proof on an idle device, last proof alone); steady_state = completion of proof 1 to completion of proof N - 1"}
            except Exception as e:                    # noqa: BLE001 — secondary leg
                out["host_witness_pipelined"] = {"error": repr(e)[:300]}
        thp = threading.Thread(target=pipelined_leg, daemon=True)
        thp.start()
        thp.join(float(os.environ.get("BJ_BENCH_PIPELINED_TIMEOUT_S", "240")))
        if thp.is_alive():
            pipe_stuck = True
            out["host_witness_pipelined"] = {"error": "did not finish within the watchdog"}
        if not pipe_stuck:
            ctx.release_workspace()               # the two lanes' arenas and witness staging (2 x 65 GB at 2^22): the legs below need the room
        del hv, hm

    # ---- secondary leg: cfg2 NTT (2^20 x 256 columns), the "NTT GB/s vs HBM peak" half of the metric
    shared_device = world > 1 and backend == "gloo" and torch.cuda.device_count() < world
    if not args.no_ntt and shared_device:
        out["ntt"] = {"skipped": "ranks share a device (functional gloo mode): the stand-alone transform legs need ~30 GB of their own"}
    if not args.no_ntt and not shared_device:
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
        for prof in ("r06_cfg2_ntt_summary.json", "r05_cfg2_ntt_summary.json", "r04_cfg2_ntt_summary.json", "r03_cfg2_ntt_summary.json", "r02_cfg2_ntt_summary.json"):
            try:   # counters of the same two kernels from the committed rocprofv3 passes over tools/cfg2_ntt.py --cfg2-only
                cs = json.load(open(os.path.join(ROOT, "profiles", prof)))
                out["ntt"]["pmc_from_committed_profile"] = {
                    "source": "NOT measured in this run: profiles/%s (tools/prof_cfg2.sh)" % prof,
                    "kernels": {k.split("(")[0].replace("void ", ""): {f: v[f] for f in
                                ("avg_ms", "SQ_INSTS_VALU", "cycles_per_valu_instruction_per_simd", "valu_busy_estimate",
                                 "traffic_bytes", "traffic_over_algorithmic") if f in v} for k, v in cs["kernels"].items()}}
                # the VALU roofline next to the HBM one: the passes are issue-bound (traffic = algorithmic bytes), so the fraction
                # that says how close they are to THEIR ceiling is mix cost / measured cycles per wave instruction per SIMD
                ks = out["ntt"]["pmc_from_committed_profile"]["kernels"]
                out["ntt"]["valu_roofline_from_committed_profile"] = {
                    "bound": "valu (integer issue)", "unit": "cycles per wave64 VALU instruction per SIMD",
                    "mix_cost": 3.8, "measured": {k: v.get("cycles_per_valu_instruction_per_simd") for k, v in ks.items()},
                    "frac": {k: v.get("valu_busy_estimate") for k, v in ks.items()},
                    "note": "mix cost of a lazy butterfly from tools/microbench_ops.hip (6 v_mad_u64_u32 at ~4.3, 9 carry-class at ~4.4 "
                            "per pair, 6 plain at ~2.6); the part runs these kernels at ~1.85 GHz (power), DESIGN.md §5"}
                break
            except (OSError, KeyError, ValueError):
                pass
        del src, dst
        # the same transforms at the size and shape the proof runs them: the witness round's LDE, 93 columns x 2^log_n x 8 cosets
        # (coset-expanding front pass + strided + local pass), algorithmic bytes 8 n (1 + L) per column
        if log_n >= 20:
            lcols, L = 93, args.fri_lde
            mono = torch.randint(0, 1 << 62, (lcols, n), dtype=torch.int64, device=dev)
            lde = torch.empty((lcols, L, n), dtype=torch.int64, device=dev)
            log_L = L.bit_length() - 1
            tiled = ctx.monomials_tiled(log_n)      # the layout bj_prove keeps monomials of this size in (2^22: tiled, contiguous front-pass tiles)
            run_lde = (lambda: ctx.lde_cosets_batch_tiled(mono.data_ptr(), lde.data_ptr(), log_n, lcols, log_L, 0, L)) if tiled else \
                      (lambda: ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, lcols, log_L))
            run_lde()
            ctx.timer_start()
            for _ in range(3):
                run_lde()
            lms = ctx.timer_stop_ms() / 3
            lb = 8.0 * n * (1 + L) * lcols
            out["ntt"]["lde_at_bench_size"] = {"workload": "LDE %d columns x 2^%d x %d cosets (the witness round's), monomials in the %s layout as bj_prove keeps them"
                                                           % (lcols, log_n, L, "tiled" if tiled else "natural"),
                                               "ms": round(lms, 3), "achieved": round(lb / lms / 1e6, 2), "unit": "GB/s",
                                               "frac": round(lb / lms / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": lb}
            # the inverse transforms in front of it: values -> monomials, 93 columns (8 n bytes in + 8 n out per column)
            run_inv = (lambda: ctx.intt_batch_tiled(mono.data_ptr(), lde.data_ptr(), log_n, lcols)) if tiled else \
                      (lambda: ctx.intt_batch(mono.data_ptr(), lde.data_ptr(), log_n, lcols))
            run_inv()
            ctx.timer_start()
            for _ in range(3):
                run_inv()
            ims = ctx.timer_stop_ms() / 3
            ib = 16.0 * n * lcols
            out["ntt"]["intt_at_bench_size"] = {"workload": "inverse transform of %d columns x 2^%d into %s monomials%s" % (
                                                    lcols, log_n, "tiled" if tiled else "natural-order",
                                                    " (two passes, the bit reversal in the last pass's store addresses)" if tiled else " (passes + bit-reversal pass)"),
                                                "ms": round(ims, 3), "achieved": round(ib / ims / 1e6, 2), "unit": "GB/s",
                                                "frac": round(ib / ims / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes": ib}
            del mono, lde

    if rank == 0:
        # parity spot check of what was just timed: the oracle's verifier restatement must accept the timed proof
        from era_boojum_amd import proof_format
        from oracle import verifier as OV
        pg = proof_format.parse(proof_buf, security_level=args.security)
        if not OV.verify(OV.VerificationKey(circuit, setup.cap(), args.fri_lde, args.cap), pg, transcript_kind=setup.transcript_kind):
            raise SystemExit("parity failure: the verifier restatement rejects the HIP proof")
        out["config"]["verified"] = "oracle/verifier.py accepts the last timed proof"
    transcript_kind = setup.transcript_kind
    stuck_legs = []          # secondary legs whose helper threads did not return: the process then leaves through os._exit
    if locals().get("pipe_stuck"):
        stuck_legs.append("host_witness_pipelined")
    if locals().get("other_bulk_stuck"):
        stuck_legs.append("other bulk transport")
    if rank == 0 and world == 1 and not args.no_two_in_flight and not stuck_legs:
        # secondary throughput figure: TWO proofs of the same circuit in flight on this one GPU — two contexts, two HIP streams, two
        # host threads (the shape of tests/test_gpu_prover.py::test_two_contexts_on_two_host_threads_prove_concurrently).  The
        # metric is rows per second, and the VALU-bound hashing of one proof can share the CUs with the HBM-bound passes and the
        # host round trips of the other; the headline stays the latency of ONE proof.
        import threading
        try:
            s2 = torch.cuda.Stream(device=dev)
            ctx2 = E.Context(local_rank, stream=s2.cuda_stream)
            setup2 = E.ProverSetup(ctx2, circuit, args.fri_lde, args.cap, args.security, transcript=args.transcript)
            k2 = max(2, min(args.steps, 4))
            bufs = [None, None]

            def run(which, st):
                for _ in range(k2):
                    bufs[which], _ = st.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())

            run(1, setup2)                      # warm-up of the second context (arena, twiddles)
            setup.prove_dev(d_vars.data_ptr(), d_mult.data_ptr())      # and of the first one: its workspace was released above (a 63 GB hipMalloc takes 0.4 ms .. 1.8 s)
            torch.cuda.synchronize()
            th = [threading.Thread(target=run, args=(0, setup), daemon=True), threading.Thread(target=run, args=(1, setup2), daemon=True)]
            c0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join(120.0)
            if any(t.is_alive() for t in th):
                stuck_legs.append("throughput_2_in_flight")
                raise TimeoutError("two concurrent provers did not finish within the watchdog")
            torch.cuda.synchronize()
            dt = time.perf_counter() - c0
            assert np.array_equal(bufs[0], proof_buf) and np.array_equal(bufs[1], proof_buf), "concurrent proofs differ from the timed one"
            out["throughput_2_in_flight"] = {
                "value": round(2 * k2 * n / dt, 1), "unit": "rows/s", "proofs": 2 * k2, "ms_per_proof_aggregate": round(dt / (2 * k2) * 1e3, 3),
                "vs_one_in_flight": round((2 * k2 * n / dt) / value, 4),
                "what": "two contexts / streams / host threads proving the same circuit concurrently on one GPU; every proof equals the timed one"}
            setup2.close()
            ctx2.release_workspace()
            ctx2.close()
            del setup2, ctx2
            torch.cuda.empty_cache()
        except Exception as e:                    # noqa: BLE001 — secondary leg
            out["throughput_2_in_flight"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not args.no_scale_replay and log_n >= 12 and not stuck_legs:
        # one rank of the sharded proof alone on this GPU, peers replayed (the only multi-GPU evidence one GPU can give)
        from era_boojum_amd import scale_replay
        t1_ms = elapsed / args.steps * 1e3
        setup_cap = setup.cap()
        single_mem = {"setup_bytes": setup.device_bytes(), "workspace": dict(setup.last_workspace)}
        setup.close()                         # the single-GPU setup and arena make room for the W rank setups
        ctx.release_workspace()
        torch.cuda.empty_cache()
        sr = {"what": "rank r of a W-rank sharded proof ALONE on this GPU, every all-gather served by a device copy of the buffer "
                      "that collective gathers (recorded one collective at a time with one rank on the device: bj_comm_replay_capture); "
                      "ms per proof = kernels + launches + host round trips of one rank; link time and waiting for peers excluded; "
                      "every replayed proof = the single-GPU bytes; hbm_per_rank_gb = the rank's setup (bj_setup_device_bytes) + the "
                      "high-water mark of its proof workspace (bj_proof_workspace_bytes)",
              "single_gpu_ms": round(t1_ms, 3), "steps": args.replay_steps,
              "single_gpu_hbm_gb": {"setup": round(single_mem["setup_bytes"] / 1e9, 2),
                                    "workspace_high_water": round(single_mem["workspace"]["high_water_bytes"] / 1e9, 2),
                                    "workspace_reserved": round(single_mem["workspace"]["reserved_bytes"] / 1e9, 2)},
              "worlds": {}}
        import threading
        for w in [int(x) for x in args.replay_world.split(",") if x]:
            if w < 2 or args.fri_lde % w or args.cap % w:
                continue
            if stuck_legs:                        # a leg that did not come back may still hold the device: do not stack another on it
                sr["worlds"][str(w)] = {"error": "skipped: an earlier leg timed out"}
                continue
            try:
                # the leg runs W host threads that meet in barriers: it gets its own watchdog so that nothing it could do wrong
                # can cost the run its headline line (printed below in any case)
                box = {}

                def leg(w=w, box=box):
                    try:
                        torch.cuda.set_device(local_rank)
                        box["r"] = scale_replay.measure(circuit, w, args.fri_lde, args.cap, args.security, args.transcript,
                                                        steps=args.replay_steps, warmup=1, device=local_rank, reference_proof=proof_buf,
                                                        d_vars=d_vars, d_mult=d_mult, setup_cap=setup_cap)
                    except BaseException as e:    # noqa: BLE001
                        box["e"] = e

                th = threading.Thread(target=leg, daemon=True)
                th.start()
                th.join(float(os.environ.get("BJ_BENCH_REPLAY_TIMEOUT_S", str(240 << max(0, log_n - 22)))))
                if th.is_alive():
                    stuck_legs.append("scale_replay W=%d" % w)
                    raise TimeoutError("did not finish within the watchdog")
                if "e" in box:
                    raise box["e"]
                r = box["r"]
                r["hbm_per_rank_gb"] = {"setup": round(max(v["setup_bytes"] for v in r["ranks"].values()) / 1e9, 2),
                                        "workspace_high_water": round(max(v["workspace"]["high_water_bytes"] for v in r["ranks"].values()) / 1e9, 2),
                                        "workspace_reserved": round(max(v["workspace"]["reserved_bytes"] for v in r["ranks"].values()) / 1e9, 2)}
                for v in r["ranks"].values():
                    v.pop("setup_bytes", None)
                    v.pop("workspace", None)
                # T(W) = R + S / W and T(1) = R + S give the replicated part R the measurement implies
                r["implied_replicated_ms"] = round((w * r["max_ms"] - t1_ms) / (w - 1), 2)
                r["speedup_compute_only"] = round(t1_ms / r["max_ms"], 3)
                # the part no single GPU can measure, modelled: (W-1)/W of every gathered buffer arrives over ONE 150 GB/s link
                # (ring-bound all-gather) + 30 us per collective
                link_ms = r["mb_gathered_per_proof"] * (w - 1) / w / 150.0 + 0.03 * r["collectives_per_proof"]
                r["modelled_link_ms"] = round(link_ms, 2)
                r["ms_with_modelled_links"] = round(r["max_ms"] + link_ms, 2)
                r["speedup_with_modelled_links"] = round(t1_ms / (r["max_ms"] + link_ms), 3)
                sr["worlds"][str(w)] = r
            except Exception as e:                # noqa: BLE001 — the headline must not be lost to the secondary leg
                sr["worlds"][str(w)] = {"error": repr(e)[:300]}
                torch.cuda.empty_cache()
        out["scale_replay"] = sr
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as O
        from oracle import prover as OP
        affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        # the affinity mask can promise more cores than the container's CPU quota delivers (OpenMP then oversubscribes and spins):
        # calibrate on a small Poseidon2 tree and keep the thread count that hashes fastest
        quota = None                                  # cgroup CPU quota of this container, in cores (None = unlimited / unknown)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else round(float(q) / float(per), 2)
        except (OSError, ValueError):
            pass
        rng = np.random.default_rng(1)
        cal = rng.integers(0, E.P, size=(93, 1 << 13), dtype=np.uint64)
        best = (None, 0.0)
        # a burst as short as this calibration is not throttled by the quota, a 40 s proof is: over minutes, a team larger than twice
        # the quota only adds waiting (tools/oracle_threads_probe.py on the GPU box: 16 and 32 threads 54 s, 64 threads 59 s), so the
        # candidates stop there
        cap_threads = affinity if quota is None else max(4, min(affinity, int(2 * quota + 0.5)))
        for t_try in sorted({min(cap_threads, t) for t in (4, 8, 16, 32, 64, 128, 256)}):
            c0 = time.perf_counter()
            O.merkle_construct(cal, args.cap, threads=t_try)
            rate = (1 << 13) * 13 / (time.perf_counter() - c0)
            if rate > best[1] * 1.05:
                best = (t_try, rate)
        threads, perm_rate = best
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass

        def oracle_proof(lg):
            """One proof of the same circuit at 2^lg rows by the oracle prover: (seconds, busy cores)."""
            if args.circuit == "sha256" and lg >= 14:
                csmall = SHA.sha256_circuit(SHA.bench_message(SHA.message_len_for_log_n(lg), seed=42))
            else:
                csmall = S.sha_shaped_circuit(lg, seed=42, table_bits=4 if lg >= 14 else 2)
            osetup = OP.Setup(csmall, args.fri_lde, args.cap, threads=threads)
            cpu0 = sum(os.times()[:2])
            c0 = time.perf_counter()
            OP.prove(csmall, osetup, args.fri_lde, args.cap, security_level=args.security, threads=threads,
                     transcript_kind=transcript_kind if transcript_kind in (1, 2) else 1)
            t = time.perf_counter() - c0
            return t, (sum(os.times()[:2]) - cpu0) / t        # CPU seconds per wall second: the cores the proof actually kept busy

        # the bounded sample: 2^20 rows (BASELINE cfg3's size, a quarter of the bench's; the oracle proves it in ~40 s on 16 cores)
        # unless the calibration says this host would need minutes, then the 2^18 one alone
        est_main = 40.0 * (3.0e6 / max(perm_rate, 1.0)) * (2.0 ** (args.cpu_log_n - 20))
        # the full oracle prover keeps every LDE in host memory: ~60 GB at 2^20 rows (x4 per step of log n); leave a wide margin
        ram_gb = 0.0
        try:
            for line in open("/proc/meminfo"):
                if line.startswith("MemAvailable:"):
                    ram_gb = int(line.split()[1]) / 1e6
            for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
                try:
                    text = open(path).read().strip()
                    if text.isdigit():
                        ram_gb = min(ram_gb, int(text) / 1e9)
                except OSError:
                    pass
        except OSError:
            pass
        need_gb = 64.0 * (2.0 ** (args.cpu_log_n - 20))
        main_log = args.cpu_log_n if (est_main < 150.0 and ram_gb >= 2.5 * need_gb) else min(args.cpu_log_n, args.cpu_micro_log_n)
        t_cpu, busy_cores = oracle_proof(main_log)
        small = None
        if args.cpu_micro_log_n < main_log:
            small = (args.cpu_micro_log_n,) + oracle_proof(args.cpu_micro_log_n)
        # the oracle's C primitives at the bench's own sizes, all host cores (benches/benchmarks.rs:479-520 and :73-79)
        a = rng.integers(0, E.P, size=(256, 1 << 20), dtype=np.uint64)
        c0 = time.perf_counter()
        O.fft_batch(a, 7, threads=threads)
        t_ntt = time.perf_counter() - c0
        del a
        tl_log = 23                                   # the bench's own size when the host hashes it in ~10 s, else the largest that does
        while tl_log > 16 and (1 << tl_log) * 13 / perm_rate > 10.0:
            tl_log -= 1
        cols = rng.integers(0, E.P, size=(93, 1 << tl_log), dtype=np.uint64)
        c0 = time.perf_counter()
        O.merkle_construct(cols, args.cap, threads=threads)
        t_tree = time.perf_counter() - c0
        perms = (1 << tl_log) * 12 + (1 << tl_log) - args.cap
        del cols
        out["cpu_baseline"] = {"value": round((1 << main_log) / t_cpu, 1), "unit": "rows/s", "cores": threads, "kind": "port",
                               "log_n": main_log, "downgraded_from": (args.cpu_log_n if main_log != args.cpu_log_n else None),
                               "downgrade_reason": (None if main_log == args.cpu_log_n else
                                                    ("estimated %.0f s at 2^%d on this host" % (est_main, args.cpu_log_n) if est_main >= 150.0
                                                     else "%.0f GB of host memory available, %.0f GB wanted" % (ram_gb, 2.5 * need_gb))),
                               "cpu_model": cpu_model, "affinity_cpus": affinity, "cgroup_quota_cores": quota, "busy_cores_measured": round(busy_cores, 1),
                               "isa": "Poseidon2 trees and NTT butterflies: " + O.poseidon2_isa() + " (run-time selected; the reference's CPU path is AVX-512 MixedGL / "
                                      "state_avx512.rs); pointwise stages scalar",
                               "sample": "one proof of the same circuit at 2^%d rows by the oracle prover (C bulk ops + python "
                                         "orchestration, OpenMP, -O3 -march=x86-64-v3 + AVX-512 kernels where the CPU has them), %.1f s" % (main_log, t_cpu),
                               "micro": {**({"proof_2p%d" % small[0]: {"s": round(small[1], 2), "rows_per_s": round((1 << small[0]) / small[1], 1),
                                                                       "busy_cores_measured": round(small[2], 1),
                                                                       "what": "the same oracle proof at the size earlier rounds reported"}} if small else {}),
                                         "ntt_2p20_x256": {"ms": round(t_ntt * 1e3, 1), "GBps": round(16.0 * 256 * (1 << 20) / t_ntt / 1e9, 2),
                                                           "what": "oracle fft_natural_to_bitreversed, coset 7, one column per thread"},
                                         "poseidon2_tree_2p%d_x93" % tl_log: {"ms": round(t_tree * 1e3, 1), "Mperm_per_s": round(perms / t_tree / 1e6, 2),
                                                                             "what": "oracle MerkleTreeWithCap::construct, leaves then node layers"}}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if stuck_legs:           # a helper thread is still inside the library: no orderly teardown behind it
        print("bench.py: leaving through os._exit, stuck: %s" % ", ".join(stuck_legs), file=sys.stderr, flush=True)
        if sharded and rank == 0 and cache:
            import shutil
            shutil.rmtree(cache, ignore_errors=True)
        os._exit(0)
    if locals().get("ran_other_bulk"):
        # the extra proofs on the other bulk transport may have ended differently on different ranks (a watchdog on one, not on another):
        # no collective may follow them — every rank leaves on its own, the line is out
        if sharded and rank == 0 and cache:      # the mapped copy of the circuit (the other ranks' mappings outlive the unlink)
            import shutil
            shutil.rmtree(cache, ignore_errors=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    barrier()           # rank 0 may still have been verifying / timing the CPU baseline: leave the group together
    setup.close()
    if sharded and rank == 0 and cache:      # the mapped copy of the circuit (the other ranks' mappings outlive the unlink)
        import shutil
        shutil.rmtree(cache, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
