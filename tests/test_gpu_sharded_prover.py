"""-m gpu: ONE proof with its LDE cosets split across `world` ranks (bj_setup_create_sharded) must be,
on every rank, bit for bit the proof a single GPU produces (which test_gpu_prover.py pins to the oracle prover).
The ranks are separate processes launched with torch.distributed.run; on a one-GPU box they share the GPU and talk
over gloo — the sharding logic (coset ranges, subtree caps, quotient gather, FRI layer gather, query ownership) is
exactly what runs over RCCL with one GPU per rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import proof_format, synthetic as S
from gpu_util import ctx
from oracle import verifier as OV

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (world, log_n, fri_lde, cap): 8/4-way = fewer cosets per rank than the quotient degree (gathered quotient),
# 2-way at LDE 8 = every rank evaluates the quotient on its own four cosets, 2-way at LDE 4 = two cosets per rank
# log_n = 0: the real SHA-256 circuit (2^14 rows) on 8 ranks, the bench's multi-GPU configuration in small
# log_n < 0: the recursion-class circuit (quotient degree 8: a rank's quotient piece is two cosets at 4 ranks, one at 8) at 2^-log_n rows
# log_n = 12: the bench geometry + gates over specialized columns, one of them with its own constant columns
# log_n = 13: lookups with the table id as a VARIABLE column (UseSpecializedColumnsWithTableIdAsVariable) + a gate over specialized columns
@pytest.mark.parametrize("world,log_n,fri_lde,cap,sec", [(2, 10, 8, 16, 30), (4, 10, 8, 16, 30), (8, 11, 8, 16, 40), (2, 9, 4, 8, 20),
                                                         (8, 0, 8, 16, 30), (4, -10, 8, 16, 30), (8, -10, 8, 16, 30), (4, 12, 8, 16, 30), (4, 13, 8, 16, 30)])
def test_sharded_proof_equals_single_gpu_proof(tmp_path, world, log_n, fri_lde, cap, sec):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "sharded_worker.py"), str(tmp_path),
           str(log_n), str(fri_lde), str(cap), str(sec)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if log_n == 0:
        from era_boojum_amd import sha256_circuit as SHA
        c = SHA.sha256_circuit(SHA.bench_message(100, seed=7))
    elif log_n < 0:
        c = S.recursion_like_circuit(-log_n, seed=7)
    else:
        c = S.sha_shaped_circuit(log_n, seed=7, table_bits=4 if log_n >= 14 else 2,
                                 **({"boolean_columns": 2, "specialized_constant_columns": 3} if log_n == 12 else
                                    {"table_id_as_variable": True, "boolean_columns": 2} if log_n == 13 else {}))
    single = E.ProverSetup(ctx(), c, fri_lde, cap, sec)
    ref, _ = single.prove()
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "proof_%d.npy" % rank))
        assert np.array_equal(got, ref), "rank %d: sharded proof differs from the single-GPU proof" % rank
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "cap_%d.npy" % rank)), single.cap())
    pg = proof_format.parse(ref, security_level=sec)
    assert OV.verify(OV.VerificationKey(c, single.cap(), fri_lde, cap), pg)
    single.close()
    calls, nbytes = map(int, open(os.path.join(str(tmp_path), "comm.txt")).read().split())
    assert calls >= 5 and nbytes > 0      # setup cap + 3 oracle caps + FRI cap/layer + queries


@pytest.mark.parametrize("world,log_n,fri_lde,cap,sec,bulk_min", [(2, 12, 8, 16, 30, 4096), (4, 13, 8, 16, 30, 4096), (8, 11, 8, 16, 40, 1024),
                                                                  (8, 0, 8, 16, 30, 1 << 20)])
def test_sharded_proof_with_full_mesh_peer_copies(tmp_path, world, log_n, fri_lde, cap, sec, bulk_min):
    """bj_comm_peer_create (csrc/comm_peer.hip, SURVEY §8e "direct full-mesh peer copies"): the ranks are processes sharing this
    GPU; every exchange of at least bulk_min bytes per rank is served by each rank copying its contribution straight into the
    receive buffers of its peers, mapped through HIP IPC (handles and the completion barrier travel over a gloo control group),
    the smaller ones by the base transport.  Every rank's proof is the single-GPU proof, bulk exchanges were really served by
    peer copies (no fallback), and a second proof reuses the mappings.  The last case keeps the default threshold (1 MiB): only
    the real bulk exchanges of the SHA-256 circuit at 8 ranks qualify."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", BJ_COMM_BULK="peer", BJ_COMM_BULK_MIN=str(bulk_min))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "sharded_worker.py"), str(tmp_path),
           str(log_n), str(fri_lde), str(cap), str(sec)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    if log_n == 0:
        from era_boojum_amd import sha256_circuit as SHA
        c = SHA.sha256_circuit(SHA.bench_message(100, seed=7))
    else:
        c = S.sha_shaped_circuit(log_n, seed=7, table_bits=4 if log_n >= 14 else 2,
                                 **({"boolean_columns": 2, "specialized_constant_columns": 3} if log_n == 12 else
                                    {"table_id_as_variable": True, "boolean_columns": 2} if log_n == 13 else {}))
    single = E.ProverSetup(ctx(), c, fri_lde, cap, sec)
    ref, _ = single.prove()
    single.close()
    for rank in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "proof_%d.npy" % rank)), ref), "rank %d" % rank
        st = json.load(open(os.path.join(str(tmp_path), "peer_%d.json" % rank)))
        assert st["fallbacks"] == 0 and st["small_calls"] > 0, st
        if bulk_min < (1 << 20):
            assert st["bulk_calls"] >= 2 and st["bulk_bytes_received"] >= 2 * world * bulk_min, st


def test_peer_transport_falls_back_together_when_one_rank_cannot_export(tmp_path):
    """One rank of four reports that it could not export its mailbox (BJ_PEER_TEST_FAIL_RANK, a test hook in comm_peer.hip): ALL
    ranks serve that exchange size through the base transport from then on — the decision travels in the negotiation, nobody is
    left waiting in a copy — and the proofs are still the single-GPU bytes."""
    import json
    world, log_n, fri_lde, cap, sec = 4, 12, 8, 16, 30
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", BJ_COMM_BULK="peer", BJ_COMM_BULK_MIN="4096",
               BJ_PEER_TEST_FAIL_RANK="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "sharded_worker.py"), str(tmp_path),
           str(log_n), str(fri_lde), str(cap), str(sec)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    c = S.sha_shaped_circuit(log_n, seed=7, table_bits=2, boolean_columns=2, specialized_constant_columns=3)
    single = E.ProverSetup(ctx(), c, fri_lde, cap, sec)
    ref, _ = single.prove()
    single.close()
    for rank in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "proof_%d.npy" % rank)), ref), "rank %d" % rank
        st = json.load(open(os.path.join(str(tmp_path), "peer_%d.json" % rank)))
        assert st["fallbacks"] >= 2 and st["bulk_calls"] == 0, st


def test_sharded_setup_rejects_bad_world():
    c = S.sha_shaped_circuit(8, seed=1, table_bits=2)

    class FakeComm:  # never called: argument validation comes first
        def __init__(self, rank, world):
            from era_boojum_amd.binding import _ALL_GATHER_FN, _Comm
            self._fn = _ALL_GATHER_FN(lambda *a: 1)
            self.struct = _Comm(rank, world, self._fn, None)

    for rank, world, fri, cap in [(0, 3, 8, 16), (0, 16, 8, 16), (4, 4, 8, 16), (0, 8, 8, 4)]:
        with pytest.raises(E.BoojumHipError):
            E.ProverSetup(ctx(), c, fri, cap, 20, comm=FakeComm(rank, world))


def test_torchcomm_all_gather_over_rccl_world_1():
    """The production transport of bj_comm: torch.distributed backend nccl (= RCCL) on device staging tensors.  A one-GPU box
    can only form a world of 1, which still exercises the whole TorchComm path (D2D staging, all_gather_into_tensor on uint8,
    synchronisation, copy back)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        comm = E.TorchComm(ctx())
        assert comm._on_device and (comm.rank, comm.world) == (0, 1)
        src = torch.arange(1 << 18, dtype=torch.int64, device="cuda") * 3 + 1
        dst = torch.zeros_like(src)
        assert comm._all_gather(None, src.data_ptr(), dst.data_ptr(), src.numel() * 8) == 0
        torch.cuda.synchronize()
        assert torch.equal(src, dst) and comm.calls == 1
    finally:
        dist.destroy_process_group()


def test_in_library_rccl_transport_world_1():
    """bj_comm_rccl_create: the library's own transport (ncclAllGather on the context's stream, no staging, no host trampoline).
    One GPU can only form a world of 1 (RCCL refuses two ranks on one device); that still loads librccl, creates the
    communicator from a unique id, and runs both entry points of the bj_comm it fills on device buffers."""
    import ctypes as C
    from gpu_util import DevBuf
    uid = E.binding.rccl_unique_id()
    assert len(uid) == E.binding.RCCL_UNIQUE_ID_BYTES
    comm = E.RcclComm(ctx(), uid, 0, 1)
    assert (comm.struct.rank, comm.struct.world) == (0, 1) and comm.struct.all_gather_stream and comm.struct.all_gather
    src = np.arange(1 << 16, dtype=np.uint64) * np.uint64(5) + np.uint64(3)
    d_s, d_r = DevBuf(src), DevBuf(nelems=src.size)
    assert comm.struct.all_gather_stream(comm.struct.user, C.c_void_p(d_s.ptr), C.c_void_p(d_r.ptr), src.nbytes, None) == 0
    ctx().sync()
    assert np.array_equal(d_r.get(), src)
    d_r2 = DevBuf(nelems=src.size)
    assert comm.struct.all_gather(comm.struct.user, C.c_void_p(d_s.ptr), C.c_void_p(d_r2.ptr), src.nbytes) == 0
    assert np.array_equal(d_r2.get(), src)
    assert comm.calls == 2 and comm.bytes == 2 * src.nbytes
    d_r3 = DevBuf(nelems=src.size)          # the same through the binding's helper (bench.py's transport self-test)
    comm.all_gather(d_s.ptr, d_r3.ptr, src.nbytes, stream=0)
    ctx().sync()
    assert np.array_equal(d_r3.get(), src)
    d_r3.free()
    # a setup created through the sharded entry point with this communicator proves like the plain one
    c = S.sha_shaped_circuit(9, seed=2, table_bits=2)
    a = E.ProverSetup(ctx(), c, 8, 16, 20, comm=comm)
    b = E.ProverSetup(ctx(), c, 8, 16, 20)
    pa, _ = a.prove()
    pb, _ = b.prove()
    assert np.array_equal(pa, pb)
    a.close(); b.close()
    comm.close()
    for d in (d_s, d_r, d_r2):
        d.free()


def test_plain_bench_command_with_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2 …` with NO launcher around it (the shape of the driver's N = 1 command) starts two ranks by
    itself: the JSON line reports the world size of the process group, the sharded proof went through collectives and rank 0's
    verifier restatement accepted it.  gloo backend: the two ranks share this box's one GPU (functional check)."""
    import json
    env = dict(os.environ, BJ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--log-n", "16", "--steps", "2",
           "--warmup", "1", "--no-ntt", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["comm"]["world"] == 2 and out["comm"]["calls_per_proof"] > 0 and out["comm"]["mb_received_per_rank_per_proof"] > 0
    assert "accepts" in out["config"]["verified"]
    # first-contact diagnostics of a multi-rank run: which rank set the time, the spread per stage, where the circuit came from
    rk = out["ranks"]
    assert rk["slowest_rank"] in (0, 1) and rk["ms_per_step_min"] <= rk["ms_per_step_max"] <= out["ms_per_step"] * 1.001
    assert set(rk["stages_ms_min"]) == set(rk["stages_ms_max"]) == set(out["stages_ms"])
    assert any("mapped from rank 0" in c for c in rk["circuit_source"]) and any("synthesised" in c for c in rk["circuit_source"])


def test_plain_bench_command_refuses_more_gpus_than_visible():
    """Without the functional gloo mode, asking for more devices than the box has is an error — never a silent single-GPU run
    that prints n_gpus 1."""
    import torch
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "BJ_BENCH_BACKEND"):
        env.pop(k, None)
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", str(want), "--log-n", "14",
                        "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


@pytest.mark.parametrize("world,log_n", [(2, 11), (4, 12), (8, 12), (8, 18), (8, 22)])
def test_one_rank_alone_with_recorded_peers(world, log_n):
    """bench.py's scale_replay leg in small (era_boojum_amd/scale_replay.py): the gathered buffer of every collective is recorded
    with ONE rank on the device at a time (a replay transport that serves collectives 0 .. k-1 and captures the rank's contribution
    to collective k: bj_comm_replay_capture), the last sweep ends with every rank holding the single-GPU proof; then every rank
    runs ALONE behind the recorded-peer transport (bj_comm_replay_create, installed with bj_setup_set_comm): each of its
    contributions equals the slice the recording holds for it, and its proof is again the single-GPU proof, byte for byte.  The
    rank's HBM (setup + workspace high-water mark) comes back too; no proof needed an overflow slab.  The last case is the bench's
    size: eight ranks of a 2^22-row proof, each with its monomials in the tiled layout, its single coset extended by the two-pass
    plan, DEEP numerator slices combined on tiled coefficients — every rank's proof = the single-GPU bytes."""
    from era_boojum_amd import scale_replay
    c = S.sha_shaped_circuit(log_n, seed=23, table_bits=2)
    single = E.ProverSetup(ctx(), c, 8, 16, 30)
    ref, _ = single.prove()
    single.close()
    got = scale_replay.measure(c, world, 8, 16, 30, steps=2, warmup=1, reference_proof=ref)
    assert got["world"] == world and sorted(got["ranks"]) == list(range(world))
    assert got["collectives_per_proof"] >= 5 and got["mb_gathered_per_proof"] > 0
    assert all(v["ms_per_step"] > 0 and "fri" in v["stages_ms"] for v in got["ranks"].values())
    assert got["max_ms"] == got["ranks"][got["slowest_rank"]]["ms_per_step"] >= got["min_ms"]
    for v in got["ranks"].values():
        assert v["setup_bytes"] > 0 and v["workspace"]["overflow_slabs"] == 0
        assert 0 < v["workspace"]["high_water_bytes"] <= v["workspace"]["reserved_bytes"]
    ctx().release_workspace()


def test_replay_transport_refuses_what_was_not_recorded():
    """The recorded-peer transport answers only the collectives it holds: another size is an error of the proof, not a wrong
    answer; a sharded setup takes another transport only for its own rank and world."""
    import ctypes as C
    from gpu_util import DevBuf
    blob = DevBuf(np.arange(64, dtype=np.uint64))
    rc = E.ReplayComm(ctx(), 1, 2, [(blob.ptr, 512)], verify=True)
    src, dst = DevBuf(np.arange(32, 64, dtype=np.uint64)), DevBuf(nelems=64)
    assert rc.struct.all_gather(rc.struct.user, C.c_void_p(src.ptr), C.c_void_p(dst.ptr), 256) == 0
    assert np.array_equal(dst.get(), np.arange(64, dtype=np.uint64)) and rc.stats() == (1, 512, 0)
    other = DevBuf(np.zeros(32, dtype=np.uint64))        # not what rank 1 contributed in the recording
    assert rc.struct.all_gather(rc.struct.user, C.c_void_p(other.ptr), C.c_void_p(dst.ptr), 256) == 0
    assert rc.stats() == (2, 1024, 1)
    assert rc.struct.all_gather(rc.struct.user, C.c_void_p(src.ptr), C.c_void_p(dst.ptr), 128) != 0      # a size nobody recorded
    with pytest.raises(E.BoojumHipError):
        E.ReplayComm(ctx(), 2, 2, [(blob.ptr, 512)])
    with pytest.raises(E.BoojumHipError):
        E.ReplayComm(ctx(), 0, 2, [(blob.ptr, 511)])
    single = E.ProverSetup(ctx(), S.sha_shaped_circuit(8, seed=1, table_bits=2), 8, 16, 20)
    with pytest.raises(E.BoojumHipError):
        single.set_comm(rc)                                # not a sharded setup
    single.close()
    rc.close()
    for d in (blob, src, dst, other):
        d.free()
