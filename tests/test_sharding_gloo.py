"""world_size-2 (and 4) CPU tests of the N>1 path over the gloo backend: partition logic + the cap all-gather of
tests/sharding_model.py.  The GPU compute is replaced by an oracle-backed stand-in *inside this test* so that the
collective/partition code that runs on 8 GPUs (backend = era_boojum_amd.Context, RCCL) is exactly what runs here."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    """Stand-in for era_boojum_amd.Context on CPU: 'device pointers' are keys into a dict of numpy arrays."""

    def __init__(self):
        import oracle
        self.O = oracle
        self.mem, self.next = {}, 1

    def malloc(self, nbytes):
        k = self.next
        self.next += 1
        self.mem[k] = np.zeros(nbytes // 8, dtype=np.uint64)
        return k

    def upload(self, arr):
        k = self.malloc(arr.size * 8)
        self.mem[k][:] = arr.reshape(-1)
        return k

    def lde_cosets_batch(self, d_mono, d_out, log_n, n_cols, log_lde, c0, cnt):
        mono = self.mem[d_mono].reshape(n_cols, 1 << log_n)
        full = self.O.lde_batch(mono, log_lde)
        self.mem[d_out][:] = full[:, c0:c0 + cnt, :].reshape(-1)

    def merkle_tree_build(self, d_cols, stride, n_cols, leaves, cap, d_tree):
        cols = self.mem[d_cols].reshape(n_cols, stride)[:, :leaves]
        self.mem[d_tree][:] = self.O.merkle_construct(np.ascontiguousarray(cols), cap).reshape(-1)

    def merkle_tree_cap(self, d_tree, leaves, cap):
        tree = self.mem[d_tree].reshape(-1, 4)
        return self.O.merkle_cap(tree, leaves, cap)


def _worker(rank, world, port, log_n, n_cols, log_lde, cap, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import sharding_model as sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(123)             # same monomials on every rank (replicated input)
        P = (1 << 64) - (1 << 32) + 1
        mono = rng.integers(0, P, size=(n_cols, 1 << log_n), dtype=np.uint64)
        be = OracleBackend()
        d_mono = be.upload(mono)
        d_lde, d_tree, local_leaves, cap_full = sharding.sharded_commit(be, d_mono, log_n, n_cols, log_lde, cap, world, rank)
        # column sharding: every rank transforms its own column range; gather and compare on rank 0
        start, cnt = sharding.column_shard(n_cols, world, rank)
        mine = be.O.fft_batch(mono[start:start + cnt], 7) if cnt else np.zeros((0, 1 << log_n), dtype=np.uint64)
        gathered = [None] * world
        dist.all_gather_object(gathered, (start, cnt, mine))
        q.put((rank, cap_full, local_leaves, gathered if rank == 0 else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_coset_sharded_commit_and_column_sharding(world):
    import oracle as O
    log_n, n_cols, log_lde, cap = 6, 5, 3, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, log_n, n_cols, log_lde, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(123)
    mono = rng.integers(0, O.P, size=(n_cols, 1 << log_n), dtype=np.uint64)
    lde = O.lde_batch(mono, log_lde)
    N = (1 << log_n) << log_lde
    want_cap = O.merkle_cap(O.merkle_construct(lde.reshape(n_cols, N), cap), N, cap)
    for rank, cap_full, local_leaves, gathered in results:
        assert np.array_equal(cap_full, want_cap), "rank %d assembled a different cap" % rank
        assert local_leaves == N // world
        if gathered is not None:
            cols = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])], axis=0)
            assert np.array_equal(cols, O.fft_batch(mono, 7))


def _quotient_worker(rank, world, port, log_n, log_lde, quot_deg, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import oracle as O
    import sharding_model as sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, L = 1 << log_n, 1 << log_lde
        rng = np.random.default_rng(321)                   # the same T on every rank: each evaluates ITS points of it
        T = rng.integers(0, O.P, size=(2, quot_deg * n), dtype=np.uint64)
        E, per_rank = quot_deg * n // world, L * n // world
        local = []
        for comp in range(2):
            # T on this rank's first E points: its cosets of the size-L*n domain are extensions of the monomials T (zero-padded)
            mono = np.zeros(L * n, dtype=np.uint64)
            mono[:quot_deg * n] = T[comp]
            full = O.fft_natural_to_bitreversed(mono, 7)   # bit-reversed enumeration of 7 * <w_{Ln}>: flat index = coset * n + i
            local.append(full[rank * per_rank: rank * per_rank + E])
        got = sharding.sharded_quotient_monomials(local, log_n, log_lde, quot_deg, world, rank, O.ifft_natural_to_natural, O.bitreverse)
        q.put((rank, np.array_equal(got, T)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_quotient_residues_over_gloo(world):
    """The quotient's exchange step on W processes (csrc/prover.hip since round 4): every rank evaluates q n / W points of its own
    cosets (here: reads them off a size-L*n transform of a random T), inverse-transforms its piece, the residues are
    all-gathered over gloo and combined — every rank ends with the coefficients of T."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_quotient_worker, args=(r, world, port, 5, 3, 4, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in results) == list(range(world)) and all(ok for _, ok in results)


def test_partition_helpers():
    import sharding_model as S
    for n_cols in (0, 1, 7, 93, 256):
        for world in (1, 2, 3, 8):
            parts = [S.column_shard(n_cols, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n_cols
            for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    assert [S.coset_shard(8, 4, r) for r in range(4)] == [(0, 2), (2, 2), (4, 2), (6, 2)]
    assert S.cap_fragment_size(16, 8) == 2
    with pytest.raises(ValueError):
        S.coset_shard(8, 3, 0)
    with pytest.raises(ValueError):
        S.cap_fragment_size(4, 8)
