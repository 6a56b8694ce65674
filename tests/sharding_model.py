"""Multi-GPU partitioning of the hot path (one process per GPU, torch.distributed over RCCL on GPUs / gloo on CPU).

Two axes:
  * independent polynomial columns (NTT/LDE batches): contiguous balanced column ranges per rank, no collective;
  * LDE cosets for commitments: rank r owns cosets [r*L/W, (r+1)*L/W) of EVERY column.  Because the Merkle leaf index is
    coset*n + i (src/cs/implementations/proof.rs:89-91) that is a contiguous leaf range, i.e. a complete subtree whose
    root layer is a contiguous fragment (cap_size/W nodes) of the cap.  One tiny all-gather (cap_size * 32 bytes in
    total) assembles the cap for the host transcript — the only collective of a commit round.
The compute is injected as a `backend` (era_boojum_amd.Context on a GPU); nothing here falls back to CPU math."""
import numpy as np


def column_shard(n_cols, world, rank):
    """Contiguous balanced range [start, start+count) of columns for `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n_cols, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def coset_shard(lde_factor, world, rank):
    """Contiguous range of LDE cosets owned by `rank`; requires world | lde_factor (both powers of two)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    if lde_factor % world != 0:
        raise ValueError("world size %d must divide the LDE factor %d" % (world, lde_factor))
    per = lde_factor // world
    return rank * per, per


def cap_fragment_size(cap_size, world):
    if cap_size % world != 0:
        raise ValueError("merkle_tree_cap_size %d must be a multiple of the world size %d" % (cap_size, world))
    return cap_size // world


def all_gather_cap(local_fragment, world, device=None):
    """All-gather the per-rank cap fragments ([cap/W, 4] u64) into the full cap ([cap, 4]) on every rank."""
    frag = np.ascontiguousarray(local_fragment, dtype=np.uint64)
    if world == 1:
        return frag.copy()
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(frag.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([o.cpu().numpy().view(np.uint64) for o in out], axis=0)


def sharded_commit(backend, d_mono, log_n, n_cols, log_lde, cap_size, world, rank, device=None):
    """One commit round on this rank's coset range: LDE of the owned cosets, Poseidon2 subtree, cap all-gather.

    backend must provide lde_cosets_batch / merkle_tree_build / merkle_tree_cap / malloc (era_boojum_amd.Context).
    Returns (d_lde, d_tree, local_leaves, cap) where cap is the FULL cap, identical on every rank."""
    n, L = 1 << log_n, 1 << log_lde
    c0, cnt = coset_shard(L, world, rank)
    frag = cap_fragment_size(cap_size, world)
    local_leaves = cnt * n
    if frag > local_leaves:
        raise ValueError("cap fragment larger than the local subtree")
    d_lde = backend.malloc(8 * n_cols * local_leaves)
    backend.lde_cosets_batch(d_mono, d_lde, log_n, n_cols, log_lde, c0, cnt)
    nd = 2 * local_leaves - frag
    d_tree = backend.malloc(32 * nd)
    backend.merkle_tree_build(d_lde, local_leaves, n_cols, local_leaves, frag, d_tree)
    local_cap = backend.merkle_tree_cap(d_tree, local_leaves, frag)
    cap = all_gather_cap(local_cap, world, device)
    return d_lde, d_tree, local_leaves, cap
