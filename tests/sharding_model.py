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


# ---------------------------------------------------------------------------------------------------------------------------
# Quotient evaluation sharded over ALL ranks ("costed, not built"): the model the device code will follow.
#
# T has degree < q*n.  Its evaluations on ONE coset s*H_n of the LDE domain (what a rank holding that coset can compute from
# its own columns) determine R = T mod (x^n - a), a = s^n, by one size-n inverse transform with shift s: writing
# T = sum_j x^(j n) T_j (T_j of degree < n), R = sum_j a^j T_j.  Any q residues with distinct a_i give T_0 .. T_{q-1} back
# through the inverse of the Vandermonde matrix V[i][j] = a_i^j, coefficient by coefficient.  A rank holding m adjacent cosets
# (m a power of two, the first of them at a multiple of m) holds one coset of H_{m n} and gets a residue modulo x^(m n) - a the
# same way, so W ranks can split the q*n evaluation points evenly whenever W <= q * (cosets per rank).
# ---------------------------------------------------------------------------------------------------------------------------
P = (1 << 64) - (1 << 32) + 1


def _omega(log_n):
    w = 0x185629DCDA58878C
    for _ in range(32 - log_n):
        w = w * w % P
    return w


def _bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def lde_coset_shift(log_n, log_lde, coset):
    """x-shift of LDE coset `coset` (bit-reversed coset enumeration: 7 * w_{nL}^bitrev_L(coset), fft/mod.rs coset LDE)."""
    return 7 * pow(_omega(log_n + log_lde), _bitrev(coset, log_lde), P) % P


def residue_from_coset(evals_bitrev, shift, ifft_natural_to_natural, bitreverse):
    """T mod (x^m - shift^m) from T's values on shift*H_m in bit-reversed order (m = len): the inverse transform with that shift."""
    return ifft_natural_to_natural(bitreverse(evals_bitrev), shift)


def combine_residues(residues, a_values):
    """residues[i] = T mod (x^m - a_i) (arrays of m coefficients, python ints or u64) -> the q*m coefficients of T."""
    q = len(residues)
    m = len(residues[0])
    # inverse of V[i][j] = a_i^j over F_p by Gauss-Jordan (q is 2, 4 or 8)
    V = [[pow(a % P, j, P) for j in range(q)] for a in a_values]
    A = [row[:] + [1 if i == k else 0 for k in range(q)] for i, row in enumerate(V)]
    for c in range(q):
        piv = next(r for r in range(c, q) if A[r][c])
        A[c], A[piv] = A[piv], A[c]
        inv = pow(A[c][c], P - 2, P)
        A[c] = [v * inv % P for v in A[c]]
        for r in range(q):
            if r != c and A[r][c]:
                f = A[r][c]
                A[r] = [(v - f * w) % P for v, w in zip(A[r], A[c])]
    Vinv = [row[q:] for row in A]
    out = np.zeros(q * m, dtype=np.uint64)
    res = [[int(v) % P for v in r] for r in residues]
    for j in range(q):
        for k in range(m):
            out[j * m + k] = sum(Vinv[j][i] * res[i][k] for i in range(q)) % P
    return out


def sharded_quotient_monomials(evals_local_bitrev, log_n, log_lde, q, world, rank, ifft_natural_to_natural, bitreverse, device=None):
    """What prove_impl does for the quotient on W ranks (csrc/prover.hip, round 4): `evals_local_bitrev` = this rank's values of T on
    the first E = q n / W points of ITS OWN coset range (one array of E words per component), inverse-transformed to the residue
    T mod (x^E - a_rank), all-gathered (torch.distributed), combined.  Returns the q n coefficients of T per component."""
    import torch
    import torch.distributed as dist
    n, L = 1 << log_n, 1 << log_lde
    E = q * n // world
    shift = lde_coset_shift(log_n, log_lde, rank * (L // world))
    mine = np.stack([np.asarray(residue_from_coset(col, shift, ifft_natural_to_natural, bitreverse), dtype=np.uint64)
                     for col in evals_local_bitrev])
    t = torch.from_numpy(mine.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    res = [o.cpu().numpy().view(np.uint64) for o in out]                       # [world][components][E]
    a = [pow(lde_coset_shift(log_n, log_lde, r * (L // world)), E, P) for r in range(world)]
    return np.stack([combine_residues([res[r][comp] for r in range(world)], a) for comp in range(mine.shape[0])])
