"""The sharded-quotient model of DESIGN.md §6 (tests/sharding_model.py): a polynomial of degree < q*n is recovered from
per-coset residues — each computable by the rank that holds the coset(s) — by a q x q Vandermonde solve per coefficient.
Checked with the oracle's transforms on the LDE's own coset enumeration, for one coset per rank (W = q) and for pairs of
adjacent cosets per rank (W = q / 2 ranks of two cosets each = cosets of H_{2n})."""
import numpy as np
import pytest

import oracle as O
from tests import sharding_model as M

P = M.P


def _evaluate_on_lde(T_mono, log_n, log_lde):
    """[L][n] values of T (any degree < L*n) on the LDE cosets, each coset bit-reversed: the layout of every LDE buffer."""
    n, L = 1 << log_n, 1 << log_lde
    out = np.zeros((L, n), dtype=np.uint64)
    coeffs = [int(v) for v in T_mono]
    for c in range(L):
        s = M.lde_coset_shift(log_n, log_lde, c)
        # T(s * w^i) for i < n: fold T modulo x^n - s^n first (exactly what the residue is), then one size-n transform
        a = pow(s, n, P)
        folded = np.zeros(n, dtype=np.uint64)
        for k in range(n):
            acc, ap = 0, 1
            for j in range((len(coeffs) + n - 1) // n):
                idx = j * n + k
                if idx < len(coeffs):
                    acc = (acc + ap * coeffs[idx]) % P
                ap = ap * a % P
            folded[k] = acc
        out[c] = O.fft_natural_to_bitreversed(folded, s)
    return out


@pytest.mark.parametrize("log_n,log_q", [(4, 2), (6, 2), (5, 3)])
def test_one_coset_per_rank_recovers_the_quotient(log_n, log_q):
    n, q, log_lde = 1 << log_n, 1 << log_q, 3
    rng = np.random.default_rng(100 * log_n + log_q)
    T = rng.integers(0, P, size=q * n, dtype=np.uint64)
    T[-1] = 0                                      # degree < q*n - 1 as for a real quotient; irrelevant to the method
    ev = _evaluate_on_lde(T, log_n, log_lde)
    # spot check of the layout against a direct evaluation
    s3 = M.lde_coset_shift(log_n, log_lde, 3)
    x = s3 * pow(O.omega(log_n), O.bitrev(5, log_n), P) % P
    assert int(ev[3][5]) == sum(int(c) * pow(x, k, P) for k, c in enumerate(T)) % P
    for cosets in (list(range(q)), list(range(1 << log_lde))[-q:], [0, 2, 5, 7][:q] if q == 4 else list(range(0, 2 * q, 2))[:q]):
        shifts = [M.lde_coset_shift(log_n, log_lde, c) for c in cosets]
        a = [pow(s, n, P) for s in shifts]
        if len(set(a)) < q:
            continue                               # cosets with equal x^n cannot be combined (never the case for the first q)
        res = [M.residue_from_coset(ev[c], s, O.ifft_natural_to_natural, O.bitreverse) for c, s in zip(cosets, shifts)]
        # the residue is what the formula says
        for r, ai in zip(res, a):
            want = [sum(pow(ai, j, P) * int(T[j * n + k]) for j in range(q)) % P for k in range(n)]
            assert [int(v) for v in r] == want
        assert np.array_equal(M.combine_residues(res, a), T)


def test_two_cosets_per_rank_recover_the_quotient():
    """W = 2 at quotient degree 4: rank 0 evaluates its cosets {0, 1}, rank 1 its cosets {4, 5}; each pair is one coset of
    H_{2n} in the bit-reversed enumeration, the two residues modulo x^(2n) - a_i give T."""
    log_n, log_lde, q = 5, 3, 4
    n = 1 << log_n
    rng = np.random.default_rng(7)
    T = rng.integers(0, P, size=q * n, dtype=np.uint64)
    ev = _evaluate_on_lde(T, log_n, log_lde)
    res, a = [], []
    for first in (0, 4):
        # cosets (first, first + 1) interleave into the coset shift(first) * H_{2n}, bit-reversed over 2n points: point
        # I = c*n + i of the pair is the bit-reversed index of the size-2n enumeration (the LDE's "first d cosets" property)
        pair = np.concatenate([ev[first], ev[first + 1]])
        s = M.lde_coset_shift(log_n, log_lde, first)
        res.append(M.residue_from_coset(pair, s, O.ifft_natural_to_natural, O.bitreverse))
        a.append(pow(s, 2 * n, P))
    got = M.combine_residues(res, a)
    assert np.array_equal(got, T)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_even_split_over_all_ranks_recovers_the_quotient(world):
    """What the sharded prover does since round 4 (csrc/prover.hip, bj_combine_residues): EVERY rank evaluates the first
    E = q n / W points of its own coset range — two cosets (W = 2), one (W = 4) or the first HALF of one (W = 8: the points of
    the coset shift * H_{n/2}, the even powers in the bit-reversed enumeration) — and the W residues modulo x^E - shift^E give T."""
    log_n, log_lde, q = 5, 3, 4
    n, L = 1 << log_n, 1 << log_lde
    rng = np.random.default_rng(70 + world)
    T = rng.integers(0, P, size=q * n, dtype=np.uint64)
    ev = _evaluate_on_lde(T, log_n, log_lde).reshape(-1)          # flat LDE: index coset * n + i
    E, per_rank = q * n // world, L * n // world
    res, a = [], []
    for r in range(world):
        first_coset = r * (L // world)
        s = M.lde_coset_shift(log_n, log_lde, first_coset)
        pts = ev[r * per_rank:r * per_rank + E]
        res.append(M.residue_from_coset(pts, s, O.ifft_natural_to_natural, O.bitreverse))
        a.append(pow(s, E, P))
    assert len(set(a)) == world
    assert np.array_equal(M.combine_residues(res, a), T)
