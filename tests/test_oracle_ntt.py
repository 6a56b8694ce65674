"""Differential properties of the oracle NTT/LDE/FRI, mirroring the reference's own tests:
NTT == O(n^2) evaluation (fft/mod.rs:1345-1384, coset :1592-1634), iNTT∘NTT = id incl. coset 7 (:1540-1589),
LDE layout (utils.rs:311-403, proof.rs:89-91) and FRI fold-by-value consistency (fri/mod.rs:961-1031)."""
import numpy as np
import pytest

import oracle as O

P = O.P


def rand_gl(rng, shape, noncanonical=False):
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    if noncanonical:  # the reference tolerates any u64 in memory (goldilocks/mod.rs:98-107)
        mask = rng.random(size=shape) < 0.05
        a = np.where(mask, np.uint64(P) + rng.integers(0, (1 << 32) - 1, size=shape, dtype=np.uint64), a)
    return a


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8])
@pytest.mark.parametrize("coset", [1, 7])
def test_ntt_equals_naive_dft(log_n, coset):
    rng = np.random.default_rng(100 + log_n)
    a = rand_gl(rng, 1 << log_n, noncanonical=True)
    got = O.fft_natural_to_bitreversed(a, coset)
    want = O.bitreverse(O.naive_dft(a, coset))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("log_n", [1, 4, 10, 14])
@pytest.mark.parametrize("coset", [1, 7, 0x1234567])
def test_intt_roundtrip(log_n, coset):
    rng = np.random.default_rng(7 + log_n)
    a = rand_gl(rng, 1 << log_n)
    ev = O.bitreverse(O.fft_natural_to_bitreversed(a, coset))
    assert np.array_equal(O.ifft_natural_to_natural(ev, coset), a)


def test_twiddle_prefix_property():
    # fri/mod.rs:318-321 relies on roots[..m/2] of a bigger table being the table of size m
    big = O.twiddles(12, inverse=True)
    for log_m in range(1, 12):
        assert np.array_equal(big[:(1 << log_m) // 2], O.twiddles(log_m, inverse=True))


def test_lde_layout_is_bitreversed_enumeration_of_big_coset():
    log_n, log_lde = 5, 3
    rng = np.random.default_rng(3)
    mono = rand_gl(rng, (2, 1 << log_n))
    lde = O.lde_batch(mono, log_lde, threads=2)
    N = log_n + log_lde
    for col in range(2):
        flat = lde[col].reshape(-1)
        for I in [0, 1, 17, 100, 255]:
            x = 7 * pow(O.omega(N), O.bitrev(I, N), P) % P
            val = sum(int(c) * pow(x, i, P) for i, c in enumerate(mono[col])) % P
            assert int(flat[I]) == val


def test_fft_batch_threads_agree():
    rng = np.random.default_rng(5)
    cols = rand_gl(rng, (6, 1 << 9))
    a = O.fft_batch(cols, 7, threads=1)
    b = O.fft_batch(cols, 7, threads=4)
    assert np.array_equal(a, b)
    assert np.array_equal(a[3], O.fft_natural_to_bitreversed(cols[3], 7))
    assert np.array_equal(O.ifft_batch(O.bitreverse(a[0])[None, :], 7)[0], cols[0])


def test_do_fri_on_low_degree_codeword():
    """Codeword of a degree < n polynomial over F_p^2 on the LDE domain: do_fri must end in final monomials whose
    high part is zero (fri/mod.rs:327-336 self-check) and which evaluate consistently with the fold chain."""
    log_n, log_lde, cap = 8, 2, 4
    rng = np.random.default_rng(11)
    mono = rand_gl(rng, (2, 1 << log_n))
    lde = O.lde_batch(mono, log_lde)
    c0, c1 = lde[0].reshape(-1), lde[1].reshape(-1)
    _, _, sched, final_degree = O.fri_schedule(20, cap, 0, log_lde, log_n)
    t = O.Transcript()
    t.absorb([1, 2, 3])
    res = O.do_fri(c0, c1, log_lde, sched, cap, t)
    f0, f1 = res["final_monomials"]
    assert res["final_degree"] == final_degree
    assert not f0[final_degree:].any() and not f1[final_degree:].any()
    # oracle i+1's source equals k folds of oracle i's source with squared challenges
    roots = O.twiddles(log_n + log_lde, inverse=True)
    kappa = O.inv(7)
    srcs = res["sources"] + [res["last_folded"]]
    for i, k in enumerate(sched):
        a0, a1 = srcs[i]
        ch = res["challenges"][i]
        for _ in range(k):
            a0, a1 = O.fri_fold(a0, a1, roots, kappa, ch)
            kappa = kappa * kappa % P
            ch = ((ch[0] * ch[0] + 7 * ch[1] * ch[1]) % P, 2 * ch[0] * ch[1] % P)
        assert np.array_equal(a0, srcs[i + 1][0]) and np.array_equal(a1, srcs[i + 1][1])
    # Merkle proofs of every oracle verify against its cap
    for i, k in enumerate(sched):
        s0, s1 = srcs[i]
        E = 1 << k
        leaves = s0.size // E
        for j in (0, leaves - 1, leaves // 3):
            leaf = np.concatenate([s0[j * E:(j + 1) * E], s1[j * E:(j + 1) * E]])
            lh, path = O.merkle_proof(res["trees"][i], leaves, cap, j)
            assert np.array_equal(lh, O.hash_leaf(leaf))
            assert O.merkle_verify(path, res["caps"][i], lh, j)


def test_batch_inverse():
    rng = np.random.default_rng(2)
    a = rand_gl(rng, 1000) | np.uint64(1)
    inv = O.batch_inverse(a)
    for x, y in zip(a[:50], inv[:50]):
        assert int(x) % P * int(y) % P == 1
