"""-m gpu parity: HIP FRI fold (through the C ABI) vs the CPU oracle and the golden proof's FRI chain."""
import numpy as np
import pytest

import oracle as O
from gpu_util import DevBuf, ctx, rand_gl, P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_len", [1, 2, 5, 12, 18])
def test_fold_matches_oracle(log_len):
    log_full = 20
    rng = np.random.default_rng(log_len)
    ln = 1 << log_len
    c = rand_gl(rng, (2, ln), noncanonical=True)
    ch = (int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))
    kappa = pow(O.inv(7), 4, P)
    roots = O.twiddles(log_full, inverse=True)
    w0, w1 = O.fri_fold(c[0], c[1], roots, kappa, ch)
    d_c, d_o = DevBuf(c), DevBuf(nelems=ln)
    ctx().fri_fold(d_c.ptr, d_c.ptr + 8 * ln, ln, d_o.ptr, d_o.ptr + 8 * (ln // 2), log_full, kappa, ch)
    got = d_o.get((2, ln // 2))
    assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1)
    d_c.free(); d_o.free()


def test_fold_chain_equals_oracle_do_fri():
    """Drive a whole do_fri schedule with the HIP fold + HIP chunked trees and compare every oracle cap and the last
    folded layer with the CPU oracle's do_fri (fri/mod.rs:49-358)."""
    log_n, log_lde, cap = 10, 2, 4
    rng = np.random.default_rng(21)
    mono = rand_gl(rng, (2, 1 << log_n))
    lde = O.lde_batch(mono, log_lde, threads=2)
    c0, c1 = lde[0].reshape(-1), lde[1].reshape(-1)
    _, _, sched, _ = O.fri_schedule(40, cap, 0, log_lde, log_n)
    t = O.Transcript(); t.absorb([5, 6, 7])
    ref = O.do_fri(c0, c1, log_lde, sched, cap, t, threads=2)
    log_full = log_n + log_lde
    ln = 1 << log_full
    cur = DevBuf(np.stack([c0, c1]))
    kappa = O.inv(7)
    for step, k in enumerate(sched):
        leaves = ln >> k
        nd = ctx().merkle_tree_digests(leaves, cap)
        d_t = DevBuf(nelems=4 * nd)
        ctx().merkle_tree_build_chunked(cur.ptr, cur.ptr + 8 * ln, ln, k, cap, d_t.ptr)
        assert np.array_equal(d_t.get((nd, 4)), ref["trees"][step])
        d_t.free()
        ch = ref["challenges"][step]
        for _ in range(k):
            nxt = DevBuf(nelems=ln)
            ctx().fri_fold(cur.ptr, cur.ptr + 8 * ln, ln, nxt.ptr, nxt.ptr + 8 * (ln // 2), log_full, kappa, ch)
            cur.free(); cur = nxt; ln //= 2
            kappa = kappa * kappa % P
            ch = ((ch[0] * ch[0] + 7 * ch[1] * ch[1]) % P, 2 * ch[0] * ch[1] % P)
    got = cur.get((2, ln))
    assert np.array_equal(got[0], ref["last_folded"][0]) and np.array_equal(got[1], ref["last_folded"][1])
    # final interpolation on the GPU: bit-reverse + iNTT with coset kappa^-1 (fri/mod.rs:312-343)
    lm = ln.bit_length() - 1
    ctx().bitreverse_batch(cur.ptr, cur.ptr, lm, 2)
    ctx().intt_batch(cur.ptr, cur.ptr, lm, 2, coset=O.inv(kappa))
    fin = cur.get((2, ln))
    assert np.array_equal(fin[0], ref["final_monomials"][0]) and np.array_equal(fin[1], ref["final_monomials"][1])
    assert not fin[:, ref["final_degree"]:].any()
    cur.free()


@pytest.mark.parametrize("k", [1, 2, 3])
def test_fused_fold_step_matches_k_single_folds(k):
    log_full, ln = 16, 1 << 14
    rng = np.random.default_rng(40 + k)
    c = rand_gl(rng, (2, ln), noncanonical=True)
    ch0 = (int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))
    roots = O.twiddles(log_full, inverse=True)
    kappa = pow(O.inv(7), 8, P)
    w0, w1, ch, kp = c[0], c[1], ch0, kappa
    for _ in range(k):
        w0, w1 = O.fri_fold(w0, w1, roots, kp, ch)
        kp = kp * kp % P
        ch = ((ch[0] * ch[0] + 7 * ch[1] * ch[1]) % P, 2 * ch[0] * ch[1] % P)
    d_c, d_o = DevBuf(c), DevBuf(nelems=2 * (ln >> k))
    ctx().fri_fold_step(d_c.ptr, d_c.ptr + 8 * ln, ln, k, d_o.ptr, d_o.ptr + 8 * (ln >> k), log_full, kappa, ch0)
    got = d_o.get((2, ln >> k))
    assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1)
    d_c.free(); d_o.free()


@pytest.mark.parametrize("log_n,log_lde,cap,sec", [(10, 2, 4, 40), (13, 3, 16, 100), (16, 1, 32, 100), (8, 3, 16, 60)])
def test_fri_prove_equals_oracle_do_fri(log_n, log_lde, cap, sec):
    """bj_fri_prove (do_fri on the device + host transcript of the product) vs the oracle: caps, challenges, final
    monomials, transcript state afterwards, and query openings that verify against the caps."""
    import era_boojum_amd as E
    rng = np.random.default_rng(log_n)
    mono = rand_gl(rng, (2, 1 << log_n))
    lde = O.lde_batch(mono, log_lde, threads=4)
    c0, c1 = lde[0].reshape(-1), lde[1].reshape(-1)
    _, nq, sched, final_degree = E.fri_schedule(sec, cap, 0, log_lde, log_n)
    to = O.Transcript(); to.absorb([11, 22, 33])
    ref = O.do_fri(c0, c1, log_lde, sched, cap, to, threads=4)
    tp = E.Transcript(); tp.absorb([11, 22, 33])
    N = 1 << (log_n + log_lde)
    d = DevBuf(np.stack([c0, c1]))
    fri = ctx().fri_prove(d.ptr, d.ptr + 8 * N, log_n, log_lde, sched, cap, tp)
    assert fri.num_oracles == len(sched) and fri.final_degree == final_degree == ref["final_degree"]
    for i in range(len(sched)):
        assert np.array_equal(fri.cap(i), ref["caps"][i]), i
        assert fri.challenge(i) == ref["challenges"][i]
    f0, f1 = fri.final_monomials()
    assert np.array_equal(f0, ref["final_monomials"][0][:final_degree])
    assert np.array_equal(f1, ref["final_monomials"][1][:final_degree])
    # both transcripts are in the same state afterwards -> same query indices
    qi = O.QueryIndexer(log_n, log_lde)
    for _ in range(min(nq, 8)):
        idx = tp.query_index(log_n, log_lde)
        assert idx == qi.next(to)
        f_idx, ln = idx, N
        for i, k in enumerate(sched):
            leaf, path = fri.query(i, f_idx, ln)
            E_ = 1 << k
            j = f_idx >> k
            s0, s1 = ref["sources"][i]
            want_leaf = np.concatenate([s0[j * E_:(j + 1) * E_], s1[j * E_:(j + 1) * E_]])
            assert np.array_equal(leaf, want_leaf)
            assert O.merkle_verify(path, ref["caps"][i], O.hash_leaf(leaf), j)
            f_idx >>= k
            ln >>= k
    fri.close(); d.free()


def test_fri_prove_rejects_high_degree_codeword():
    import era_boojum_amd as E
    rng = np.random.default_rng(3)
    N = 1 << 10
    c = rand_gl(rng, (2, N))          # random values are not a low-degree codeword
    d = DevBuf(c)
    tp = E.Transcript()
    with pytest.raises(E.BoojumHipError):
        ctx().fri_prove(d.ptr, d.ptr + 8 * N, 8, 2, [3, 3], 4, tp)
    d.free()
