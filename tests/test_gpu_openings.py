"""-m gpu parity: barycentric evaluation and DEEP quotient accumulation (through the C ABI) vs the CPU oracle
(whose DEEP point function is pinned by the golden proof, tests/test_oracle_fixture.py)."""
import numpy as np
import pytest

import oracle as O
from gpu_util import DevBuf, ctx, rand_gl, P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", [0, 1, 3, 8, 12, 15])
def test_barycentric_weights_and_eval_match_oracle(log_n):
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    z = (int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))
    w0, w1 = O.barycentric_weights(log_n, 7, z)
    d_w = DevBuf(nelems=2 * n)
    ctx().barycentric_weights(log_n, 7, z, d_w.ptr, d_w.ptr + 8 * n)
    got = d_w.get((2, n))
    assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1)
    n_cols = 11
    cols = rand_gl(rng, (n_cols, n), noncanonical=True)
    d_c = DevBuf(cols)
    out = ctx().barycentric_eval_batch([d_c.ptr + 8 * n * c for c in range(n_cols)], log_n, d_w.ptr, d_w.ptr + 8 * n)
    for c in range(n_cols):
        assert (int(out[c][0]), int(out[c][1])) == O.barycentric_eval_base(cols[c], w0, w1), c
    # extension-valued polynomial stored as two columns: combine on the host as the ABI documents
    e0, e1 = out[0], out[1]
    comb = ((int(e0[0]) + 7 * int(e1[1])) % P, (int(e0[1]) + int(e1[0])) % P)
    if n > 1:
        assert comb == O.barycentric_eval_ext(cols[0], cols[1], w0, w1)
    d_w.free(); d_c.free()


def test_barycentric_equals_polynomial_evaluation_end_to_end():
    """trace -> (GPU) monomials -> LDE; evaluate every column at z from coset 0 of the LDE; compare with Horner."""
    log_n, log_lde, n_cols = 10, 2, 3
    n = 1 << log_n
    rng = np.random.default_rng(5)
    mono = rand_gl(rng, (n_cols, n))
    d_m, d_l = DevBuf(mono), DevBuf(nelems=n_cols * n << log_lde)
    ctx().lde_batch(d_m.ptr, d_l.ptr, log_n, n_cols, log_lde)
    z = (1234567, 7654321)
    d_w = DevBuf(nelems=2 * n)
    ctx().barycentric_weights(log_n, 7, z, d_w.ptr, d_w.ptr + 8 * n)
    out = ctx().barycentric_eval_batch([d_l.ptr + 8 * (n << log_lde) * c for c in range(n_cols)], log_n, d_w.ptr, d_w.ptr + 8 * n)
    for c in range(n_cols):
        acc, zp = (0, 0), (1, 0)
        for coef in mono[c]:
            acc = ((acc[0] + int(coef) * zp[0]) % P, (acc[1] + int(coef) * zp[1]) % P)
            zp = ((zp[0] * z[0] + 7 * zp[1] * z[1]) % P, (zp[0] * z[1] + zp[1] * z[0]) % P)
        assert (int(out[c][0]), int(out[c][1])) == acc
    d_m.free(); d_l.free(); d_w.free()


@pytest.mark.parametrize("log_n,log_lde,n_base,n_ext", [(4, 1, 1, 0), (6, 2, 5, 3), (10, 3, 40, 9), (13, 1, 3, 1)])
def test_deep_quotient_matches_oracle(log_n, log_lde, n_base, n_ext):
    N = 1 << (log_n + log_lde)
    rng = np.random.default_rng(log_n * 7 + n_ext)
    base_cols = rand_gl(rng, (n_base, N), noncanonical=True)
    ext_cols = rand_gl(rng, (max(n_ext, 1), 2, N))
    k = n_base + n_ext
    values = rand_gl(rng, (k, 2))
    challenges = rand_gl(rng, (k, 2))
    at = (int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64)))
    dst = rand_gl(rng, (2, N), noncanonical=True)
    srcs = [(base_cols[i], None) for i in range(n_base)] + [(ext_cols[i][0], ext_cols[i][1]) for i in range(n_ext)]
    w0, w1 = dst[0].copy(), dst[1].copy()
    O.deep_quotient_accumulate(srcs, values, challenges, at, log_n, log_lde, w0, w1, threads=4)
    d_b, d_e, d_d = DevBuf(base_cols), DevBuf(ext_cols), DevBuf(dst)
    dsrc = [(d_b.ptr + 8 * N * i, None) for i in range(n_base)] + \
           [(d_e.ptr + 8 * N * (2 * i), d_e.ptr + 8 * N * (2 * i + 1)) for i in range(n_ext)]
    ctx().deep_quotient_accumulate(dsrc, values, challenges, at, log_n, log_lde, d_d.ptr, d_d.ptr + 8 * N, accumulate=True)
    got = d_d.get((2, N))
    assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1)
    # overwrite mode == accumulate into zeros
    z0, z1 = np.zeros(N, dtype=np.uint64), np.zeros(N, dtype=np.uint64)
    O.deep_quotient_accumulate(srcs, values, challenges, at, log_n, log_lde, z0, z1, threads=4)
    ctx().deep_quotient_accumulate(dsrc, values, challenges, at, log_n, log_lde, d_d.ptr, d_d.ptr + 8 * N, accumulate=False)
    got = d_d.get((2, N))
    assert np.array_equal(got[0], z0) and np.array_equal(got[1], z1)
    d_b.free(); d_e.free(); d_d.free()


@pytest.mark.timeout(240)
def test_deep_quotient_at_a_trace_domain_point_2p25():
    """The opening set of a public input: at = (omega_n^row, 0) over the 2^25-point LDE domain.  In one thread of this launch
    the running product of the four denominators has a 2-power order, so the inversion chain squares 2^48 (2^96 = -1):
    the borrow-without-carry branch of gl::mul_weak, taken in the middle of a counted loop.  (Its s_andn2_b64 writes SCC;
    with SCC missing from the asm clobbers the loop lost its exit test and ran 2^29 more trips.)"""
    log_n, log_lde = 22, 3
    N = 1 << (log_n + log_lde)
    rng = np.random.default_rng(2225)
    col = rand_gl(rng, (1, N))
    values, challenges = rand_gl(rng, (1, 2)), rand_gl(rng, (1, 2))
    values[0][1] = 0
    om = pow(0x185629dcda58878c, 1 << (32 - log_n), P)
    at = (pow(om, 5, P), 0)
    dst = rand_gl(rng, (2, N))
    w0, w1 = dst[0].copy(), dst[1].copy()
    O.deep_quotient_accumulate([(col[0], None)], values, challenges, at, log_n, log_lde, w0, w1, threads=8)
    d_b, d_d = DevBuf(col), DevBuf(dst)
    ctx().deep_quotient_accumulate([(d_b.ptr, None)], values, challenges, at, log_n, log_lde, d_d.ptr, d_d.ptr + 8 * N,
                                   accumulate=True)
    got = d_d.get((2, N))
    assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1)
    d_b.free(); d_d.free()


def test_deep_then_fri_is_low_degree():
    """Size-independent property: the DEEP combination of true openings of low-degree polynomials is itself a
    low-degree codeword, so bj_fri_prove accepts it (and rejects it if one opening value is wrong)."""
    import era_boojum_amd as E
    log_n, log_lde, n_cols, cap = 9, 2, 6, 4
    n, N = 1 << log_n, 1 << (log_n + log_lde)
    rng = np.random.default_rng(77)
    mono = rand_gl(rng, (n_cols, n))
    d_m, d_l = DevBuf(mono), DevBuf(nelems=n_cols * N)
    ctx().lde_batch(d_m.ptr, d_l.ptr, log_n, n_cols, log_lde)
    z = (424242, 171717)
    d_w = DevBuf(nelems=2 * n)
    ctx().barycentric_weights(log_n, 7, z, d_w.ptr, d_w.ptr + 8 * n)
    cols = [d_l.ptr + 8 * N * c for c in range(n_cols)]
    vals = ctx().barycentric_eval_batch(cols, log_n, d_w.ptr, d_w.ptr + 8 * n)
    chs = rand_gl(rng, (n_cols, 2))
    d_d = DevBuf(nelems=2 * N)
    _, _, sched, _ = E.fri_schedule(40, cap, 0, log_lde, log_n)
    for corrupt in (False, True):
        v = vals.copy()
        if corrupt:
            v[2][0] = (int(v[2][0]) + 1) % P
        ctx().deep_quotient_accumulate([(c, None) for c in cols], v, chs, z, log_n, log_lde, d_d.ptr, d_d.ptr + 8 * N,
                                       accumulate=False)
        t = E.Transcript()
        if corrupt:
            with pytest.raises(E.BoojumHipError):
                ctx().fri_prove(d_d.ptr, d_d.ptr + 8 * N, log_n, log_lde, sched, cap, t)
        else:
            ctx().fri_prove(d_d.ptr, d_d.ptr + 8 * N, log_n, log_lde, sched, cap, t).close()
    for b in (d_m, d_l, d_w, d_d):
        b.free()
