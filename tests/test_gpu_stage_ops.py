"""-m gpu parity at KERNEL level for SURVEY §8 rows a9 / a10 / a12 (seam S2): bj_copy_perm_stage2, bj_lookup_polys and the three
quotient-term operators bj_quotient_{gates,lookup,copy_perm} against the CPU restatements of
compute_partial_products_in_extension (copy_permutation.rs:649-760), compute_lookup_poly_pairs_specialized
(lookup_argument_in_ext.rs:320-700) and the quotient terms (copy_permutation.rs:1000-1249, lookup_argument_in_ext.rs:949-1319,
prover.rs:1031-1227, utils.rs:770-817) in oracle/prover_ops.c — so that a regression localises to one kernel instead of
"the stage-2 cap differs"."""
import numpy as np
import pytest

import era_boojum_amd as E
import oracle as O
from era_boojum_amd import synthetic as S
from gpu_util import DevBuf, ctx, rand_gl
from oracle import prover as OP

pytestmark = pytest.mark.gpu
P = O.P

BETA, GAMMA = (0x1234567890ABCDEF % P, 0x0FEDCBA987654321), (77, P - 5)
LBETA, LGAMMA = (P - 1, 3), (0xDEADBEEFCAFEF00D % P, 0x1111111122222222)
ALPHA = (0x9E3779B97F4A7C15 % P, 0xBF58476D1CE4E5B9 % P)


def _circuit(log_n, **kw):
    c = S.sha_shaped_circuit(log_n, seed=31 + log_n, table_bits=2, **kw)
    S.check_satisfied(c)
    return c


@pytest.mark.parametrize("log_n,kw", [(8, {}), (11, {}), (13, dict(num_gp_vars=24, lookup_width=3, lookup_reps=4, num_public_inputs=0)),
                                      (10, dict(num_gp_vars=7, lookup_width=4, lookup_reps=1, num_public_inputs=1))])
def test_copy_permutation_grand_product_and_partial_products(log_n, kw):
    c = _circuit(log_n, **kw)
    q, V, n = c.quotient_degree, c.num_vars, c.n
    wz, wp = OP.copy_perm_stage2(c.variables, c.sigmas, c.non_residues, log_n, q, BETA, GAMMA, threads=8)
    n_chunks = (V + q - 1) // q
    d_v, d_s = DevBuf(c.variables), DevBuf(c.sigmas)
    d_z, d_p = DevBuf(nelems=2 * n), DevBuf(nelems=max(1, 2 * (n_chunks - 1) * n))
    ctx().copy_perm_stage2(d_v.ptr, n, d_s.ptr, n, c.non_residues, V, q, log_n, BETA, GAMMA, d_z.ptr, d_p.ptr if n_chunks > 1 else None)
    z = d_z.get((2, n))
    assert z[0][0] == 1 and z[1][0] == 0                          # z(1) = 1 (copy_permutation.rs:425-510: shifted grand product)
    assert np.array_equal(z, wz)
    if n_chunks > 1:
        assert np.array_equal(d_p.get((n_chunks - 1, 2, n)), wp)
    # multipliers k_c that do not fit 32 bits take the 64 x 64-bit product path (make_non_residues' own output always fits: the
    # default path multiplies by it as a 32-bit integer); the operator is the same function of whatever k_c it is given
    big = [(int(k) * 0x9E3779B97F4A7C15 + 12345) % P for k in c.non_residues]
    assert max(big) >= 1 << 32
    wz2, wp2 = OP.copy_perm_stage2(c.variables, c.sigmas, big, log_n, q, BETA, GAMMA, threads=8)
    ctx().copy_perm_stage2(d_v.ptr, n, d_s.ptr, n, big, V, q, log_n, BETA, GAMMA, d_z.ptr, d_p.ptr if n_chunks > 1 else None)
    assert np.array_equal(d_z.get((2, n)), wz2) and not np.array_equal(wz2, wz)
    if n_chunks > 1:
        assert np.array_equal(d_p.get((n_chunks - 1, 2, n)), wp2)
    # a satisfied copy permutation closes the cycle: z(omega^(n-1)) * last row's full ratio = 1 is implied by the quotient test
    for d in (d_v, d_s, d_z, d_p):
        d.free()


def test_copy_permutation_with_noncanonical_inputs_and_strided_columns():
    log_n = 9
    c = _circuit(log_n)
    n, V, q = c.n, c.num_vars, c.quotient_degree
    rng = np.random.default_rng(3)
    stride = n + 64
    vars_p = np.zeros((V, stride), dtype=np.uint64)
    sig_p = np.zeros((V, stride), dtype=np.uint64)
    vars_p[:, :n], sig_p[:, :n] = c.variables, c.sigmas
    bump = rng.random((V, n)) < 0.1                                # same residues, representatives in [p, 2^64)
    small = c.variables < np.uint64((1 << 32) - 1)
    vars_p[:, :n][bump & small] += np.uint64(P)
    wz, wp = OP.copy_perm_stage2(c.variables, c.sigmas, c.non_residues, log_n, q, BETA, GAMMA, threads=8)
    n_chunks = (V + q - 1) // q
    d_v, d_s, d_z, d_p = DevBuf(vars_p), DevBuf(sig_p), DevBuf(nelems=2 * n), DevBuf(nelems=2 * (n_chunks - 1) * n)
    ctx().copy_perm_stage2(d_v.ptr, stride, d_s.ptr, stride, c.non_residues, V, q, log_n, BETA, GAMMA, d_z.ptr, d_p.ptr)
    assert np.array_equal(d_z.get((2, n)), wz) and np.array_equal(d_p.get((n_chunks - 1, 2, n)), wp)
    for d in (d_v, d_s, d_z, d_p):
        d.free()


def _lookup_vars(c):
    """The lookup sub-arguments' variable columns: width per sub-argument, width + 1 with the table id as a variable."""
    return np.ascontiguousarray(c.variables[c.num_gp_vars:c.num_gp_vars + c.lookup_reps * c.lookup_cols_per_sub])


# table_id_as_variable: LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (lookup_argument_in_ext.rs:354-366) — the
# operators take d_table_id = NULL and read the id from the (width+1)-th variable column of every sub-argument
@pytest.mark.parametrize("log_n,kw", [(8, {}), (12, {}), (10, dict(num_gp_vars=24, lookup_width=3, lookup_reps=4, num_public_inputs=0)),
                                      (9, dict(num_gp_vars=20, lookup_width=4, lookup_reps=1, num_public_inputs=0)),
                                      (11, dict(table_id_as_variable=True)),
                                      (9, dict(num_gp_vars=20, lookup_width=3, lookup_reps=11, num_public_inputs=0, table_id_as_variable=True))])
def test_lookup_polynomials(log_n, kw):
    c = _circuit(log_n, **kw)
    n, reps, w = c.n, c.lookup_reps, c.lookup_width
    lv = _lookup_vars(c)
    tid = OP.lookup_table_id(c, c.constants)
    wA, wB = OP.lookup_polys(lv, tid, c.tables, c.multiplicities[0], reps, w, log_n, LBETA, LGAMMA, threads=8)
    d_l, d_t, d_tab, d_m = DevBuf(lv), DevBuf(tid if tid is not None else np.zeros(1, dtype=np.uint64)), DevBuf(c.tables), DevBuf(c.multiplicities[0])
    d_A, d_B = DevBuf(nelems=2 * reps * n), DevBuf(nelems=2 * n)
    ctx().lookup_polys(d_l.ptr, n, d_t.ptr if tid is not None else None, d_tab.ptr, n, d_m.ptr, reps, w, log_n, LBETA, LGAMMA, d_A.ptr, d_B.ptr)
    A, B = d_A.get((reps, 2, n)), d_B.get((2, n))
    assert np.array_equal(A, wA) and np.array_equal(B, wB)
    # the log-derivative argument itself: sum_rows (sum_i A_i - B) = 0 (lookup_argument_in_ext.rs, the sumcheck the verifier does at 0)
    tot0 = (int(A[:, 0, :].astype(object).sum()) - int(B[0].astype(object).sum())) % P
    tot1 = (int(A[:, 1, :].astype(object).sum()) - int(B[1].astype(object).sum())) % P
    assert tot0 == 0 and tot1 == 0
    for d in (d_l, d_t, d_tab, d_m, d_A, d_B):
        d.free()


def _quotient_inputs(c):
    """LDEs of every column the quotient reads, restricted to the first q cosets, by the oracle; plus the alpha powers."""
    log_n, q, V = c.log_n, c.quotient_degree, c.num_vars
    log_q = q.bit_length() - 1
    Q = c.n * q
    z, partials = OP.copy_perm_stage2(c.variables, c.sigmas, c.non_residues, log_n, q, BETA, GAMMA, threads=8)
    stage2 = [z[0], z[1]] + [partials[j][k] for j in range(partials.shape[0]) for k in range(2)]
    reps, w = c.lookup_reps, c.lookup_width
    A, B = OP.lookup_polys(_lookup_vars(c), OP.lookup_table_id(c, c.constants), c.tables, c.multiplicities[0], reps, w, log_n, LBETA, LGAMMA,
                           threads=8)
    stage2 += [A[i][k] for i in range(reps) for k in range(2)] + [B[0], B[1]]

    def lde_q(cols):
        cols = np.ascontiguousarray(np.stack(cols) if isinstance(cols, list) else cols)
        return np.ascontiguousarray(O.lde_batch(O.ifft_batch(cols, 1, threads=8), log_q, threads=8).reshape(cols.shape[0], Q))
    d = dict(vars=lde_q(c.variables), mult=lde_q(c.multiplicities[:1])[0], sig=lde_q(c.sigmas), con=lde_q(c.constants),
             tab=lde_q(c.tables), s2=lde_q(stage2))
    n_part = partials.shape[0]
    d["n_part"], d["Q"], d["log_q"] = n_part, Q, log_q
    n_lookup, n_gate = reps + 1, sum(g.reps * g.num_terms for g in c.gates)
    n_chunks = (V + q - 1) // q
    al = [(1, 0)]
    while len(al) < n_lookup + n_gate + 1 + n_chunks:
        al.append(OP.emul(al[-1], ALPHA))
    d["alphas"], d["n_lookup"], d["n_gate"], d["n_chunks"] = al, n_lookup, n_gate, n_chunks
    return d


def _oracle_quotient(c, d, alphas):
    s2, n_part, reps = d["s2"], d["n_part"], c.lookup_reps
    o = 2 + 2 * n_part
    return OP.quotient(d["vars"], d["con"], d["sig"], np.ascontiguousarray(s2[0:2]), np.ascontiguousarray(s2[2:o]),
                       np.ascontiguousarray(s2[o:o + 2 * reps]), np.ascontiguousarray(s2[o + 2 * reps:]), d["mult"], d["tab"], c,
                       d["log_q"], alphas, BETA, GAMMA, LBETA, LGAMMA, threads=8)


@pytest.mark.parametrize("log_n,kw", [(9, {}), (11, dict(num_gp_vars=24, lookup_width=3, lookup_reps=4, num_public_inputs=0)),
                                      (10, dict(table_id_as_variable=True))])
def test_quotient_term_kernels_one_by_one_and_together(log_n, kw):
    c = _circuit(log_n, **kw)
    d = _quotient_inputs(c)
    Q, V, q, n = d["Q"], c.num_vars, c.quotient_degree, c.n
    reps, w, n_part = c.lookup_reps, c.lookup_width, d["n_part"]
    al, nl, ng, nch = d["alphas"], d["n_lookup"], d["n_gate"], d["n_chunks"]
    zero = [(0, 0)]
    a_lookup, a_gates, a_cp = al[:nl], al[nl:nl + ng], al[nl + ng:]
    bufs = {k: DevBuf(d[k]) for k in ("vars", "mult", "sig", "con", "tab", "s2")}
    out = DevBuf(nelems=2 * Q)
    o0, o1 = out.ptr, out.ptr + 8 * Q
    lv_ptr = bufs["vars"].ptr + 8 * Q * c.num_gp_vars
    tid_ptr = None if c.table_id_as_variable else bufs["con"].ptr + 8 * Q * c.table_id_col
    A_ptr = bufs["s2"].ptr + 8 * Q * (2 + 2 * n_part)
    B_ptr = A_ptr + 8 * Q * 2 * reps
    C = ctx()

    def gates(alphas):
        C.quotient_gates(bufs["vars"].ptr, Q, c.num_gp_vars, bufs["con"].ptr, Q, c.num_constant_cols, c.gates, alphas, Q, o0, o1)

    def lookup(alphas):
        C.quotient_lookup(lv_ptr, Q, tid_ptr, bufs["tab"].ptr, Q, bufs["mult"].ptr, A_ptr, B_ptr, Q, reps, w, LBETA, LGAMMA, alphas,
                          Q, o0, o1)

    def copy_perm(alphas, points=Q, first=0, non_residues=None):
        off = 8 * first
        C.quotient_copy_perm(bufs["vars"].ptr + off, Q, bufs["sig"].ptr + off, Q, bufs["s2"].ptr + off, Q,
                             c.non_residues if non_residues is None else non_residues, V, q,
                             log_n, d["log_q"], BETA, GAMMA, alphas, points, first, o0 + off, o1 + off)

    def clear():
        C.h2d(out.ptr, np.zeros(2 * Q, dtype=np.uint64))

    # all three, in the prover's order
    gates(a_gates); lookup(a_lookup); copy_perm(a_cp)
    want = _oracle_quotient(c, d, al)
    assert np.array_equal(out.get((2, Q)), want)
    # the numerator of a satisfied circuit is divisible by x^n - 1: T then has degree < q n - 1, the check the reference prover
    # makes on the top coefficient (prover.rs:1425-1438)
    mono = O.ifft_batch(np.stack([O.bitreverse(want[0]), O.bitreverse(want[1])]), 7, threads=2)
    assert mono[0][-1] == 0 and mono[1][-1] == 0
    # gate terms alone (then the division by the vanishing polynomial through the third operator with zero challenges): once
    # through the kernel that reads every variable column once for all gates (windows of 20 columns), once through the
    # per-gate kernel that circuits with several gates of one kind fall back to
    want_gates = _oracle_quotient(c, d, zero * nl + a_gates + zero * (1 + nch))
    gates(a_gates); copy_perm(zero * (1 + nch))
    assert np.array_equal(out.get((2, Q)), want_gates)
    import os
    os.environ["BJ_GATES_WINDOWED"] = "0"          # the switches are read once per process: bj_env_reload re-reads them
    try:
        E.load_library().bj_env_reload()
        clear(); gates(a_gates); copy_perm(zero * (1 + nch))
    finally:
        del os.environ["BJ_GATES_WINDOWED"]
        E.load_library().bj_env_reload()
    assert np.array_equal(out.get((2, Q)), want_gates)
    # lookup terms alone
    clear(); lookup(a_lookup); copy_perm(zero * (1 + nch))
    assert np.array_equal(out.get((2, Q)), _oracle_quotient(c, d, a_lookup + zero * (ng + 1 + nch)))
    # (z - 1) L1 and the copy-permutation chain alone, computed as two coset ranges (what two GPUs of a sharded proof do)
    clear(); copy_perm(a_cp, Q // 2, 0); copy_perm(a_cp, Q // 2, Q // 2)
    assert np.array_equal(out.get((2, Q)), _oracle_quotient(c, d, zero * (nl + ng) + a_cp))
    # the L1 term alone
    clear(); copy_perm(a_cp[:1] + zero * nch)
    assert np.array_equal(out.get((2, Q)), _oracle_quotient(c, d, zero * (nl + ng) + a_cp[:1] + zero * nch))
    # the copy-permutation chain with multipliers that do not fit 32 bits (the 64 x 64-bit product path; not a satisfied argument any
    # more, the operator is the same function of whatever k_c it gets)
    import copy
    cb = copy.copy(c)
    cb.non_residues = [(int(k) * 0x9E3779B97F4A7C15 + 12345) % P for k in c.non_residues]
    clear(); copy_perm(a_cp, non_residues=cb.non_residues)
    assert np.array_equal(out.get((2, Q)), _oracle_quotient(cb, d, zero * (nl + ng) + a_cp))
    for b in list(bufs.values()) + [out]:
        b.free()


def test_stage_operators_report_bad_arguments():
    import era_boojum_amd as E
    d = DevBuf(nelems=64)
    with pytest.raises(E.BoojumHipError):
        ctx().copy_perm_stage2(d.ptr, 4, d.ptr, 8, [1], 1, 4, 3, BETA, GAMMA, d.ptr, d.ptr)          # stride below n
    with pytest.raises(E.BoojumHipError):
        ctx().lookup_polys(d.ptr, 8, d.ptr, d.ptr, 8, d.ptr, 1, 9, 3, BETA, GAMMA, d.ptr, d.ptr)      # width 9
    g = S.sha_bench_gates(60, 4)
    with pytest.raises(E.BoojumHipError, match="reads past"):
        ctx().quotient_gates(d.ptr, 8, 3, d.ptr, 8, 8, g, [(1, 0)] * 40, 8, d.ptr, d.ptr)              # 60-column gates on 3 columns
    with pytest.raises(E.BoojumHipError):
        ctx().quotient_copy_perm(d.ptr, 8, d.ptr, 8, d.ptr, 8, [1], 1, 4, 3, 7, BETA, GAMMA, [(1, 0)] * 2, 8, 0, d.ptr, d.ptr)   # LDE 128
    d.free()


@pytest.mark.parametrize("world,log_n", [(2, 9), (4, 9), (8, 9), (8, 5), (1, 6)])
def test_residue_combination_of_the_sharded_quotient(world, log_n):
    """bj_combine_residues against the definition: T (two columns, degree < q n) is reduced modulo x^E - a_i by hand (R_i[k] =
    sum_j a_i^j T[j E + k]) for the moduli the sharded prover uses — a_i = x_{I0_i}^E, I0_i = the first LDE point of rank i, E =
    q n / W points per rank — and the device's Vandermonde solve must give T back.  Equal moduli are refused."""
    from tests import sharding_model as M
    q, log_lde = 4, 3
    n, P = 1 << log_n, E.P
    Elen = q * n // world
    rng = np.random.default_rng(world * 100 + log_n)
    T = rng.integers(0, P, size=(2, q * n), dtype=np.uint64)
    a = [pow(M.lde_coset_shift(log_n, log_lde, r * ((1 << log_lde) // world)), Elen, P) for r in range(world)]
    res = np.zeros((world, 2, Elen), dtype=np.uint64)
    for i in range(world):
        for col in range(2):
            acc = np.zeros(Elen, dtype=object)
            ap = 1
            for j in range(world):
                acc = (acc + ap * T[col, j * Elen:(j + 1) * Elen].astype(object)) % P
                ap = ap * a[i] % P
            res[i, col] = acc.astype(np.uint64)
    d_res, d_out = DevBuf(res), DevBuf(nelems=2 * q * n)
    ctx().combine_residues(d_res.ptr, world, Elen, 2, a, d_out.ptr)
    assert np.array_equal(d_out.get((2, q * n)), T)
    if world > 1:
        with pytest.raises(E.BoojumHipError, match="pairwise distinct"):
            ctx().combine_residues(d_res.ptr, world, Elen, 2, [a[0]] * world, d_out.ptr)
    d_res.free(); d_out.free()
