"""-m gpu parity: HIP Poseidon2 leaf/node hashing and Merkle trees (through the C ABI) vs the CPU oracle and vs the
reference's golden proof fixture."""
import numpy as np
import pytest

import oracle as O
from gpu_util import DevBuf, ctx, rand_gl

pytestmark = pytest.mark.gpu


def test_permutation_matches_oracle():
    rng = np.random.default_rng(1)
    st = rand_gl(rng, (300, 12), noncanonical=True)
    st[0] = 0
    want = np.stack([O.poseidon2_permutation(s) for s in st])
    d = DevBuf(st)
    ctx().poseidon2_permute(d.ptr, st.shape[0])
    assert np.array_equal(d.get(st.shape), want)
    d.free()


def test_permutation_on_extreme_words():
    """The device permutation works on non-canonical ("weak") words with single-step carry corrections; drive it with
    every combination-rich pattern of boundary values so a missed double carry/borrow would show."""
    P = O.P
    specials = [0, 1, 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - 2, P - 1, P, P + 1, (1 << 64) - (1 << 32), (1 << 64) - 2,
                (1 << 64) - 1, (1 << 63), (1 << 63) - 1, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF, 0xFFFFFFFEFFFFFFFF]
    rng = np.random.default_rng(77)
    st = np.zeros((4096, 12), dtype=np.uint64)
    for i in range(st.shape[0]):
        for k in range(12):
            st[i, k] = specials[int(rng.integers(0, len(specials)))] if rng.random() < 0.8 else int(rng.integers(0, 1 << 63)) * 2 + 1
    st[0] = (1 << 64) - 1
    st[1] = P - 1
    st[2] = P
    want = np.stack([O.poseidon2_permutation(s) for s in st])
    d = DevBuf(st)
    ctx().poseidon2_permute(d.ptr, st.shape[0])
    assert np.array_equal(d.get(st.shape), want)
    d.free()


@pytest.mark.parametrize("n_cols", [1, 4, 7, 8, 9, 16, 17, 58, 93])
def test_tree_matches_oracle(n_cols):
    num_leaves, cap = 1 << 10, 16
    rng = np.random.default_rng(10 + n_cols)
    cols = rand_gl(rng, (n_cols, num_leaves), noncanonical=True)
    want = O.merkle_construct(cols, cap, threads=4)
    d_c = DevBuf(cols)
    d_t = DevBuf(nelems=want.size)
    ctx().merkle_tree_build(d_c.ptr, num_leaves, n_cols, num_leaves, cap, d_t.ptr)
    got = d_t.get(want.shape)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx().merkle_tree_cap(d_t.ptr, num_leaves, cap), O.merkle_cap(want, num_leaves, cap))
    for idx in (0, 1, 513, num_leaves - 1):
        leaf, path = ctx().merkle_tree_proof(d_t.ptr, num_leaves, cap, idx)
        wl, wp = O.merkle_proof(want, num_leaves, cap, idx)
        assert np.array_equal(leaf, wl) and np.array_equal(path, wp)
        assert O.merkle_verify(path, O.merkle_cap(want, num_leaves, cap), leaf, idx)
    # pointer-list variant, columns in a permuted order
    perm = list(reversed(range(n_cols)))
    ptrs = [d_c.ptr + 8 * num_leaves * c for c in perm]
    ctx().merkle_tree_build_ptrs(ptrs, num_leaves, cap, d_t.ptr)
    assert np.array_equal(d_t.get(want.shape), O.merkle_construct(cols[perm], cap, threads=4))
    d_c.free(); d_t.free()


@pytest.mark.parametrize("n_cols,stride_pad", [(1, 0), (5, 3), (8, 0)])
def test_narrow_tree_at_2p16_leaves_matches_oracle(n_cols, stride_pad):
    """Trees over at most eight columns (one absorption per leaf: the quotient oracle's shape) at a size where every node
    kernel variant is used, with a column stride larger than the leaf count."""
    num_leaves, cap = 1 << 16, 16
    rng = np.random.default_rng(900 + n_cols)
    stride = num_leaves + stride_pad
    buf = rand_gl(rng, (n_cols, stride), noncanonical=True)
    cols = np.ascontiguousarray(buf[:, :num_leaves])
    want = O.merkle_construct(cols, cap, threads=8)
    d_c = DevBuf(buf)
    d_t = DevBuf(nelems=want.size)
    ctx().merkle_tree_build(d_c.ptr, stride, n_cols, num_leaves, cap, d_t.ptr)
    assert np.array_equal(d_t.get(want.shape), want)
    d_c.free(); d_t.free()


@pytest.mark.parametrize("cap,num_leaves", [(1, 1), (1, 2), (4, 4), (32, 64)])
def test_tree_edge_shapes(cap, num_leaves):
    rng = np.random.default_rng(99)
    cols = rand_gl(rng, (5, num_leaves))
    want = O.merkle_construct(cols, cap)
    d_c, d_t = DevBuf(cols), DevBuf(nelems=want.size)
    ctx().merkle_tree_build(d_c.ptr, num_leaves, 5, num_leaves, cap, d_t.ptr)
    assert np.array_equal(d_t.get(want.shape), want)
    d_c.free(); d_t.free()


@pytest.mark.parametrize("log_e", [0, 1, 2, 3])
def test_chunked_tree_matches_oracle(log_e):
    ln, cap = 1 << 12, 8
    rng = np.random.default_rng(log_e)
    src = rand_gl(rng, (2, ln), noncanonical=True)
    want = O.merkle_construct_chunked(src, 1 << log_e, cap, threads=4)
    d_s, d_t = DevBuf(src), DevBuf(nelems=want.size)
    ctx().merkle_tree_build_chunked(d_s.ptr, d_s.ptr + 8 * ln, ln, log_e, cap, d_t.ptr)
    assert np.array_equal(d_t.get(want.shape), want)
    d_s.free(); d_t.free()


def test_golden_fixture_leaves_and_paths_on_gpu(fixture_json):
    """Known-answer test: the reference's own proof.json openings hashed by the HIP kernels must climb to the caps
    recorded in the proof (leaf widths 156 / 58 / 16 / 167, depth-16 paths)."""
    fx = fixture_json
    import test_oracle_fixture as TF
    # replay() is a pytest fixture function; call its body through the module helper
    t = O.Transcript()
    t.absorb_cap(fx["setup_merkle_tree_cap"]); t.absorb(fx["public_inputs"]); t.absorb_cap(fx["witness_oracle_cap"])
    for _ in range(4):
        t.challenge_ext()
    t.absorb_cap(fx["stage_2_oracle_cap"]); t.challenge_ext()
    t.absorb_cap(fx["quotient_oracle_cap"]); t.challenge_ext()
    for k in ("values_at_z", "values_at_z_omega", "values_at_0"):
        t.absorb(np.array(fx[k], dtype=np.uint64))
    t.challenge_ext()
    t.absorb_cap(fx["fri_base_oracle_cap"]); t.challenge_ext()
    for cap in fx["fri_intermediate_oracles_caps"]:
        t.absorb_cap(cap); t.challenge_ext()
    t.absorb(fx["final_fri_monomials"][0]); t.absorb(fx["final_fri_monomials"][1])
    qi = O.QueryIndexer(20, 1)
    caps = {"witness_query": fx["witness_oracle_cap"], "stage_2_query": fx["stage_2_oracle_cap"],
            "quotient_query": fx["quotient_oracle_cap"], "setup_query": fx["setup_merkle_tree_cap"]}
    d_t = DevBuf(nelems=4 * 3)
    for q in fx["queries"][:3]:
        idx = qi.next(t)
        for name, cap in caps.items():
            leaf = np.array(q[name]["leaf_elements"], dtype=np.uint64)
            # leaf hash on the GPU: a 1-leaf "tree" whose columns are the leaf's elements
            d_l = DevBuf(leaf)
            ctx().merkle_tree_build(d_l.ptr, 1, leaf.size, 1, 1, d_t.ptr)
            cur = d_t.get((3, 4))[0].copy()
            assert np.array_equal(cur, O.hash_leaf(leaf))
            d_l.free()
            # climb with GPU node hashing: 2-leaf tree over [left, right] digests, cap 1
            i = idx
            for sib in q[name]["proof"]:
                pair = np.array([cur, sib] if i & 1 == 0 else [sib, cur], dtype=np.uint64)
                ctx().h2d(d_t.ptr, pair.reshape(-1))
                ctx().merkle_tree_nodes(d_t.ptr, 2, 1)
                cur = d_t.get((3, 4))[2].copy()
                i >>= 1
            assert [int(x) for x in cur] == cap[i], name
    d_t.free()


def test_full_size_tree_properties():
    """2^20 leaves x 93 columns (witness-tree width of the SHA-shaped circuit, one coset of cfg3): sampled leaves and
    every sampled authentication path agree with the oracle; cap equals the oracle's node hashing of the GPU leaf layer."""
    num_leaves, n_cols, cap = 1 << 20, 93, 16
    rng = np.random.default_rng(4)
    cols = rand_gl(rng, (n_cols, num_leaves))
    d_c = DevBuf(cols)
    nd = ctx().merkle_tree_digests(num_leaves, cap)
    d_t = DevBuf(nelems=4 * nd)
    ctx().merkle_tree_build(d_c.ptr, num_leaves, n_cols, num_leaves, cap, d_t.ptr)
    tree = d_t.get((nd, 4))
    for I in rng.integers(0, num_leaves, size=64):
        assert np.array_equal(tree[I], O.hash_leaf(cols[:, I]))
    want = O.merkle_nodes_from_leaf_hashes(tree[:num_leaves], cap, threads=8)
    assert np.array_equal(tree, want)
    capv = ctx().merkle_tree_cap(d_t.ptr, num_leaves, cap)
    for I in rng.integers(0, num_leaves, size=8):
        leaf, path = ctx().merkle_tree_proof(d_t.ptr, num_leaves, cap, int(I))
        assert O.merkle_verify(path, capv, leaf, int(I))
    d_c.free(); d_t.free()


def _ext_matrix():
    m4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]                     # suggested_mds.rs:21-56
    return [[m4[i % 4][j % 4] * (2 if i // 4 == j // 4 else 1) for j in range(12)] for i in range(12)]


def _solve_mod_p(M, rhs):
    P = O.P
    n = len(M)
    A = [row[:] + [rhs[i]] for i, row in enumerate(M)]
    for c in range(n):
        piv = next(r for r in range(c, n) if A[r][c] % P)
        A[c], A[piv] = A[piv], A[c]
        inv = pow(A[c][c], P - 2, P)
        A[c] = [x * inv % P for x in A[c]]
        for r in range(n):
            if r != c and A[r][c]:
                f = A[r][c]
                A[r] = [(x - f * y) % P for x, y in zip(A[r], A[c])]
    return [A[i][n] for i in range(n)]


def test_permutation_takes_the_rare_branches_of_its_products():
    """States built so that every S-box of the first round sees x = 2^48 (x * x = 2^96 = -1: the product's final subtraction
    borrows without a carry, a 2^-32 event on random data that the scheduled instruction stream handles out of line), or other
    operands of the rare-vector fixture, in every word and in single words of an otherwise random state."""
    import json, os
    P = O.P
    rc0 = [int(x) % P for x in O.poseidon_round_constants()[0]]
    M = _ext_matrix()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gl_mul_rare.json")) as f:
        rare = [v["a"] for v in json.load(f)["vectors"]]
    rng = np.random.default_rng(3)
    states = []
    for t in range(64):
        target = [int(rng.integers(0, P, dtype=np.uint64)) for _ in range(12)]
        if t == 0:
            target = [1 << 48] * 12
        elif t < 13:
            target[t - 1] = 1 << 48                      # one word only: a lone lane of a chain takes the stub
        else:
            for k in range(12):
                if rng.random() < 0.5:
                    target[k] = [1 << 48, (1 << 48) * 3 % P, rare[int(rng.integers(0, len(rare)))] % P][int(rng.integers(0, 3))]
        st = _solve_mod_p(M, [(x - r) % P for x, r in zip(target, rc0)])
        chk = [(sum(M[i][j] * st[j] for j in range(12)) + rc0[i]) % P for i in range(12)]
        assert chk == target
        states.append(st)
    st = np.array(states, dtype=np.uint64)
    want = np.stack([O.poseidon2_permutation(s) for s in st])
    for reps in (1, 80):                                   # a single wave, and many waves with the rare lanes spread out
        batch = np.tile(st, (reps, 1))
        d = DevBuf(batch)
        ctx().poseidon2_permute(d.ptr, batch.shape[0])
        assert np.array_equal(d.get(batch.shape), np.tile(want, (reps, 1)))
        d.free()


def _ext_layer_carry_words(state, rc):
    """Which output words of the first external layer wrap in the fold of their low / high plane sums (tools/gen_p2_asm.py
    `combine`: T = A + B.hi * EPS, then T.hi + B.lo >= 2^32) — a model of the planes, not of the field arithmetic."""
    M = _ext_matrix()
    hit = []
    for i in range(12):
        A = sum(M[i][j] * (state[j] & 0xFFFFFFFF) for j in range(12)) + (rc[i] & 0xFFFFFFFF)
        B = sum(M[i][j] * (state[j] >> 32) for j in range(12)) + (rc[i] >> 32)
        T = A + (B >> 32) * 0xFFFFFFFF
        if (T >> 32) + (B & 0xFFFFFFFF) >= 1 << 32:
            hit.append(i)
    return hit


def test_permutation_takes_the_out_of_line_carry_of_the_linear_layers():
    """The linear layers fold (low-plane sum, high-plane sum) into one word with ONE add on the high word; where that add wraps
    (about 2^-25 per word in an external layer on random data) a wave-uniform branch adds 2^64 mod p out of line.  States are
    built so that the first layer wraps in chosen words — all of them, single ones, lone lanes of a wave — and the model above
    confirms they do; the partial rounds take their stubs on random data (2^-18 per word) in every large test."""
    P = O.P
    rc0 = [int(x) % P for x in O.poseidon_round_constants()[0]]
    rng = np.random.default_rng(11)
    M = _ext_matrix()
    states = []
    for rep in range(8):                                    # for every output word i: high halves solved so that B_i.lo = 2^32 - 1 - small
        for i in range(12):
            hi = [int(rng.integers(0, 1 << 32)) for _ in range(12)]
            lo = [0xFFFFFFFF if rep % 2 == 0 else int(rng.integers(1 << 31, 1 << 32)) for _ in range(12)]
            js = next(j for j in range(12) if M[i][j] % 2 == 1)            # an odd entry is invertible mod 2^32
            rest = sum(M[i][j] * hi[j] for j in range(12) if j != js) + (rc0[i] >> 32)
            want_lo = (0xFFFFFFFF - int(rng.integers(0, 8))) & 0xFFFFFFFF
            hi[js] = (want_lo - rest) * pow(M[i][js], -1, 1 << 32) % (1 << 32)
            st = [(h << 32) | l for h, l in zip(hi, lo)]
            assert i in _ext_layer_carry_words(st, rc0)
            states.append(st)
    words = set()
    for st in states:
        words.update(_ext_layer_carry_words(st, rc0))
    assert len(words) == 12                                 # every word's stub is reached by some state
    st = np.array(states, dtype=np.uint64)
    want = np.stack([O.poseidon2_permutation(np.array([x % P for x in s], dtype=np.uint64)) for s in states])
    filler = rng.integers(0, P, size=(64 * 40, 12), dtype=np.uint64)
    want_filler = np.stack([O.poseidon2_permutation(s) for s in filler])
    for reps in (1, 33):
        batch = np.tile(st, (reps, 1))
        d = DevBuf(batch)
        ctx().poseidon2_permute(d.ptr, batch.shape[0])
        assert np.array_equal(d.get(batch.shape), np.tile(want, (reps, 1)))
        d.free()
    lone = filler.copy()                                    # one wrapping lane in otherwise random waves
    expect = want_filler.copy()
    for w in range(0, 40, 3):
        lone[64 * w + (7 * w) % 64] = st[(5 * w) % len(states)]
        expect[64 * w + (7 * w) % 64] = want[(5 * w) % len(states)]
    d = DevBuf(lone)
    ctx().poseidon2_permute(d.ptr, lone.shape[0])
    assert np.array_equal(d.get(lone.shape), expect)
    d.free()
