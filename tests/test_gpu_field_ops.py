"""-m gpu parity, SURVEY §8 rows a1/a2 at operator level: bj_field_op_batch (GoldilocksField add / sub / mul / square / inverse,
F_p^2 product; src/field/goldilocks/mod.rs:188-255, 294-360, src/field/traits/field.rs:407-512) against Python integers —
including the operand pairs that take the rare (2^-32) branches of the device multiplication."""
import json
import os

import numpy as np
import pytest

from gpu_util import ctx, rand_gl

pytestmark = pytest.mark.gpu
P = (1 << 64) - (1 << 32) + 1
HERE = os.path.dirname(os.path.abspath(__file__))

SPECIALS = [0, 1, 2, 7, (1 << 16), (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 48), (1 << 48) - 1, P - 2, P - 1, P, P + 1,
            (1 << 64) - (1 << 32), (1 << 64) - 2, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF,
            0xFFFFFFFE00000001, 0x8000000080000000, 0x00000001FFFFFFFF]


def _pairs():
    rng = np.random.default_rng(5)
    a = [x for x in SPECIALS for _ in SPECIALS]
    b = [y for _ in SPECIALS for y in SPECIALS]
    # a * b = k * 2^96 = -k: the reduction's subtraction borrows with no carry before it; powers of two in every position
    for i in range(0, 64, 3):
        for j in range(0, 64, 5):
            for k in (1, 3, 0xFFFF):
                a.append((1 << i) % (1 << 64))
                b.append((k << j) % (1 << 64))
    with open(os.path.join(HERE, "golden", "gl_mul_rare.json")) as f:
        for v in json.load(f)["vectors"]:
            a.append(v["a"])
            b.append(v["b"])
            assert v["a"] * v["b"] % P == v["product"]
    ra = rand_gl(rng, (1 << 16,), noncanonical=True)
    rb = rand_gl(rng, (1 << 16,), noncanonical=True)
    a = np.concatenate([np.array(a, dtype=np.uint64), ra])
    b = np.concatenate([np.array(b, dtype=np.uint64), rb])
    return a, b


@pytest.mark.parametrize("op", ["add", "sub", "mul", "mul_lazy", "square"])
def test_field_operator_matches_python_integers(op):
    a, b = _pairs()
    got = ctx().field_op(op, a, None if op == "square" else b)
    fn = {"add": lambda x, y: (x + y) % P, "sub": lambda x, y: (x - y) % P, "mul": lambda x, y: x * y % P,
          "mul_lazy": lambda x, y: x * y % P, "square": lambda x, y: x * x % P}[op]
    want = np.array([fn(int(x), int(y)) for x, y in zip(a, b)], dtype=np.uint64)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at %d: a=%d b=%d got=%d want=%d" % (bad[0], a[bad[0]], b[bad[0]], got[bad[0]], want[bad[0]])


def test_rare_branch_vectors_on_a_lone_wave_and_on_a_full_grid():
    """The wave-uniform rare branch must work whether one lane of one wave takes it or every lane does."""
    with open(os.path.join(HERE, "golden", "gl_mul_rare.json")) as f:
        vec = json.load(f)["vectors"]
    for n in (1, 63, 64, 65, 1 << 14):
        a = np.array([vec[i % len(vec)]["a"] for i in range(n)], dtype=np.uint64)
        b = np.array([vec[i % len(vec)]["b"] for i in range(n)], dtype=np.uint64)
        want = np.array([vec[i % len(vec)]["product"] for i in range(n)], dtype=np.uint64)
        for op in ("mul", "mul_lazy"):
            assert np.array_equal(ctx().field_op(op, a, b), want), (op, n)
    # one rare pair inside an otherwise random wave
    rng = np.random.default_rng(9)
    a = rand_gl(rng, (4096,))
    b = rand_gl(rng, (4096,))
    for k, v in enumerate(vec):
        a[37 + 64 * k] = v["a"]
        b[37 + 64 * k] = v["b"]
    want = np.array([int(x) * int(y) % P for x, y in zip(a, b)], dtype=np.uint64)
    assert np.array_equal(ctx().field_op("mul", a, b), want)


def test_inverse_and_extension_product():
    rng = np.random.default_rng(6)
    a = rand_gl(rng, (2048,), noncanonical=True)
    a[:4] = [0, 1, P - 1, P]
    got = ctx().field_op("inverse", a)
    want = np.array([pow(int(x) % P, P - 2, P) for x in a], dtype=np.uint64)
    assert np.array_equal(got, want)
    x = rand_gl(rng, (2, 1024), noncanonical=True)
    y = rand_gl(rng, (2, 1024), noncanonical=True)
    got = ctx().field_op("ext2_mul", x, y)
    w0 = [(int(p) * int(r) + 7 * int(q) * int(s)) % P for p, q, r, s in zip(x[0], x[1], y[0], y[1])]
    w1 = [(int(p) * int(s) + int(q) * int(r)) % P for p, q, r, s in zip(x[0], x[1], y[0], y[1])]
    assert np.array_equal(got, np.array([w0, w1], dtype=np.uint64))


def _butterfly_inputs():
    """u, v, w triples: the product's rare branches (v, w from the fixture), sums and differences that wrap twice (operands
    next to 2^64), boundary words, random weak values."""
    with open(os.path.join(HERE, "golden", "gl_mul_rare.json")) as f:
        vec = json.load(f)["vectors"]
    big = [(1 << 64) - 1, (1 << 64) - 2, (1 << 64) - (1 << 32), (1 << 64) - (1 << 32) + 1, P - 1, P, P + 1, 0, 1, (1 << 32) - 1, 1 << 32]
    u, v, w = [], [], []
    for x in vec:                      # rare products, with u at the edges
        for uu in (0, (1 << 64) - 1, P - 1, x["product"], (P - x["product"]) % P):
            u.append(uu); v.append(x["a"]); w.append(x["b"])
    for uu in big:                     # v * 1 = v next to 2^64: the sum / difference wrap twice
        for vv in big:
            u.append(uu); v.append(vv); w.append(1)
            u.append(uu); v.append(vv); w.append(P + 1 if vv < (1 << 32) - 2 else 1)
    rng = np.random.default_rng(12)
    k = (1 << 15) - (len(u) % (1 << 15))
    u += [int(x) for x in rng.integers(0, 1 << 63, size=k, dtype=np.uint64) * 2 + rng.integers(0, 2, size=k, dtype=np.uint64)]
    v += [int(x) for x in rng.integers(0, 1 << 63, size=k, dtype=np.uint64) * 2 + 1]
    w += [int(x) for x in rand_gl(rng, (k,), noncanonical=True)]
    if len(u) % 2:
        u.append(0); v.append(0); w.append(0)
    return np.array(u, dtype=np.uint64), np.array(v, dtype=np.uint64), np.array(w, dtype=np.uint64)


def test_lazy_ntt_butterflies_match_python_integers():
    u, v, w = _butterfly_inputs()
    n = u.size
    got = ctx().field_op("butterfly", np.stack([u, v]), w)
    want_s = np.array([(int(a) + int(b) * int(c)) % P for a, b, c in zip(u, v, w)], dtype=np.uint64)
    want_d = np.array([(int(a) - int(b) * int(c)) % P for a, b, c in zip(u, v, w)], dtype=np.uint64)
    assert np.array_equal(got[0], want_s) and np.array_equal(got[1], want_d)
    got = ctx().field_op("addsub", np.stack([u, v]))
    assert np.array_equal(got[0], np.array([(int(a) + int(b)) % P for a, b in zip(u, v)], dtype=np.uint64))
    assert np.array_equal(got[1], np.array([(int(a) - int(b)) % P for a, b in zip(u, v)], dtype=np.uint64))
    # a lone wave whose only work is a rare case (the wave-uniform fallback must trigger on one lane's flag)
    for i in (0, 5, 57):
        uu, vv, ww = np.full(2, u[i]), np.full(2, v[i]), np.full(2, w[i])
        g = ctx().field_op("butterfly", np.stack([uu, vv]), ww)
        assert int(g[0][0]) == (int(u[i]) + int(v[i]) * int(w[i])) % P and int(g[1][1]) == (int(u[i]) - int(v[i]) * int(w[i])) % P


def test_weak_sum_difference_and_extension_product_on_extreme_words():
    """gl::add_weak / sub_weak / e2_mul_weak (the links of the copy-permutation quotient's product chains) take ANY u64 and never
    canonicalise in between: all pairs of the words around 0, 2^32, p and 2^64 — where the first and the second wrap of a sum or
    a borrow of a difference happen — and random words, against Python integers."""
    edge = [0, 1, 2, (1 << 32) - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - 2, P - 1, P, P + 1, (1 << 64) - (1 << 32) - 1,
            (1 << 64) - (1 << 32), (1 << 64) - (1 << 32) + 2, (1 << 64) - 2, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, 0xFFFFFFFF00000000]
    rng = np.random.default_rng(7)
    a = np.array([x for x in edge for _ in edge] + [int(v) for v in rng.integers(0, 1 << 64, 4096, dtype=np.uint64)], dtype=np.uint64)
    b = np.array([y for _ in edge for y in edge] + [int(v) for v in rng.integers(0, 1 << 64, 4096, dtype=np.uint64)], dtype=np.uint64)
    for op, f in (("add_lazy", lambda x, y: (x + y) % P), ("sub_lazy", lambda x, y: (x - y) % P)):
        got = ctx().field_op(op, a, b)
        assert [int(v) for v in got] == [f(int(x), int(y)) for x, y in zip(a, b)], op
    # F_p^2: (a0, a1) x (b0, b1), halves at +m; every combination of edge words in the four slots would be 19^4: sample them
    m = 6000
    pick = lambda: np.array([edge[int(i)] if k < 0.7 else int(r) for i, k, r in zip(rng.integers(0, len(edge), m), rng.random(m),
                                                                                     rng.integers(0, 1 << 64, m, dtype=np.uint64))], dtype=np.uint64)
    x = np.concatenate([pick(), pick()])
    y = np.concatenate([pick(), pick()])
    got = ctx().field_op("ext2_mul_lazy", x, y).reshape(2, m)
    for i in range(m):
        a0, a1, b0, b1 = int(x[i]), int(x[m + i]), int(y[i]), int(y[m + i])
        assert (int(got[0, i]), int(got[1, i])) == ((a0 * b0 + 7 * a1 * b1) % P, (a0 * b1 + a1 * b0) % P), i
