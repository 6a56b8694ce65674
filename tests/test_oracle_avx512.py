"""The oracle's AVX-512 paths — Poseidon2 (oracle/poseidon2_avx512.c: eight permutations lane-wise) and the NTT butterfly loops
(oracle/ntt_avx512.c) — against their scalar definitions (oracle/poseidon2.c, oracle/ntt.c): permutation, leaf sponge, node layers,
chunked leaves, transforms — on random, non-canonical and edge inputs.  Both
are pinned by the reference's golden proof as well (tests/test_oracle_fixture.py runs under whichever path the CPU selects; the
scalar path is forced in a subprocess below).  Skipped on CPUs without AVX-512."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O

P = O.P
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_avx512 = pytest.mark.skipif(not O.poseidon2_avx512_available(), reason="no AVX-512 on this CPU")


def _states(rng, kind):
    if kind == "random":
        return rng.integers(0, P, size=(8, 12), dtype=np.uint64)
    if kind == "noncanonical":           # representatives in [p, 2^64)
        s = rng.integers(0, 1 << 32, size=(8, 12), dtype=np.uint64)
        return s + np.uint64(P - 1) * (rng.integers(0, 2, size=(8, 12)).astype(np.uint64))
    edge = np.array([0, 1, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 63), P - (1 << 32), 0xFFFFFFFF00000000, 0xFFFFFFFF, 7, P // 2], dtype=np.uint64)
    return np.stack([np.roll(edge, k) for k in range(8)])


@needs_avx512
@pytest.mark.parametrize("kind", ["random", "noncanonical", "edge"])
def test_eight_lane_permutation_equals_the_scalar_one(kind):
    rng = np.random.default_rng(11)
    for _ in range(50 if kind == "random" else 5):
        s = _states(rng, kind)
        want = np.stack([O.poseidon2_permutation(s[k]) for k in range(8)])
        assert np.array_equal(O.poseidon2_permutation_x8(s), want)


@needs_avx512
@pytest.mark.parametrize("n_cols,num_leaves", [(1, 64), (7, 64), (8, 64), (9, 64), (16, 64), (93, 64), (93, 8), (20, 16), (5, 4)])
def test_trees_are_the_same_under_both_paths(n_cols, num_leaves):
    """Leaves (sixteen per call as two interleaved groups, eight per call, the scalar remainder), node layers and the cap of a tree
    whose top layers are narrower than a register: the AVX-512 build of the tree equals the tree made of scalar hash_leaf /
    hash_node calls."""
    rng = np.random.default_rng(100 + n_cols)
    cols = rng.integers(0, P, size=(n_cols, num_leaves), dtype=np.uint64)
    cols[0, :4] = np.array([P, P + 5, 2**64 - 1, 0], dtype=np.uint64)          # non-canonical inputs are legal
    tree = O.merkle_construct(cols, 2, threads=2)
    leaves = np.stack([O.hash_leaf(cols[:, i]) for i in range(num_leaves)])
    assert np.array_equal(tree[:num_leaves], leaves)
    layer, off = leaves, num_leaves
    while layer.shape[0] > 2:
        nxt = np.stack([O.hash_node(layer[2 * i], layer[2 * i + 1]) for i in range(layer.shape[0] // 2)])
        assert np.array_equal(tree[off:off + nxt.shape[0]], nxt)
        off += nxt.shape[0]
        layer = nxt


@needs_avx512
@pytest.mark.parametrize("elems_per_leaf,n_srcs", [(1, 2), (4, 2), (8, 2), (3, 1)])
def test_chunked_leaves_are_the_same_under_both_paths(elems_per_leaf, n_srcs):
    rng = np.random.default_rng(7)
    num_leaves = 32
    srcs = [rng.integers(0, P, size=num_leaves * elems_per_leaf, dtype=np.uint64) for _ in range(n_srcs)]
    tree = O.merkle_construct_chunked(srcs, elems_per_leaf, 4, threads=2)
    for j in (0, 1, 7, 8, 31):
        leaf = np.concatenate([s[j * elems_per_leaf:(j + 1) * elems_per_leaf] for s in srcs])
        assert np.array_equal(tree[j], O.hash_leaf(leaf))


@needs_avx512
def test_the_golden_proof_pins_the_scalar_path_too():
    """tests/test_oracle_fixture.py's Poseidon2 / Merkle / transcript checks and tests/test_oracle_ntt.py (transforms against the
    naive DFT of the reference's own differential tests) once more in a process where ORC_NO_AVX512=1 forces the scalar code: the
    golden vectors and the DFT hold for both paths on this CPU."""
    env = dict(os.environ, ORC_NO_AVX512="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.join(ROOT, "tests", "test_oracle_fixture.py"), "-k",
                        "poseidon2 or merkle or leaf or path or transcript"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.join(ROOT, "tests", "test_oracle_ntt.py")], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@needs_avx512
@pytest.mark.parametrize("log_n", [3, 4, 10, 16])
def test_transforms_are_the_same_under_both_paths(log_n):
    """oracle/ntt_avx512.c (eight butterflies per instruction, vector distribute_powers / scaling) against the scalar loops of
    oracle/ntt.c in a child process: forward coset transform, inverse coset transform and the LDE on random columns with
    non-canonical words."""
    import json
    rng = np.random.default_rng(5 + log_n)
    cols = rng.integers(0, P, size=(3, 1 << log_n), dtype=np.uint64)
    cols[0, :2] = np.array([2**64 - 1, P], dtype=np.uint64)
    got = {"fwd": O.fft_batch(cols, 7, 2), "inv": O.ifft_batch(cols, 7, 2), "plain": O.fft_batch(cols, 1, 2)}
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); import oracle as O; "
            "assert not O.poseidon2_avx512_available(); "
            "c = np.array(json.loads(sys.stdin.read()), dtype=np.uint64); "
            "print(json.dumps({'fwd': O.fft_batch(c, 7, 2).tolist(), 'inv': O.ifft_batch(c, 7, 2).tolist(), 'plain': O.fft_batch(c, 1, 2).tolist()}))") % ROOT
    r = subprocess.run([sys.executable, "-c", code], input=json.dumps(cols.tolist()), env=dict(os.environ, ORC_NO_AVX512="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = json.loads(r.stdout)
    for k in got:
        assert np.array_equal(got[k], np.array(want[k], dtype=np.uint64)), k
