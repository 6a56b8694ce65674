"""Poseidon (v1) — the round function of the bench script's GoldilocksPoisedonTranscript.  The reference holds NO
known-answer vector for it (its tests only compare its own naive and optimised implementations), so parity is
"unpinned": what is checked here is that three restatements written from different parts of the reference agree —
  * this file: python big integers, the explicit matrix MDS_MATRIX[row][col] = 2^EXPS[(12 - row + col) % 12]
    (poseidon_goldilocks_naive.rs:12-35) and FULL/PARTIAL round structure (:120-160),
  * oracle/poseidon2.c `orc_poseidon_permutation`: 128-bit shift-and-add of mds_mul_naive (:66-92),
  * the product's host code (csrc/host_transcript.hpp) through bj_transcript_*.
"""
import numpy as np

import era_boojum_amd as E
import oracle as O

P = O.P
EXPS = [0, 0, 1, 0, 3, 5, 1, 8, 12, 3, 16, 10]
MDS = [[1 << EXPS[(12 - row + col) % 12] for col in range(12)] for row in range(12)]


def poseidon1_python(state):
    rc = O.poseidon_round_constants()
    s = [int(x) % P for x in state]
    for r in range(30):
        s = [(x + c) % P for x, c in zip(s, rc[r])]
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = [sum(m * x for m, x in zip(row, s)) % P for row in MDS]
    return s


def test_c_oracle_matches_big_integer_restatement():
    rng = np.random.default_rng(1)
    states = [np.zeros(12, dtype=np.uint64), np.ones(12, dtype=np.uint64), np.full(12, P - 1, dtype=np.uint64),
              np.full(12, 2**64 - 1, dtype=np.uint64)] + [rng.integers(0, 2**64 - 1, size=12, dtype=np.uint64) for _ in range(20)]
    for st in states:
        assert [int(x) for x in O.poseidon_permutation(st)] == poseidon1_python(st)


def test_matrix_is_circulant_and_permutation_is_not_poseidon2():
    assert all(MDS[r][c] == MDS[0][(c - r) % 12] for r in range(12) for c in range(12))
    st = np.arange(12, dtype=np.uint64)
    assert [int(x) for x in O.poseidon_permutation(st)] != [int(x) for x in O.poseidon2_permutation(st)]


def test_product_transcript_matches_oracle_for_poseidon_v1():
    rng = np.random.default_rng(5)
    tp, to = E.Transcript(kind=2), O.Transcript(kind=2)
    t2 = O.Transcript(kind=1)
    seq = [("c", 9), ("a", 1), ("c", 3), ("a", 8), ("c", 1), ("a", 7), ("c", 17), ("a", 23), ("c", 2)]
    differs = False
    for kind, n in seq:
        if kind == "a":
            els = rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)
            els[0] = np.uint64(2**64 - 1)
            tp.absorb(els); to.absorb(els); t2.absorb(els)
        else:
            a, b, c = [tp.challenge() for _ in range(n)], [to.challenge() for _ in range(n)], [t2.challenge() for _ in range(n)]
            assert a == b
            differs |= a != c
    assert differs
    # the sponge's first challenge after absorbing [5]: one permutation of (5, 1, 0, ..., 0)
    t = E.Transcript(kind=2)
    t.absorb(np.array([5], dtype=np.uint64))
    assert t.challenge() == poseidon1_python([5, 1] + [0] * 10)[0]
