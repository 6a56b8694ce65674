"""The generated Poseidon2 instruction stream (tools/gen_p2_asm.py -> csrc/p2_asm.inc) checked on the CPU: a single-lane
emulator of the gfx950 instructions it uses (tools/p2_emulate.py) runs the stream — the default schedule and every generator
option — and the result must be the oracle's permutation, on random states, on states that take the out-of-line paths of the
products (borrow without carry) and of the folded sums (second carry), and on non-canonical input words.  The emulator also
counts the VALU instructions a permutation executes: the figure the PMC profile measures on the GPU
(profiles/r03_pmc_bench_2p22_leaf_traffic.json: SQ_INSTS_VALU per permutation of the leaf kernel = stream + the kernel's own
loads / absorption / canonicalisation)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import p2_emulate as EM   # noqa: E402

try:
    G = importlib.import_module("test_gpu_poseidon2")           # the state builders of the GPU test of the same paths
except ImportError:
    G = importlib.import_module("tests.test_gpu_poseidon2")
P = O.P

VARIANTS = {
    "default": None,
    "ways2": {"BJ_P2_WAYS": "2"},
    "ways4": {"BJ_P2_WAYS": "4"},
    "combine_inline": {"BJ_P2_COMBINE": "inline"},
    "zero_hoist_only": {"BJ_P2_ZERO_HOIST": "1", "BJ_P2_LATE_CONST": "0"},
    "round3_stream": {"BJ_P2_ZERO_HOIST": "0", "BJ_P2_LATE_CONST": "0"},          # the defaults until round 4
}


def _want(state):
    return [int(x) for x in O.poseidon2_permutation(np.array([int(w) % P for w in state], dtype=np.uint64))]


def _rare_product_states():
    """First-round S-box inputs forced to 2^48 (x * x = 2^96 = -1: borrow without carry) and to the rare-vector fixture."""
    rc0 = [int(x) % P for x in O.poseidon_round_constants()[0]]
    M = G._ext_matrix()
    with open(os.path.join(ROOT, "tests", "golden", "gl_mul_rare.json")) as f:
        rare = [v["a"] for v in json.load(f)["vectors"]]
    rng = np.random.default_rng(3)
    states = []
    for t in range(20):
        target = [int(rng.integers(0, P, dtype=np.uint64)) for _ in range(12)]
        if t == 0:
            target = [1 << 48] * 12
        elif t < 13:
            target[t - 1] = 1 << 48
        else:
            for k in range(12):
                if rng.random() < 0.5:
                    target[k] = [1 << 48, (1 << 48) * 3 % P, rare[int(rng.integers(0, len(rare)))] % P][int(rng.integers(0, 3))]
        states.append(G._solve_mod_p(M, [(x - r) % P for x, r in zip(target, rc0)]))
    return states


def _carry_states():
    """High halves solved so that the fold of the first external layer wraps in a chosen output word."""
    rc0 = [int(x) % P for x in O.poseidon_round_constants()[0]]
    M = G._ext_matrix()
    rng = np.random.default_rng(11)
    states = []
    for rep in range(2):
        for i in range(12):
            hi = [int(rng.integers(0, 1 << 32)) for _ in range(12)]
            lo = [0xFFFFFFFF if rep % 2 == 0 else int(rng.integers(1 << 31, 1 << 32)) for _ in range(12)]
            js = next(j for j in range(12) if M[i][j] % 2 == 1)
            rest = sum(M[i][j] * hi[j] for j in range(12) if j != js) + (rc0[i] >> 32)
            want_lo = (0xFFFFFFFF - int(rng.integers(0, 8))) & 0xFFFFFFFF
            hi[js] = (want_lo - rest) * pow(M[i][js], -1, 1 << 32) % (1 << 32)
            st = [(h << 32) | l for h, l in zip(hi, lo)]
            assert i in G._ext_layer_carry_words(st, rc0)
            states.append(st)
    return states


@pytest.fixture(scope="module")
def emulators():
    return {name: EM.build(env) for name, env in VARIANTS.items()}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_stream_equals_the_oracle_permutation(emulators, name):
    e = emulators[name]
    rng = np.random.default_rng(5)
    states = [[int(x) for x in rng.integers(0, P, size=12, dtype=np.uint64)] for _ in range(12)]
    states.append([0] * 12)
    states.append([P - 1] * 12)
    states.append([(1 << 64) - 1] * 12)                                 # non-canonical words in: any u64 is a valid input
    states.append([P + k for k in range(12)])
    states.append([int(x) for x in rng.integers(0, 1 << 64, size=12, dtype=np.uint64)])
    for st in states:
        got = e.run(st)
        assert [g % P for g in got] == _want(st), (name, st)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_stream_takes_its_out_of_line_paths_and_stays_exact(emulators, name):
    e = emulators[name]
    product_stubs = carry_stubs = 0
    for st in _rare_product_states():
        got = e.run(st)
        product_stubs += e.counts["stub_entries"]
        assert [g % P for g in got] == _want(st), (name, "product", st)
    assert product_stubs >= 12 + 12            # all twelve words at once, then one word at a time: every chain's stub is entered
    if VARIANTS[name] and VARIANTS[name].get("BJ_P2_COMBINE") == "inline":
        return                                 # the in-line fold has no out-of-line carry
    for st in _carry_states():
        got = e.run(st)
        carry_stubs += e.counts["stub_entries"]
        assert [g % P for g in got] == _want(st), (name, "carry", st)
    assert carry_stubs >= 24


def test_executed_instruction_counts(emulators):
    """What one permutation costs: the round-3 stream executed 8 682 VALU instructions (the PMC profile of the leaf kernel said
    8 708 per permutation: + its loads, the absorption into the state and the canonicalisation of the digest); the two options
    switched on in round 4 save 93 and a further 26: 8 563."""
    st = [(0x0123456789ABCDEF * (k + 1)) & ((1 << 64) - 1) for k in range(12)]
    counts = {}
    for name, e in emulators.items():
        e.run(st)
        assert e.counts["stub_entries"] == 0
        counts[name] = e.counts["VALU"]
    assert counts["round3_stream"] == 8682
    assert counts["round3_stream"] - counts["zero_hoist_only"] == 93
    assert counts["zero_hoist_only"] - counts["default"] == 26
    for prof, stream in (("r03_pmc_bench_2p22_leaf_traffic.json", "round3_stream"), ("r04_pmc_bench_2p22_leaf_traffic.json", "default")):
        prof = os.path.join(ROOT, "profiles", prof)
        if os.path.exists(prof):
            valu = json.load(open(prof))["valu"]["SQ_INSTS_VALU_mean"]
            per_perm = valu * 64 / ((1 << 25) * 12)                    # wave instructions x 64 lanes / permutations of one launch
            assert 0 <= per_perm - counts[stream] < 60                 # the kernel's own instructions around twelve permutations
