"""-m gpu: Keccak-256 as tree hasher (`impl TreeHasher<F> for sha3::Keccak256`, src/cs/oracle/mod.rs:247-312) with the
Keccak256Transcript (transcript.rs:264-372).  Oracle: oracle/keccak.py — a numpy restatement of the sponge that is pinned
through hashlib (with the FIPS-202 domain byte it must equal hashlib.sha3_256, tests/test_keccak.py)."""
import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import proof_format, synthetic as S
from gpu_util import DevBuf, ctx, rand_gl
from oracle import keccak as K
from oracle import prover as OP
from oracle import verifier as OV

pytestmark = pytest.mark.gpu
L = K.layer()


@pytest.fixture(autouse=True)
def _keccak_hasher():
    ctx().set_tree_hasher(3)
    yield
    ctx().set_tree_hasher(1)


@pytest.mark.parametrize("n_cols", [1, 8, 16, 17, 18, 33, 34, 35, 93])
@pytest.mark.parametrize("num_leaves,cap", [(64, 16), (512, 2)])
def test_tree_matches_oracle(n_cols, num_leaves, cap):
    rng = np.random.default_rng(n_cols * 1000 + num_leaves)
    cols = rand_gl(rng, (n_cols, num_leaves), noncanonical=True)
    d_cols = DevBuf(cols)
    nd = ctx().merkle_tree_digests(num_leaves, cap)
    d_tree = DevBuf(nelems=4 * nd)
    ctx().merkle_tree_build(d_cols.ptr, num_leaves, n_cols, num_leaves, cap, d_tree.ptr)
    want = L.merkle_construct(cols, cap)
    assert np.array_equal(d_tree.get((nd, 4)), want)
    leaf, path = ctx().merkle_tree_proof(d_tree.ptr, num_leaves, cap, num_leaves - 3)
    assert L.merkle_verify(path, L.merkle_cap(want, num_leaves, cap), leaf, num_leaves - 3)
    d_cols.free(); d_tree.free()


@pytest.mark.parametrize("log_e", [1, 2, 3])
def test_chunked_tree_matches_oracle(log_e):
    rng = np.random.default_rng(log_e)
    n, cap = 2048, 4
    c0, c1 = rand_gl(rng, (n,), noncanonical=True), rand_gl(rng, (n,))
    d0, d1 = DevBuf(c0), DevBuf(c1)
    leaves = n >> log_e
    nd = ctx().merkle_tree_digests(leaves, cap)
    d_tree = DevBuf(nelems=4 * nd)
    ctx().merkle_tree_build_chunked(d0.ptr, d1.ptr, n, log_e, cap, d_tree.ptr)
    assert np.array_equal(d_tree.get((nd, 4)), L.merkle_construct_chunked([c0, c1], 1 << log_e, cap))


def test_keccak_proof_equals_oracle_proof():
    c = S.sha_shaped_circuit(9, seed=71, table_bits=2)
    osetup = OP.Setup(c, 8, 16, threads=4, hasher=3)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=4, transcript_kind=4)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30, transcript="keccak256")
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    for k in ("public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega",
              "values_at_0", "fri_base_oracle_cap", "fri_intermediate_oracles_caps", "final_fri_monomials",
              "queries_per_fri_repetition"):
        assert pg[k] == po[k], k
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True, transcript_kind=4)
    assert not OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, transcript_kind=3)
    gsetup.close()
