"""-m gpu: the Blake2s-256 flavour of the hashing layer (the reference's non-recursive configuration: tree hasher
`blake2::Blake2s256`, src/cs/oracle/mod.rs:179-245, + `Blake2sTranscript`, transcript.rs:155-262) against an oracle whose
hash is python's hashlib.blake2s — an implementation nobody here wrote."""
import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import proof_format, synthetic as S
from gpu_util import DevBuf, ctx, rand_gl
from oracle import blake as B
from oracle import prover as OP
from oracle import verifier as OV

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _blake_hasher():
    ctx().set_tree_hasher(2)
    yield
    ctx().set_tree_hasher(1)


@pytest.mark.parametrize("n_cols", [1, 5, 7, 8, 9, 16, 17, 93])
@pytest.mark.parametrize("num_leaves,cap", [(64, 16), (256, 1), (1024, 4)])
def test_tree_matches_hashlib(n_cols, num_leaves, cap):
    rng = np.random.default_rng(n_cols * 1000 + num_leaves)
    cols = rand_gl(rng, (n_cols, num_leaves), noncanonical=True)     # words >= p must be hashed as their canonical residue
    d_cols = DevBuf(cols)
    nd = ctx().merkle_tree_digests(num_leaves, cap)
    d_tree = DevBuf(nelems=4 * nd)
    ctx().merkle_tree_build(d_cols.ptr, num_leaves, n_cols, num_leaves, cap, d_tree.ptr)
    got = d_tree.get((nd, 4))
    want = B.merkle_construct(cols, cap)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx().merkle_tree_cap(d_tree.ptr, num_leaves, cap), B.merkle_cap(want, num_leaves, cap))
    for idx in (0, 1, num_leaves // 2 + 1, num_leaves - 1):
        leaf, path = ctx().merkle_tree_proof(d_tree.ptr, num_leaves, cap, idx)
        assert np.array_equal(leaf, B.hash_leaf(cols[:, idx]))
        assert B.merkle_verify(path, B.merkle_cap(want, num_leaves, cap), leaf, idx)
    d_cols.free(); d_tree.free()


@pytest.mark.parametrize("log_e", [1, 2, 3])
def test_chunked_tree_matches_hashlib(log_e):
    rng = np.random.default_rng(log_e)
    n, cap = 2048, 4
    c0, c1 = rand_gl(rng, (n,), noncanonical=True), rand_gl(rng, (n,))
    d0, d1 = DevBuf(c0), DevBuf(c1)
    leaves = n >> log_e
    nd = ctx().merkle_tree_digests(leaves, cap)
    d_tree = DevBuf(nelems=4 * nd)
    ctx().merkle_tree_build_chunked(d0.ptr, d1.ptr, n, log_e, cap, d_tree.ptr)
    assert np.array_equal(d_tree.get((nd, 4)), B.merkle_construct_chunked([c0, c1], 1 << log_e, cap))


@pytest.mark.parametrize("log_n,fri_lde,cap,sec", [(8, 8, 16, 20), (10, 4, 8, 30)])
def test_blake2s_proof_equals_oracle_proof(log_n, fri_lde, cap, sec):
    """Whole proof with Blake2s trees + Blake2s transcript == the oracle prover's (whose hashing is hashlib's)."""
    c = S.sha_shaped_circuit(log_n, seed=50 + log_n, table_bits=2)
    osetup = OP.Setup(c, fri_lde, cap, threads=4, hasher=2)
    po = OP.prove(c, osetup, fri_lde, cap, security_level=sec, threads=4, transcript_kind=3)
    gsetup = E.ProverSetup(ctx(), c, fri_lde, cap, sec, transcript="blake2s")
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=sec)
    for k in ("public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega",
              "values_at_0", "fri_base_oracle_cap", "fri_intermediate_oracles_caps", "final_fri_monomials"):
        assert pg[k] == po[k], k
    assert pg["queries_per_fri_repetition"] == po["queries_per_fri_repetition"]
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), fri_lde, cap), pg, verbose=True, transcript_kind=3)
    gsetup.close()


def test_cfg1_geometry_2p16_blake2s_proves_and_verifies():
    """BASELINE config 1 (sha256_bench_non_recursive.sh): trace 2^16, LDE 8, cap 16, security 100, Blake2s tree + transcript."""
    c = S.sha_shaped_circuit(16, seed=42, table_bits=4)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 100, transcript="blake2s")
    buf, stages = gsetup.prove()
    pg = proof_format.parse(buf, security_level=100)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True, transcript_kind=3)
    assert not OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, transcript_kind=1)
    gsetup.close()


def test_hasher_and_transcript_must_match():
    """Transcript::CompatibleCap must be the TreeHasher::Output: Blake2s transcript + Poseidon2 hasher is refused."""
    c = S.sha_shaped_circuit(8, seed=1, table_bits=2)
    orig = E.binding._ProofConfig
    try:
        E.binding._ProofConfig = lambda *a: orig(a[0], a[1], a[2], a[3], 3, 1)
        with pytest.raises(E.BoojumHipError):
            E.ProverSetup(ctx(), c, 8, 16, 20, transcript="blake2s")
    finally:
        E.binding._ProofConfig = orig
