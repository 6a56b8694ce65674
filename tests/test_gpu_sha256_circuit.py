"""-m gpu: the HIP prover on the REAL SHA-256 bench circuit (era_boojum_amd/sha256_circuit.py): the proof equals the
oracle prover's and the verifier restatement accepts it; at the reference bench's own size (8 KiB of input, 2^16 rows,
BASELINE config 1) the proof is checked by the verifier restatement."""
import hashlib

import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import proof_format
from era_boojum_amd import sha256_circuit as S
from gpu_util import ctx, oracle_threads
from oracle import prover as OP
from oracle import verifier as OV
from test_gpu_prover import _compare

pytestmark = pytest.mark.gpu


def test_hip_proof_of_real_sha256_equals_oracle_proof():
    msg = S.bench_message(100, seed=7)
    c, info = S.sha256_circuit(msg, return_info=True)
    assert info["digest"] == hashlib.sha256(msg).digest() and c.log_n == 14
    osetup = OP.Setup(c, 8, 16, threads=8)
    po = OP.prove(c, osetup, 8, 16, security_level=40, threads=8)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 40)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=40)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    gsetup.close()


@pytest.mark.parametrize("transcript", ["poseidon2", "poseidon", "blake2s"])
def test_hip_proof_of_the_8_kib_bench_circuit_verifies(transcript):
    """`prove_sha256(8 * (1 << 10))` with the bench's parameters: LDE 8, cap 16, security 100, no PoW
    (sha256/mod.rs:294, 309-316); "poseidon" is the recursive-mode bench script's transcript, "blake2s" (tree hasher and
    transcript) is `run_sha256_prover_non_recursive` = BASELINE config 1 (sha256/mod.rs:265-270)."""
    c = S.sha256_circuit(S.bench_message(8 << 10))
    assert c.log_n == 16
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 100, transcript=transcript)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=100)
    kind = {"poseidon2": 1, "poseidon": 2, "blake2s": 3}[transcript]
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, transcript_kind=kind)
    # a witness with one wrong bit of the message is rejected by the prover's own satisfiability check
    bad = c.variables.copy()
    r = int(np.flatnonzero(c.constants[0] == 1)[0])
    bad[3, r] = (int(bad[3, r]) + 1) % E.P
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        gsetup.prove(variables=bad)
    gsetup.close()


@pytest.mark.parametrize("transcript,kind", [("poseidon2", 1), ("poseidon", 2), ("blake2s", 3)])
def test_hip_proof_of_the_8_kib_bench_circuit_equals_oracle_proof(transcript, kind):
    """BASELINE config 1's circuit (8 KiB, 2^16 rows, LDE 8, cap 16, security 100) under both transcripts the bench offers
    and under cfg1's exact pairing — Blake2s tree hasher + Blake2s transcript, `run_sha256_prover_non_recursive`
    (sha256/mod.rs:265-270): every cap, opening, FRI layer and query of the HIP proof equals the oracle prover's (not only
    "verifier accepts")."""
    c = S.sha256_circuit(S.bench_message(8 << 10))
    assert c.log_n == 16
    osetup = OP.Setup(c, 8, 16, threads=32, hasher=2 if kind == 3 else 1)
    po = OP.prove(c, osetup, 8, 16, security_level=100, threads=32, transcript_kind=kind)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 100, transcript=transcript)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    _compare(proof_format.parse(buf, security_level=100), po)
    gsetup.close()


def test_hip_proof_at_2p18_rows_equals_oracle_proof():
    """The same identity four times larger (SHA-256 of 34 KB, 2^18 rows, the bench's parameters): the 3-pass NTT sizes, the
    multi-workgroup scans of stage 2 and the large-tree kernels are all on this path.  ~1 minute of oracle time."""
    c = S.sha256_circuit(S.bench_message(34000, seed=11))
    assert c.log_n == 18
    osetup = OP.Setup(c, 8, 16, threads=oracle_threads())
    po = OP.prove(c, osetup, 8, 16, security_level=100, threads=oracle_threads(), transcript_kind=1)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 100, transcript="poseidon2")
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=100)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg)
    gsetup.close()


def test_hip_proof_at_2p20_rows_equals_oracle_proof_under_both_bench_transcripts():
    """BASELINE config 3 (cfg3): the real SHA-256 circuit at 2^20 rows (SHA-256 of ~139 kB), the bench's parameters (LDE 8, cap
    16, security 100, no PoW), full prove incl. Poseidon2 Merkle trees and FRI: the HIP proof equals the oracle prover's proof
    — every cap, opening, FRI layer, final monomial and all query openings — under the golden-pinned Poseidon2 transcript and
    under the bench script's Poseidon one (prover.rs:153-168 is the function replaced).  ~1.5 minutes of oracle time."""
    c = S.sha256_circuit(S.bench_message(S.message_len_for_log_n(20), seed=13))
    assert c.log_n == 20
    osetup = OP.Setup(c, 8, 16, threads=oracle_threads())
    for transcript, kind in (("poseidon2", 1), ("poseidon", 2)):
        po = OP.prove(c, osetup, 8, 16, security_level=100, threads=oracle_threads(), transcript_kind=kind)
        gsetup = E.ProverSetup(ctx(), c, 8, 16, 100, transcript=transcript)
        assert np.array_equal(gsetup.cap(), osetup.cap)
        buf, _ = gsetup.prove()
        pg = proof_format.parse(buf, security_level=100)
        _compare(pg, po)
        if kind == 1:
            assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg)
        gsetup.close()
        del po, pg


def _host_ram_gb():
    """Memory this process may actually use: MemAvailable, capped by the container's cgroup limit where there is one."""
    avail = 0.0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) / 1e6
    except OSError:
        return 0.0
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            text = open(path).read().strip()
        except OSError:
            continue
        if text.isdigit():
            avail = min(avail, int(text) / 1e9)
    return avail


@pytest.mark.streaming(140)
def test_hip_proof_at_2p22_rows_equals_the_oracle():
    """BASELINE config 4's size, the bench's own workload (cfg4): the real SHA-256 circuit at 2^22 rows (SHA-256 of 557 kB,
    bench parameters: LDE 8, cap 16, security 100, Poseidon2 tree + golden-pinned Poseidon2 transcript).
    With >= 500 GB of host memory (the MI355X boxes have 3 TB) the FULL oracle prover runs — ~250 GB of LDEs, ~6 minutes — and the
    HIP proof must equal its proof byte for byte: every cap, opening, FRI layer, final monomial and all 34 x 4 query openings
    with their paths (prover.rs:153-168 is the function replaced).  With less memory the coset-streaming restatement
    (oracle/prover_streaming.py, checked against the full prover on CPU) recomputes EVERY transcript input one coset at a time:
    the three oracle caps over 2^25 leaves each, all 241 values at z, z*omega and 0, all eight cosets of the setup oracle against the
    verification key's cap (round 6: the oracle hashes eight leaves per AVX-512 call, oracle/poseidon2_avx512.c), then (round 5)
    the DEEP accumulator of all 2^25 points from a second pass over the cosets, the FRI base oracle, every intermediate oracle
    and fold challenge, the final monomials, the query indices and every query opening (prover.rs:1803-2266, fri/mod.rs:49-345);
    all of it equals the HIP proof's, the four base-oracle openings of every query included."""
    c = S.sha256_circuit(S.bench_message(S.message_len_for_log_n(22)))
    assert c.log_n == 22
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 100, transcript="poseidon2")
    buf, _ = gsetup.prove()
    cap = gsetup.cap()
    gsetup.close()
    ctx().release_workspace()
    pg = proof_format.parse(buf, security_level=100)
    if _host_ram_gb() >= 500:
        osetup = OP.Setup(c, 8, 16, threads=oracle_threads())
        assert np.array_equal(cap, osetup.cap)
        po = OP.prove(c, osetup, 8, 16, security_level=100, threads=oracle_threads(), transcript_kind=1)
        _compare(pg, po)
        return
    from oracle import prover_streaming as PS
    po = PS.commitments_and_openings(c, cap, 8, 16, threads=oracle_threads(), transcript_kind=1, check_setup_cosets=tuple(range(8)), rest_of_the_proof=True,
                                     security_level=100)
    for k in ("public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega",
              "values_at_0"):
        assert pg[k] == po[k], k
    for cs, frag in po["setup_cap_fragments"].items():
        assert np.array_equal(frag, cap[2 * cs:2 * cs + 2]), "setup cap nodes of coset %d" % cs
    # every transcript input after the openings, and the queries: FRI base cap, every intermediate cap, final monomials, all
    # 34 x 7 FRI query openings with their paths byte for byte; witness / second-stage / quotient / setup opening (through the leaf
    # hash of the restatement's own tree) + path of all 34 queries
    compared = PS.compare_rest_of_the_proof(pg, po)
    nq = len(pg["queries_per_fri_repetition"])
    assert nq == 34 and compared == 4 * nq
    assert OV.verify(OV.VerificationKey(c, cap, 8, 16), pg)


def test_prove_from_memcopy_dumps():
    """A Rust host's `SetupBaseStorage` / `WitnessVec` / `DenseVariablesCopyHint` dumps (era_boojum_amd/memcopy_format.py)
    give the same proof as the in-memory circuit."""
    from era_boojum_amd import memcopy_format as M
    from era_boojum_amd.synthetic import sha_bench_gates
    c, info = S.sha256_circuit(S.bench_message(33, seed=3), return_info=True)
    total = sum(t.shape[0] for t in S.sha_tables())
    back = M.circuit_from_dumps(M.write_setup_base(c),
                                M.write_witness_vec([], info["all_values"], c.multiplicities[0, :total].astype(np.uint32)),
                                M.write_variables_hint(info["var_ids"]), sha_bench_gates(), num_gp_vars=60, lookup_width=4,
                                lookup_reps=8)
    a = E.ProverSetup(ctx(), c, 8, 16, 30)
    b = E.ProverSetup(ctx(), back, 8, 16, 30)
    assert np.array_equal(a.cap(), b.cap())
    pa, _ = a.prove()
    pb, _ = b.prove()
    assert np.array_equal(pa, pb)
    assert OV.verify(OV.VerificationKey(back, b.cap(), 8, 16), proof_format.parse(pb, security_level=30))
    a.close(); b.close()


def test_dumps_through_the_c_abi():
    """The same dumps straight into the library (bj_setup_create_from_dump / bj_prove_from_dumps, csrc/dumps.hip): the C++ reader
    takes columns, selector paths (bincode TreeNode), table-id column and quotient degree from the SetupBaseStorage bytes,
    computes the non-residues itself, materialises the cells from WitnessVec + DenseVariablesCopyHint on the device — and
    produces the proof of the in-memory circuit, byte for byte.  Malformed dumps are refused with a message."""
    import copy
    from era_boojum_amd import memcopy_format as M
    c, info = S.sha256_circuit(S.bench_message(700, seed=5), return_info=True)
    total = sum(t.shape[0] for t in S.sha_tables())
    setup_dump = M.write_setup_base(c)
    wit_dump = M.write_witness_vec([(col, row) for col, row, _ in c.public_inputs], info["all_values"],
                                   c.multiplicities[0, :total].astype(np.uint32))
    hint_dump = M.write_variables_hint(info["var_ids"])
    a = E.ProverSetup(ctx(), c, 8, 16, 30)
    pa, _ = a.prove()
    bare = copy.copy(c)                  # what a Rust host has as CODE: geometry + gate list; no paths, no degree, no non-residues
    bare.gates = [copy.copy(g) for g in c.gates]
    for g in bare.gates:
        g.path = []
    b = E.ProverSetup(ctx(), bare, 8, 16, 30, setup_base_dump=setup_dump)
    assert np.array_equal(a.cap(), b.cap())
    pb, _ = b.prove_from_dumps(wit_dump, hint_dump)
    assert np.array_equal(pa, pb)
    for bad, what in ((setup_dump[:-3], "truncated"), (setup_dump + b"\0", "trailing"), (setup_dump[:1000], "truncated")):
        with pytest.raises(E.BoojumHipError, match=what):
            E.ProverSetup(ctx(), bare, 8, 16, 30, setup_base_dump=bad)
    with pytest.raises(E.BoojumHipError, match="WitnessVec"):
        b.prove_from_dumps(wit_dump[:-2], hint_dump)
    with pytest.raises(E.BoojumHipError, match="CopyHint"):
        b.prove_from_dumps(wit_dump, hint_dump[:-8])
    broken = bytearray(hint_dump)
    broken[16:24] = (1 << 40).to_bytes(8, "little")      # a cell naming a variable beyond all_values
    with pytest.raises(E.BoojumHipError, match="beyond all_values"):
        b.prove_from_dumps(wit_dump, bytes(broken))
    a.close(); b.close()
