"""-m gpu: the whole HIP prover (bj_setup_create + bj_prove through the C ABI) against the CPU oracle prover on the same
SHA-shaped synthetic circuit: IDENTICAL proof (every cap, opening, FRI layer, query), and the oracle's restatement of
the reference verifier accepts it."""
import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import proof_format, synthetic as S
from gpu_util import ctx, oracle_threads
from oracle import prover as OP
from oracle import verifier as OV

pytestmark = pytest.mark.gpu


def _compare(pg, po):
    for k in ("public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z",
              "values_at_z_omega", "values_at_0", "fri_base_oracle_cap", "fri_intermediate_oracles_caps",
              "final_fri_monomials"):
        assert pg[k] == po[k], k
    assert len(pg["queries_per_fri_repetition"]) == len(po["queries_per_fri_repetition"])
    for qg, qo in zip(pg["queries_per_fri_repetition"], po["queries_per_fri_repetition"]):
        for name in ("witness_query", "stage_2_query", "quotient_query", "setup_query"):
            assert qg[name] == qo[name], name
        assert qg["fri_queries"] == qo["fri_queries"]


@pytest.mark.parametrize("log_n,fri_lde,cap,sec,bits", [(9, 8, 16, 30, 2), (8, 2, 4, 20, 2), (12, 8, 16, 40, 2), (14, 4, 16, 50, 4)])
def test_hip_proof_equals_oracle_proof_and_verifies(log_n, fri_lde, cap, sec, bits):
    c = S.sha_shaped_circuit(log_n, seed=100 + log_n, table_bits=bits)
    osetup = OP.Setup(c, fri_lde, cap, threads=8)
    po = OP.prove(c, osetup, fri_lde, cap, security_level=sec, threads=8)
    gsetup = E.ProverSetup(ctx(), c, fri_lde, cap, sec)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, stage_ms = gsetup.prove()
    pg = proof_format.parse(buf, security_level=sec)
    _compare(pg, po)
    vk = OV.VerificationKey(c, gsetup.cap(), fri_lde, cap)
    assert OV.verify(vk, pg, verbose=True)
    assert set(stage_ms) == set(E.binding.STAGE_NAMES) | {"witness_tree_leaf_kernel"}
    gsetup.close()


def test_hip_prover_reports_unsatisfied_witness():
    c = S.sha_shaped_circuit(8, seed=5, table_bits=2)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 20)
    rows = np.nonzero(c.constants[0] == 1)[0]
    bad = c.variables.copy()
    bad[3, rows[0]] = (int(bad[3, rows[0]]) + 1) % E.P
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        gsetup.prove(variables=bad)
    gsetup.close()


def test_hip_prover_without_lookups():
    c = S.sha_shaped_circuit(8, seed=9, table_bits=2)
    # strip the lookup argument: keep only the general-purpose part of the circuit
    c.variables = np.ascontiguousarray(c.variables[:c.num_gp_vars]); c.sigmas = np.ascontiguousarray(c.sigmas[:c.num_gp_vars])
    c.non_residues = c.non_residues[:c.num_gp_vars]
    c.constants = np.ascontiguousarray(c.constants[:c.num_constants_for_gates]); c.num_constant_cols = c.num_constants_for_gates
    c.num_lookup_vars = 0; c.lookup_reps = 0
    osetup = OP.Setup(c, 4, 8, threads=4)
    po = OP.prove(c, osetup, 4, 8, security_level=20, threads=4)
    gsetup = E.ProverSetup(ctx(), c, 4, 8, 20)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=20)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 4, 8), pg, verbose=True)
    gsetup.close()


@pytest.mark.parametrize("log_n,kw,fri_lde,cap,sec", [
    (10, dict(num_gp_vars=24, lookup_width=3, lookup_reps=4, num_public_inputs=0), 8, 16, 30),   # fixture-like width-3 lookups, no public inputs
    (11, dict(num_gp_vars=40, lookup_width=4, lookup_reps=2, num_public_inputs=3), 16, 32, 40),  # LDE 16, cap 32
    (9, dict(num_gp_vars=60, lookup_width=3, lookup_reps=8, num_public_inputs=1), 4, 4, 20),     # LDE 4 = quotient degree, cap 4
    (13, dict(num_gp_vars=20, lookup_width=4, lookup_reps=1, num_public_inputs=2, mix=(0.2, 0.2, 0.2)), 8, 8, 60),
])
def test_hip_proof_equals_oracle_proof_other_geometries(log_n, kw, fri_lde, cap, sec):
    """Column counts, lookup width / repetitions, public inputs, LDE factor and cap size away from the bench's."""
    c = S.sha_shaped_circuit(log_n, seed=7 * log_n, table_bits=2, **kw)
    S.check_satisfied(c)
    osetup = OP.Setup(c, fri_lde, cap, threads=8)
    po = OP.prove(c, osetup, fri_lde, cap, security_level=sec, threads=8)
    gsetup = E.ProverSetup(ctx(), c, fri_lde, cap, sec)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=sec)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), fri_lde, cap), pg, verbose=True)
    gsetup.close()


def test_golden_pinned_quotient_identity_accepts_hip_proof():
    """The quotient identity code that holds on the reference's golden proof (oracle/golden_quotient.py), fed with the
    VerificationKey JSON emitted for our circuit, accepts the HIP prover's openings."""
    import json
    import oracle as O
    from oracle import golden_quotient as GQ
    from era_boojum_amd import wire_format as W
    c = S.sha_shaped_circuit(10, seed=21, table_bits=2)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    buf, _ = gsetup.prove()
    proof = proof_format.parse(buf, security_level=30)
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, gsetup.cap(), 8, 16)))
    t = O.Transcript()
    t.absorb_cap(gsetup.cap())
    t.absorb(proof["public_inputs"])
    t.absorb_cap(np.array(proof["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(proof["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(proof["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), [g.name for g in c.gates], [], c.non_residues,
                                    dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    proof["values_at_z"], proof["values_at_z_omega"][0])
    assert lhs == rhs
    gsetup.close()


def _golden_identity_holds(c, cap, pg, specialized=()):
    """The quotient-identity code pinned by the reference's own proof (oracle/golden_quotient.py) on a proof of circuit c, fed with
    the VerificationKey JSON this repository emits for it."""
    import json
    import oracle as O
    from oracle import golden_quotient as GQ
    from era_boojum_amd import wire_format as W
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, cap, 8, 16)))
    t = O.Transcript()
    t.absorb_cap(cap)
    t.absorb(pg["public_inputs"])
    t.absorb_cap(np.array(pg["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(pg["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(pg["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), [g.name for g in c.gates], list(specialized), c.non_residues,
                                    dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    pg["values_at_z"], pg["values_at_z_omega"][0])
    return lhs == rhs, vk


@pytest.mark.parametrize("log_n,kw", [(10, {}), (12, dict(boolean_columns=2, specialized_constant_columns=3)),
                                      (9, dict(num_gp_vars=24, num_constant_cols=6, lookup_width=3, lookup_reps=11))])
def test_lookup_with_the_table_id_as_a_variable_column_equals_oracle_proof(log_n, kw):
    """LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (cs/mod.rs:237-241; compute_lookup_poly_pairs_specialized and
    compute_quotient_terms_for_lookup_specialized, lookup_argument_in_ext.rs:354-366, 949-1000; verifier.rs:1402-1464): width + 1
    variable columns per sub-argument, the last one the table id of THAT sub-argument on that row (the synthetic circuit gives
    every sub-argument its own table per row), no table-id constant column.  The HIP proof equals the oracle prover's byte for
    byte, the verifier restatement and the golden-pinned identity code accept it, the SetupBaseStorage dump (empty
    table_ids_column_idxes) selects the mode by itself, and a wrong table id breaks the verifier's lookup sumcheck."""
    import copy
    c = S.sha_shaped_circuit(log_n, seed=700 + log_n, table_bits=2, table_id_as_variable=True, **kw)
    w, reps = c.lookup_width, c.lookup_reps
    assert c.table_id_col == S.TABLE_ID_AS_VARIABLE and c.num_lookup_vars == (w + 1) * reps and S.check_satisfied(c)
    assert c.num_constant_cols == c.num_constants_for_gates + sum(g.reps * g.const_stride for g in c.specialized_gates)
    osetup = OP.Setup(c, 8, 16, threads=8)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=8)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=30)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), pg, verbose=True)
    spec = [(g.name, g.reps) for g in c.specialized_gates]
    ok, vk_json = _golden_identity_holds(c, gsetup.cap(), pg, spec)
    assert ok and "UseSpecializedColumnsWithTableIdAsVariable" in vk_json["fixed_parameters"]["lookup_parameters"]
    assert vk_json["fixed_parameters"]["table_ids_column_idxes"] == []
    # the verifier restatement built from that JSON alone (the reference's layout) accepts the proof too
    from oracle import golden_quotient as GQ
    vk2 = OV.vk_from_reference_geometry(GQ.geometry_from_vk_json(vk_json), gsetup.cap(), [g.name for g in c.gates], spec,
                                        c.non_residues, 8, 16)
    assert vk2.table_id_as_variable and vk2.num_vars == c.num_vars and vk2.num_constant_cols == c.num_constant_cols
    assert OV.verify(vk2, pg)
    if not c.specialized_gates:             # the dump reader takes the mode from the dump: no table-id column in it
        from era_boojum_amd import memcopy_format as M
        bare = copy.copy(c)
        bare.gates = [copy.copy(g) for g in c.gates]
        for g in bare.gates:
            g.path = []
        b = E.ProverSetup(ctx(), bare, 8, 16, 30, setup_base_dump=M.write_setup_base(c))
        assert np.array_equal(b.cap(), gsetup.cap())
        pb, _ = b.prove()
        assert np.array_equal(pb, buf)
        b.close()
    bad = c.variables.copy()                # a table id that names no table: the tuple of that sub-argument is in none
    col = c.num_gp_vars + (reps - 1) * (w + 1) + w
    bad[col, 7] = np.uint64(77)
    # every lookup term of the quotient vanishes by construction (A_i = 1 / denominator): the wrong tuple breaks the sum the
    # verifier checks at 0 (verifier.rs:1236-1256), as with the reference's prover
    bbuf, _ = gsetup.prove(variables=bad)
    assert not OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), proof_format.parse(bbuf, security_level=30))
    gsetup.close()
    wrong = copy.copy(c)                    # the table id declared as a constant column that does not exist
    wrong.table_id_col = c.num_constant_cols
    with pytest.raises(E.BoojumHipError):
        E.ProverSetup(ctx(), wrong, 8, 16, 30)


def test_poseidon_v1_transcript_proof_equals_oracle_proof():
    """The bench script's pairing: Poseidon2 tree hasher + Poseidon (v1) transcript (gadgets/sha256/mod.rs:289-293)."""
    c = S.sha_shaped_circuit(10, seed=41, table_bits=2)
    osetup = OP.Setup(c, 8, 16, threads=4)
    po = OP.prove(c, osetup, 8, 16, security_level=40, threads=4, transcript_kind=2)
    p2 = OP.prove(c, osetup, 8, 16, security_level=40, threads=4, transcript_kind=1)
    assert po["values_at_z"] != p2["values_at_z"]            # different challenges, as expected
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 40, transcript="poseidon")
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=40)
    _compare(pg, po)
    vk = OV.VerificationKey(c, gsetup.cap(), 8, 16)
    assert OV.verify(vk, pg, verbose=True, transcript_kind=2)
    assert not OV.verify(vk, pg, transcript_kind=1)
    gsetup.close()


@pytest.mark.parametrize("pow_bits,transcript,kind,runner", [(10, "poseidon2", 1, "blake2s"), (17, "poseidon", 2, "blake2s"),
                                                             (12, "blake2s", 3, "blake2s"), (4, "keccak256", 4, "keccak256"),
                                                             (18, "keccak256", 4, "keccak256"), (11, "poseidon2", 1, "keccak256")])
def test_proof_of_work(pow_bits, transcript, kind, runner):
    """PoW after the FRI commit phase with either runner of pow.rs — Blake2s256 (:50-133) or Keccak256 (:139-230), a type parameter
    independent of the transcript in the reference, so both the Keccak transcript / hasher pairing and a Poseidon2 transcript are
    run with the Keccak runner: fewer queries (compute_fri_schedule), the smallest valid nonce (what the reference's serial
    search returns), nonce absorbed as (low, high); proof identical to the oracle prover's (whose PoW runs on hashlib / the
    hashlib-pinned Keccak sponge) and accepted by the verifier restatement; a wrong nonce and the other runner are rejected."""
    rk = {"blake2s": 1, "keccak256": 2}[runner]
    c = S.sha_shaped_circuit(9, seed=60 + pow_bits, table_bits=2)
    osetup = OP.Setup(c, 8, 16, threads=4, hasher={3: 2, 4: 3}.get(kind, 1))
    po = OP.prove(c, osetup, 8, 16, security_level=40, pow_bits=pow_bits, threads=4, transcript_kind=kind, pow_runner=rk)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 40, pow_bits=pow_bits, transcript=transcript, pow_runner=runner)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=40)
    assert pg["proof_config"]["pow_bits"] == pow_bits
    assert pg["pow_challenge"] == po["pow_challenge"] and pg["pow_challenge"] != 0
    _compare(pg, po)
    p0 = OP.prove(c, osetup, 8, 16, security_level=40, pow_bits=0, threads=4, transcript_kind=kind)
    assert len(pg["queries_per_fri_repetition"]) < len(p0["queries_per_fri_repetition"])
    vk = OV.VerificationKey(c, gsetup.cap(), 8, 16)
    assert OV.verify(vk, pg, verbose=True, transcript_kind=kind, pow_runner=rk)
    if pow_bits >= 10:      # the other hash accepts this nonce with probability 2^-pow_bits
        assert not OV.verify(vk, pg, transcript_kind=kind, pow_runner=3 - rk)
    wrong = []
    for k in (1, 2, 3):      # a neighbouring nonce solves the puzzle too with probability 2^-pow_bits: not all three
        bad = dict(pg)
        bad["pow_challenge"] = pg["pow_challenge"] + k
        wrong.append(OV.verify(vk, bad, transcript_kind=kind, pow_runner=rk))
    assert not all(wrong)
    gsetup.close()


def test_repeated_proofs_are_identical_and_do_not_leak_device_memory():
    """Serving shape: one setup, many proofs.  Every proof of the same witness is the same bytes (no state leaks between
    proofs through the arena, the staging ring or the transcript), a failed proof in between does not poison the next
    one, and device memory stops growing after the first proof (workspace comes from the reused arena)."""
    import torch
    c = S.sha_shaped_circuit(10, seed=77, table_bits=2)
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    first, _ = gsetup.prove()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    bad = c.variables.copy()
    rows = np.nonzero(c.constants[0] == 1)[0]
    bad[3, rows[0]] = (int(bad[3, rows[0]]) + 1) % E.P
    for i in range(40):
        if i == 17:
            with pytest.raises(E.BoojumHipError, match="not satisfied"):
                gsetup.prove(variables=bad)
        buf, _ = gsetup.prove()
        assert np.array_equal(buf, first), i
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20)
    # bj_ctx_release_workspace hands the arena back (more free memory than while it was held) and the next proof, which
    # rebuilds it, is the same bytes again
    ctx().release_workspace()
    assert torch.cuda.mem_get_info()[0] > free0
    buf, _ = gsetup.prove()
    assert np.array_equal(buf, first)
    gsetup.close()


def test_setup_rejects_descriptors_that_would_index_past_the_columns():
    """bj_setup_create validates every column index a gate descriptor implies (repetitions x stride + operand), the public
    input locations and the domain sizes of the per-coset tables instead of letting kernels read past HBM buffers."""
    import copy
    c = S.sha_shaped_circuit(8, seed=3, table_bits=2)
    bad = copy.copy(c)
    bad.gates = [copy.copy(g) for g in c.gates]
    bad.gates[1].reps = 16                                   # 16 FMA repetitions of stride 4 need 64 > 60 columns
    with pytest.raises(E.BoojumHipError, match="reads variable column"):
        E.ProverSetup(ctx(), bad, 8, 16, 20)
    bad = copy.copy(c)
    bad.gates = [copy.copy(g) for g in c.gates]
    bad.gates[0].const_stride = 7                            # ConstantsAllocator: repetition 3 would read constant column 3 + 21
    with pytest.raises(E.BoojumHipError, match="constant column"):
        E.ProverSetup(ctx(), bad, 8, 16, 20)
    bad = copy.copy(c)
    bad.public_inputs = [(c.num_vars, 0, 1)]                 # column past the trace
    with pytest.raises(E.BoojumHipError, match="outside"):
        E.ProverSetup(ctx(), bad, 8, 16, 20)
    bad = copy.copy(c)
    bad.public_inputs = [(0, c.n, 1)]                        # row past the trace
    with pytest.raises(E.BoojumHipError, match="outside"):
        E.ProverSetup(ctx(), bad, 8, 16, 20)
    with pytest.raises(E.BoojumHipError, match="limited to 64"):
        E.ProverSetup(ctx(), c, 128, 16, 20)


def test_prover_checks_public_values_against_the_witness():
    c = S.sha_shaped_circuit(8, seed=4, table_bits=2)
    assert c.public_inputs
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 20)
    vals = [v for (_, _, v) in c.public_inputs]
    vals[0] = (vals[0] + 1) % E.P
    with pytest.raises(E.BoojumHipError, match="public input 0"):
        gsetup.prove(public_values=vals)
    buf, _ = gsetup.prove()                                  # and the setup is still usable
    assert OV.verify(OV.VerificationKey(c, gsetup.cap(), 8, 16), proof_format.parse(buf, security_level=20))
    gsetup.close()


@pytest.mark.streaming(260)
@pytest.mark.timeout(900)
def test_proof_at_2p23_rows_equals_the_streaming_oracle():
    """BASELINE config 5's size (2^23 rows, LDE 8 => 2^26-point oracles), prover.rs:153-168.  The coset-streaming restatement
    (oracle/prover_streaming.py) recomputes from the witness alone EVERY transcript input: the caps of the witness, second-stage and
    quotient oracles over all eight cosets (2^26 leaves each; since round 6 the oracle hashes eight leaves per AVX-512 call,
    oracle/poseidon2_avx512.c, and no cap is taken as claimed any more), all eight cosets of the setup oracle against the
    verification key's cap, all 241 values at z, z*omega and 0, then DEEP on every one of the 2^26 points (a second pass over the
    cosets), do_fri, queries — FRI base cap, every intermediate cap, final monomials, every FRI query opening and the path + leaf
    hash of every base-oracle opening of every query equal the HIP proof's.  The verifier restatement accepts the proof as a whole
    (public inputs included — their DEEP opening sets run over all 2^26 points; one of those launches lost a loop's exit test to
    an undeclared SCC write until round 2, tests/test_gpu_openings.py)."""
    from oracle import prover_streaming as PS
    c = S.sha_shaped_circuit(23, seed=42, table_bits=4)
    setup = E.ProverSetup(ctx(), c, 8, 16, 100)
    d_vars, d_mult = ctx().upload(c.variables), ctx().upload(c.multiplicities)
    buf, stage_ms = setup.prove_dev(d_vars, d_mult)
    assert sum(v for k, v in stage_ms.items() if k != "witness_tree_leaf_kernel") < 5000.0      # ms; the regression took 500 s
    cap = setup.cap()
    setup.close()
    ctx().free(d_vars)
    ctx().free(d_mult)
    ctx().release_workspace()      # ~150 GB of arena: give it back before the multi-process tests share this GPU
    pg = proof_format.parse(buf, security_level=100)
    assert OV.verify(OV.VerificationKey(c, cap, 8, 16), pg, verbose=True)
    po = PS.commitments_and_openings(c, cap, 8, 16, threads=oracle_threads(), transcript_kind=1, check_setup_cosets=tuple(range(8)),
                                     rest_of_the_proof=True, security_level=100)
    for k in ("public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap", "values_at_z", "values_at_z_omega",
              "values_at_0"):
        assert pg[k] == po[k], k
    for cs, frag in po["setup_cap_fragments"].items():
        assert np.array_equal(frag, cap[2 * cs:2 * cs + 2]), "setup cap nodes of coset %d" % cs
    assert sorted(po["setup_cap_fragments"]) == list(range(8))
    # past the openings: the DEEP accumulator of all 2^26 points coset by coset, every FRI cap, the final monomials and all FRI query
    # openings byte for byte; the witness / second-stage / quotient / setup opening (leaf hash) + path of EVERY query
    compared = PS.compare_rest_of_the_proof(pg, po)
    nq = len(pg["queries_per_fri_repetition"])
    assert compared == 4 * nq and len(po["query_indexes"]) == nq


def test_two_contexts_on_two_host_threads_prove_concurrently():
    """The shape a Rust host with 8 GPUs uses: one bj_ctx per host thread, all inside ONE process (here both on this box's one
    device, each context on its own HIP stream).  Each thread proves its own circuit eight times while the other one is
    proving: every proof is the bytes the same setup gave when it ran alone — no cross-talk through the twiddle caches, the
    arenas, the staging rings, the gate-kernel caches or the environment switches (read once per process, kernels.h)."""
    import threading
    import torch
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = [E.Context(0, stream=s.cuda_stream) for s in streams]
    circuits = [S.sha_shaped_circuit(12, seed=101, table_bits=2), S.sha_shaped_circuit(13, seed=102, table_bits=2)]
    setups = [E.ProverSetup(ctxs[i], circuits[i], 8, 16, 40) for i in range(2)]
    alone = [setups[i].prove()[0].copy() for i in range(2)]
    for i in range(2):
        assert OV.verify(OV.VerificationKey(circuits[i], setups[i].cap(), 8, 16), proof_format.parse(alone[i], security_level=40))
    errors, barrier = [], threading.Barrier(2)

    def work(i):
        try:
            barrier.wait()
            for k in range(8):
                buf, _ = setups[i].prove()
                if not np.array_equal(buf, alone[i]):
                    errors.append("thread %d, proof %d differs from the proof made alone" % (i, k))
        except Exception as e:          # noqa: BLE001 - reported by the main thread
            errors.append("thread %d: %r" % (i, e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for s in setups:
        s.close()
    for c in ctxs:
        c.release_workspace()
        c.close()


def test_monomial_layouts_at_2p22_rows_give_the_same_proof():
    """At 2^22 rows bj_prove keeps monomials in the tiled layout (inverse transforms without a bit-reversal pass, front-pass tiles read
    contiguously; csrc/ntt_r16.hip).  BJ_MONO_TILED=0 keeps them in natural order (the path every other trace length takes) and
    BJ_NTT_TWO_PASS=0 goes back to the three-pass plan: setup caps and proofs are the same bytes under all three, resident and
    host-witness entry points alike."""
    import os
    ctx()                                  # PyTorch's HIP runtime first (tests/gpu_util.py), then the library
    lib = E.load_library()
    c = S.sha_shaped_circuit(22, seed=42, table_bits=4)
    d_vars, d_mult = ctx().upload(c.variables), ctx().upload(c.multiplicities)
    got = {}
    try:
        for name, env in (("tiled", {}), ("natural", {"BJ_MONO_TILED": "0"}), ("three_pass", {"BJ_NTT_TWO_PASS": "0"})):
            os.environ.update(env)
            lib.bj_env_reload()
            assert ctx().monomials_tiled(22) == (name == "tiled")
            setup = E.ProverSetup(ctx(), c, 8, 16, 100)
            buf, _ = setup.prove_dev(d_vars, d_mult)
            got[name] = (buf.copy(), setup.cap().copy())
            if name != "three_pass":
                assert np.array_equal(setup.prove()[0], buf), name + ": host-witness entry point"
            setup.close()
            for k in env:
                os.environ.pop(k)
    finally:
        for k in ("BJ_MONO_TILED", "BJ_NTT_TWO_PASS"):
            os.environ.pop(k, None)
        lib.bj_env_reload()
    for name in ("natural", "three_pass"):
        assert np.array_equal(got[name][0], got["tiled"][0]) and np.array_equal(got[name][1], got["tiled"][1]), name
    ctx().free(d_vars)
    ctx().free(d_mult)
    ctx().release_workspace()


def test_pipelined_drop_in_call_gives_the_serial_proofs():
    """bj_prove_async / bj_proof_wait (csrc/prove_async.hip): the host loop over witnesses around prove_cpu_basic
    (prover.rs:153-168, convenience.rs:119-196) with two proofs in flight from ONE host thread.  Two setups of different sizes share
    the context's two lanes; every proof that comes back is the bytes the serial bj_prove gives, in submission order whatever
    order the lanes finish in; a third submission waits for its lane; an unsatisfied witness fails in `wait` with the
    prover's message and leaves the lanes usable; the lanes' workspaces go with bj_ctx_release_workspace."""
    c1, c2 = S.sha_shaped_circuit(13, seed=31, table_bits=2), S.sha_shaped_circuit(11, seed=32, table_bits=2)
    s1, s2 = E.ProverSetup(ctx(), c1, 8, 16, 40), E.ProverSetup(ctx(), c2, 8, 16, 40)
    ref1, ref2 = s1.prove()[0].copy(), s2.prove()[0].copy()
    assert OV.verify(OV.VerificationKey(c1, s1.cap(), 8, 16), proof_format.parse(ref1, security_level=40))
    # the documented loop: t[k] = async(w[k]); wait(t[k-1])
    order = [s1, s2, s1, s1, s2, s2, s1]
    refs = {id(s1): ref1, id(s2): ref2}
    prev = order[0].prove_async()
    for k in range(1, len(order)):
        cur = order[k].prove_async()
        buf, stages = order[k - 1].wait(prev)
        assert np.array_equal(buf, refs[id(order[k - 1])]), "proof %d" % (k - 1)
        assert stages["witness_lde_and_tree"] > 0
        prev = cur
    buf, _ = order[-1].wait(prev)
    assert np.array_equal(buf, refs[id(order[-1])])
    # three submissions before the first wait (the third blocks inside the library until lane 0 is free), waited out of order
    ts = [s1.prove_async(), s2.prove_async(), s1.prove_async()]
    assert np.array_equal(s1.wait(ts[2])[0], ref1)
    assert np.array_equal(s1.wait(ts[0])[0], ref1)
    assert s2.done(ts[1]) in (True, False)
    assert np.array_equal(s2.wait(ts[1])[0], ref2)
    # an unsatisfied witness: the error arrives in wait, the other proof in flight is untouched
    rows = np.nonzero(c1.constants[0] == 1)[0]
    bad = c1.variables.copy()
    bad[3, rows[0]] = (int(bad[3, rows[0]]) + 1) % E.P
    tb, tg = s1.prove_async(variables=bad), s2.prove_async()
    with pytest.raises(E.BoojumHipError, match="not satisfied"):
        s1.wait(tb)
    assert np.array_equal(s2.wait(tg)[0], ref2)
    assert np.array_equal(s1.wait(s1.prove_async())[0], ref1)
    # a circuit whose gates are op lists (generated / run-time compiled kernels, non-copiable witness columns): both lanes launch the
    # same uploaded programs concurrently
    c3 = S.recursion_like_circuit(10, seed=7)
    s3 = E.ProverSetup(ctx(), c3, 8, 16, 30)
    ref3 = s3.prove()[0].copy()
    ts = [s3.prove_async(), s3.prove_async(), s3.prove_async(), s3.prove_async()]
    for t in ts:
        assert np.array_equal(s3.wait(t)[0], ref3)
    # the other lane policies (by default chosen by witness size, csrc/prove_async.hip): whole witness first (1), groups transferred and
    # transformed as they land but hashed once (2), bj_prove as is without the stagger — same bytes
    import os
    lib = E.load_library()
    try:
        for env in ({"BJ_ASYNC_MODE": "1"}, {"BJ_ASYNC_MODE": "2"}, {"BJ_ASYNC_MODE": "0", "BJ_ASYNC_STAGGER": "0"}):
            os.environ.update(env)
            lib.bj_env_reload()
            ts = [s1.prove_async(), s2.prove_async(), s1.prove_async(), s1.prove_async(), s2.prove_async()]
            for t, (st, ref) in zip(ts, ((s1, ref1), (s2, ref2), (s1, ref1), (s1, ref1), (s2, ref2))):
                assert np.array_equal(st.wait(t)[0], ref), env
            for k in env:
                os.environ.pop(k)
    finally:
        for k in ("BJ_ASYNC_MODE", "BJ_ASYNC_STAGGER"):
            os.environ.pop(k, None)
        lib.bj_env_reload()
    s1.close(); s2.close(); s3.close()
    ctx().release_workspace()          # frees the lanes' arenas and witness staging too


def test_host_witness_group_plans_and_non_residue_paths_give_the_same_proof():
    """bj_prove takes the witness over PCIe in groups and hashes it group by group (prover.hip: quarters of the first eight
    columns, then 8, 16, 16, 24, ... columns per absorption run); bj_prove_dev hashes the resident witness in one kernel.  Every
    plan — the default one, the round-4 uniform one, other group widths, no group-wise absorption — and both forms of the
    copy-permutation numerator (32-bit non-residue multipliers, the default for make_non_residues' output, and 64-bit products)
    must give the bytes of the resident path, for the bench geometry (93 leaf columns) and for a 156-column witness oracle."""
    import os
    lib = E.load_library()
    for c in (S.sha_shaped_circuit(11, seed=77, table_bits=2), S.recursion_like_circuit(10, seed=7)):
        setup = E.ProverSetup(ctx(), c, 8, 16, 30)
        d_vars, d_mult = ctx().upload(c.variables), ctx().upload(c.multiplicities)
        ref, _ = setup.prove_dev(d_vars, d_mult)
        for env in ({}, {"BJ_PROVE_UNIFORM_GROUPS": "1"}, {"BJ_PROVE_H2D_GROUP": "16"}, {"BJ_PROVE_H2D_GROUP": "3"}, {"BJ_PROVE_NO_ABSORB": "1"},
                    {"BJ_COPY_PERM_WIDE_K": "1"}):
            os.environ.update(env)
            try:
                lib.bj_env_reload()
                got, _ = setup.prove()
                if "BJ_COPY_PERM_WIDE_K" in env:
                    dev, _ = setup.prove_dev(d_vars, d_mult)
                    assert np.array_equal(dev, ref), env
            finally:
                for k in env:
                    del os.environ[k]
                lib.bj_env_reload()
            assert np.array_equal(got, ref), env
        setup.close()
        ctx().free(d_vars)
        ctx().free(d_mult)
