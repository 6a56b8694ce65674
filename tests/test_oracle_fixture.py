"""Pins the C oracle (oracle/*.c) against the reference's own golden proof (tests/golden/boojum_fixture.json, cut from
/root/reference/proof.json + vk.json by tests/golden/make_fixture.py).

What is pinned bit-exactly here:
  * Poseidon2 permutation + overwrite sponge + leaf/node hashing + Merkle path/cap convention   (oracle/poseidon2.c)
  * Poseidon2 Fiat–Shamir transcript and BoolsBuffer query-index extraction                     (oracle/transcript.c)
  * compute_fri_schedule                                                                        (oracle/fri.c)
  * the FRI fold formula incl. root indexing / coset_inv squaring / challenge squaring          (oracle/fri.c)
  * iNTT convention (final monomials == iNTT of the last folded layer, checked via Horner)      (oracle/ntt.c)
  * the DEEP quotient formula and opening order                                                 (python ints here)
Order of the replay follows verifier.rs:924-1076, 1819-1955.
"""
import numpy as np
import pytest

import oracle as O

P = O.P


def emul(a, b): return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def eadd(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def esub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def escale(a, s): return (a[0] * s % P, a[1] * s % P)
def einv(a):
    ni = O.inv((a[0] * a[0] - 7 * a[1] * a[1]) % P)
    return (a[0] * ni % P, (-a[1] * ni) % P)


@pytest.fixture(scope="module")
def replay(fixture_json):
    fx = fixture_json
    t = O.Transcript()
    t.absorb_cap(fx["setup_merkle_tree_cap"])
    t.absorb(fx["public_inputs"])
    t.absorb_cap(fx["witness_oracle_cap"])
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(fx["stage_2_oracle_cap"])
    alpha = t.challenge_ext()
    t.absorb_cap(fx["quotient_oracle_cap"])
    z = t.challenge_ext()
    for k in ("values_at_z", "values_at_z_omega", "values_at_0"):
        t.absorb(np.array(fx[k], dtype=np.uint64))
    c = t.challenge_ext()
    t.absorb_cap(fx["fri_base_oracle_cap"])
    fri_ch = [t.challenge_ext()]
    for cap in fx["fri_intermediate_oracles_caps"]:
        t.absorb_cap(cap)
        fri_ch.append(t.challenge_ext())
    t.absorb(fx["final_fri_monomials"][0])
    t.absorb(fx["final_fri_monomials"][1])
    n_log = fx["geometry"]["domain_size"].bit_length() - 1
    log_lde = fx["proof_config"]["fri_lde_factor"].bit_length() - 1
    qi = O.QueryIndexer(n_log, log_lde)
    idxs = [qi.next(t) for _ in fx["queries"]]
    return dict(z=z, c=c, fri_ch=fri_ch, idxs=idxs, n_log=n_log, log_lde=log_lde, alpha=alpha, beta=beta, gamma=gamma)


def test_fri_schedule_matches_fixture_shape(fixture_json):
    fx = fixture_json
    pc = fx["proof_config"]
    n_log = fx["geometry"]["domain_size"].bit_length() - 1
    new_pow, nq, sched, final_deg = O.fri_schedule(pc["security_level"], pc["merkle_tree_cap_size"], pc["pow_bits"],
                                                   pc["fri_lde_factor"].bit_length() - 1, n_log)
    assert sched == [3, 3, 3, 3, 3, 1]
    assert final_deg == len(fx["final_fri_monomials"][0]) == 16
    assert nq == fx["num_queries_total"] == 100
    assert new_pow == 0
    assert len(fx["fri_intermediate_oracles_caps"]) == len(sched) - 1
    # SHA bench shapes quoted in SURVEY.md appendix A
    assert O.fri_schedule(100, 16, 0, 3, 16)[1:] == (34, [3, 3, 3, 3, 3], 2)
    assert O.fri_schedule(100, 16, 0, 3, 20)[2] == [3, 3, 3, 3, 3, 3, 1]


def test_transcript_replay_gives_query_indices(replay):
    # query 0 index found independently by brute-forcing path directions in SURVEY.md D8
    assert replay["idxs"][0] == 1192677
    assert replay["idxs"][:4] == [1192677, 1852513, 74792, 367254]


def test_base_oracle_merkle_paths(fixture_json, replay):
    fx = fixture_json
    caps = {"witness_query": fx["witness_oracle_cap"], "stage_2_query": fx["stage_2_oracle_cap"],
            "quotient_query": fx["quotient_oracle_cap"], "setup_query": fx["setup_merkle_tree_cap"]}
    widths = {"witness_query": 156, "stage_2_query": 58, "quotient_query": 16, "setup_query": 167}
    for q, idx in zip(fx["queries"], replay["idxs"]):
        for name, cap in caps.items():
            leaf = q[name]["leaf_elements"]
            assert len(leaf) == widths[name]
            path = np.array(q[name]["proof"], dtype=np.uint64)
            assert path.shape == (16, 4)
            assert O.merkle_verify(path, np.array(cap, dtype=np.uint64), O.hash_leaf(leaf), idx), (name, idx)
            # negative control: wrong index must fail
            assert not O.merkle_verify(path, np.array(cap, dtype=np.uint64), O.hash_leaf(leaf), idx ^ 1)


def _deep_value(fx, rp, q, idx):
    """DEEP quotient at x = g*w_N^bitrev(idx): prover.rs:1828-2043, 2523-2706; source order verifier.rs:2233-2290."""
    n_log, LOGN = rp["n_log"], rp["n_log"] + rp["log_lde"]
    g = fx["geometry"]
    NV, NW, NM = g["num_variable_columns"] + 25, 0, 1  # 130 general-purpose + 8*3+1 specialized columns = 155
    assert NV == 155
    NC = g["num_constant_columns"] + g["extra_constant_polys_for_selectors"] + 1
    NS, NT, NI, NA = NV, 4, 19, 8
    W, S2, Q, SU = (q[k]["leaf_elements"] for k in ("witness_query", "stage_2_query", "quotient_query", "setup_query"))
    base = lambda l: [(e, 0) for e in l]
    ext = lambda l: [(l[i], l[i + 1]) for i in range(0, len(l), 2)]
    vz = [tuple(e) for e in fx["values_at_z"]]
    vzo = [tuple(e) for e in fx["values_at_z_omega"]]
    v0 = [tuple(e) for e in fx["values_at_0"]]
    pubs = [(cr[0], cr[1], v) for cr, v in zip(g["public_inputs_locations"], fx["public_inputs"])]
    total = len(vz) + len(vzo) + len(v0) + len(pubs)
    chs = [(1, 0), rp["c"]]
    while len(chs) < total:
        chs.append(emul(chs[-1], rp["c"]))
    x = pow(O.omega(LOGN), O.bitrev(idx, LOGN), P) * 7 % P
    src = (base(W[0:NV]) + base(W[NV:NV + NW]) + base(SU[NS:NS + NC]) + base(SU[0:NS]) + ext(S2[0:2])
           + ext(S2[2:2 + 2 * NI]) + base(W[NV + NW:NV + NW + NM]) + ext(S2[2 + 2 * NI:2 + 2 * NI + 2 * NA])
           + ext(S2[2 + 2 * NI + 2 * NA:]) + base(SU[NS + NC:NS + NC + NT]) + ext(Q))
    assert len(src) == len(vz)

    def quot(srcs, vals, at, ws):
        acc = (0, 0)
        for s_, v_, w_ in zip(srcs, vals, ws):
            acc = eadd(acc, emul(w_, esub(s_, v_)))
        return emul(acc, einv(esub((x, 0), at)))
    o = 0
    h = quot(src, vz, rp["z"], chs[o:o + len(vz)]); o += len(vz)
    h = eadd(h, quot(ext(S2[0:2]), vzo, escale(rp["z"], O.omega(n_log)), chs[o:o + 1])); o += 1
    h = eadd(h, quot(ext(S2[2 + 2 * NI:]), v0, (0, 0), chs[o:o + len(v0)])); o += len(v0)
    at = (pow(O.omega(n_log), pubs[0][1], P), 0)
    h = eadd(h, quot([(W[ci], 0) for ci, _, _ in pubs], [(v, 0) for _, _, v in pubs], at, chs[o:o + len(pubs)]))
    return h


def test_c_oracle_deep_quotient_matches_fixture(fixture_json, replay):
    """The C oracle's DEEP point function (oracle/pointwise.c, restating prover.rs:2523-2706) summed over the four
    opening sets must reproduce the value committed in the FRI base oracle of the golden proof."""
    fx, rp = fixture_json, replay
    n_log, LOGN = rp["n_log"], rp["n_log"] + rp["log_lde"]
    g = fx["geometry"]
    NV, NW, NM, NS, NT, NI, NA = 155, 0, 1, 155, 4, 19, 8
    NC = g["num_constant_columns"] + g["extra_constant_polys_for_selectors"] + 1
    vz = [tuple(e) for e in fx["values_at_z"]]
    vzo = [tuple(e) for e in fx["values_at_z_omega"]]
    v0 = [tuple(e) for e in fx["values_at_0"]]
    pubs = [(cr[0], cr[1], v) for cr, v in zip(g["public_inputs_locations"], fx["public_inputs"])]
    total = len(vz) + len(vzo) + len(v0) + len(pubs)
    chs = [(1, 0), rp["c"]]
    while len(chs) < total:
        chs.append(emul(chs[-1], rp["c"]))
    for q, idx in zip(fx["queries"], rp["idxs"]):
        W, S2, Q, SU = (q[k]["leaf_elements"] for k in ("witness_query", "stage_2_query", "quotient_query", "setup_query"))
        base = lambda l: [(e, None) for e in l]
        ext = lambda l: [(l[i], l[i + 1]) for i in range(0, len(l), 2)]
        src = (base(W[0:NV]) + base(W[NV:NV + NW]) + base(SU[NS:NS + NC]) + base(SU[0:NS]) + ext(S2[0:2])
               + ext(S2[2:2 + 2 * NI]) + base(W[NV + NW:NV + NW + NM]) + ext(S2[2 + 2 * NI:2 + 2 * NI + 2 * NA])
               + ext(S2[2 + 2 * NI + 2 * NA:]) + base(SU[NS + NC:NS + NC + NT]) + ext(Q))
        x = pow(O.omega(LOGN), O.bitrev(idx, LOGN), P) * 7 % P
        o = 0
        h = O.deep_quotient_point(src, vz, chs[o:o + len(vz)], rp["z"], x); o += len(vz)
        h = eadd(h, O.deep_quotient_point(ext(S2[0:2]), vzo, chs[o:o + 1], escale(rp["z"], O.omega(n_log)), x)); o += 1
        h = eadd(h, O.deep_quotient_point(ext(S2[2 + 2 * NI:]), v0, chs[o:o + len(v0)], (0, 0), x)); o += len(v0)
        at = (pow(O.omega(n_log), pubs[0][1], P), 0)
        h = eadd(h, O.deep_quotient_point([(W[ci], None) for ci, _, _ in pubs], [(v, 0) for _, _, v in pubs],
                                          chs[o:o + len(pubs)], at, x))
        assert h == _deep_value(fx, rp, q, idx)
        le = q["fri_queries"][0]["leaf_elements"]
        assert h == (le[idx % 8], le[8 + idx % 8])


def test_fri_chain_fold_and_final_monomials(fixture_json, replay):
    fx, rp = fixture_json, replay
    LOGN = rp["n_log"] + rp["log_lde"]
    sched = [3, 3, 3, 3, 3, 1]
    roots = O.twiddles(LOGN, inverse=True)          # inverse twiddles of the FULL LDE domain (fri/mod.rs:192-193)
    caps = [fx["fri_base_oracle_cap"]] + fx["fri_intermediate_oracles_caps"]
    for q, idx in zip(fx["queries"], rp["idxs"]):
        cur = _deep_value(fx, rp, q, idx)
        fidx, kappa = idx, O.inv(7)
        for layer, (k, fq) in enumerate(zip(sched, q["fri_queries"])):
            m = 1 << k
            sub, tree_idx = fidx % m, fidx >> k
            le = fq["leaf_elements"]
            assert len(le) == 2 * m
            c0 = np.array(le[:m], dtype=np.uint64)
            c1 = np.array(le[m:], dtype=np.uint64)
            # the value carried from the previous layer (layer 0: the DEEP value) sits in the queried slot
            assert (int(c0[sub]), int(c1[sub])) == cur, ("carried value", layer)
            path = np.array(fq["proof"], dtype=np.uint64).reshape(-1, 4)
            assert O.merkle_verify(path, np.array(caps[layer], dtype=np.uint64), O.hash_leaf(le), tree_idx), layer
            chal = rp["fri_ch"][layer]
            start = tree_idx * m
            for _ in range(k):
                half = c0.size // 2
                r = roots[(start >> 1):(start >> 1) + half]
                c0, c1 = O.fri_fold(c0, c1, r, kappa, chal)      # C oracle fold on the leaf's values
                start >>= 1
                chal = emul(chal, chal)
                kappa = kappa * kappa % P
            cur, fidx = (int(c0[0]), int(c1[0])), tree_idx
        x = pow(O.omega(LOGN), O.bitrev(idx, LOGN), P) * 7 % P
        for _ in range(sum(sched)):
            x = x * x % P
        acc = (0, 0)
        for cc in reversed(list(zip(*fx["final_fri_monomials"]))):
            acc = eadd(escale(acc, x), cc)
        assert acc == cur, "final fold == Horner(final monomials)"


def test_last_fri_layer_leaves_in_cap(fixture_json):
    # fri_queries[5]: 4-element leaves, zero-length path -> the leaf hash IS a cap element
    fx = fixture_json
    cap = [tuple(c) for c in fx["fri_intermediate_oracles_caps"][4]]
    for q in fx["queries"]:
        fq = q["fri_queries"][5]
        assert fq["proof"] == []
        assert tuple(int(x) for x in O.hash_leaf(fq["leaf_elements"])) in cap


def test_lookup_sum_relation_at_zero(fixture_json):
    """Log-derivative lookup argument (verifier.rs:1139-1260): sum_i A_i(0) == B(0) over the multiplicity polys — pins
    the number and order of the lookup polynomials in `values_at_0` (8 sub-arguments A_i, then B) on the golden proof."""
    fx = fixture_json
    v0 = fx["values_at_0"]
    reps = fx["geometry"]["lookup"]["UseSpecializedColumnsWithTableIdAsConstant"]["num_repetitions"]
    assert len(v0) == reps + 1
    for comp in (0, 1):
        assert sum(e[comp] for e in v0[:reps]) % P == v0[reps][comp]


GOLDEN_GENERAL_GATES = ["ConstantsAllocatorGate", "U8x4FMAGate", "Poseidon2FlattenedGate", "DotProductGate<4>", "ZeroCheckGate",
                        "FmaGateInBaseFieldWithoutConstant", "UIntXAddGate", "SelectionGate", "ParallelSelectionGate<4>", "NopGate",
                        "ReductionGate<4>"]   # evaluator order of the inner circuit, recursive_verifier.rs:2302-2362


def test_quotient_identity_of_the_golden_proof(fixture_json, replay):
    """verifier.rs:1090-1810 on the reference's own proof: sum_i alpha^i term_i(z) == t(z) * (z^n - 1) with the lookup terms,
    the BooleanConstraintGate over its specialized column, the eleven evaluators over general-purpose columns behind their
    selector paths (incl. the 118-relation Poseidon2 flattened gate), (z(x)-1)*L_1 and the 20 copy-permutation chunks.
    Pins the alpha order, the selector-path convention and the copy-permutation / lookup / gate formulas that
    oracle/prover.py, oracle/verifier.py and the HIP prover share."""
    from oracle import golden_quotient as GQ
    from era_boojum_amd.synthetic import non_residues
    fx = fixture_json
    t = O.Transcript()
    t.absorb_cap(fx["setup_merkle_tree_cap"])
    t.absorb(fx["public_inputs"])
    t.absorb_cap(fx["witness_oracle_cap"])
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    ch = dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=replay["alpha"], z=replay["z"])
    assert (beta, gamma) == (replay["beta"], replay["gamma"])
    nr = non_residues(155, fx["geometry"]["domain_size"])                     # make_non_residues, utils.rs:636-688
    lhs, rhs = GQ.quotient_identity(fx["geometry"], GOLDEN_GENERAL_GATES, [("BooleanConstraintGate", 1)], nr, ch,
                                    fx["values_at_z"], fx["values_at_z_omega"][0])
    assert lhs == rhs
    # and it is a real check: any single opening changed breaks it
    bad = [list(v) for v in fx["values_at_z"]]
    bad[17][0] = (bad[17][0] + 1) % P
    lhs2, rhs2 = GQ.quotient_identity(fx["geometry"], GOLDEN_GENERAL_GATES, [("BooleanConstraintGate", 1)], nr, ch,
                                      bad, fx["values_at_z_omega"][0])
    assert lhs2 != rhs2


def test_golden_pinned_identity_accepts_our_proofs():
    """The same (golden-pinned) identity code, fed with the VerificationKey JSON this repository emits for its own circuit
    class, accepts a proof of the oracle prover — ties the conventions pinned above to the provers under test."""
    import json
    from oracle import golden_quotient as GQ
    from oracle import prover as OP
    from era_boojum_amd import synthetic as S, wire_format as W
    c = S.sha_shaped_circuit(9, seed=11, table_bits=2)
    setup = OP.Setup(c, 4, 8, threads=2)
    proof = OP.prove(c, setup, 4, 8, security_level=20, threads=2)
    vk = json.loads(W.dumps(W.vk_to_reference_json(c, np.asarray(setup.cap), 4, 8)))
    t = O.Transcript()
    t.absorb_cap(np.asarray(setup.cap))
    t.absorb(proof["public_inputs"])
    t.absorb_cap(np.array(proof["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma, lbeta, lgamma = (t.challenge_ext() for _ in range(4))
    t.absorb_cap(np.array(proof["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(proof["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    gates = [g.name for g in c.gates]
    lhs, rhs = GQ.quotient_identity(GQ.geometry_from_vk_json(vk), gates, [], c.non_residues,
                                    dict(beta=beta, gamma=gamma, lookup_beta=lbeta, lookup_gamma=lgamma, alpha=alpha, z=z),
                                    proof["values_at_z"], proof["values_at_z_omega"][0])
    assert lhs == rhs


def _golden_proof_dict(fx):
    return {"proof_config": fx["proof_config"], "public_inputs": fx["public_inputs"],
            "witness_oracle_cap": fx["witness_oracle_cap"], "stage_2_oracle_cap": fx["stage_2_oracle_cap"],
            "quotient_oracle_cap": fx["quotient_oracle_cap"], "values_at_z": fx["values_at_z"],
            "values_at_z_omega": fx["values_at_z_omega"], "values_at_0": fx["values_at_0"],
            "fri_base_oracle_cap": fx["fri_base_oracle_cap"], "fri_intermediate_oracles_caps": fx["fri_intermediate_oracles_caps"],
            "final_fri_monomials": fx["final_fri_monomials"], "pow_challenge": fx["pow_challenge"],
            "queries_per_fri_repetition": fx["queries"]}


def test_whole_proof_verifier_accepts_the_golden_proof(fixture_json):
    """oracle/verifier.py::verify — the very function that accepts the proofs of the oracle prover and of the HIP prover —
    run on the reference's own proof.json / vk.json (the six query openings the fixture keeps): transcript, opening
    counts and order, lookup sumcheck, quotient identity over twelve evaluator types, DEEP over all four point sets incl.
    the public inputs, Merkle paths of the four base oracles, the FRI chain down to the final monomials.  One function,
    pinned end to end by reference-produced data; every tampered part below is rejected."""
    import copy
    from oracle import verifier as OV
    from era_boojum_amd.synthetic import non_residues
    fx = fixture_json
    cfg = fx["proof_config"]
    vk = OV.vk_from_reference_geometry(fx["geometry"], fx["setup_merkle_tree_cap"], GOLDEN_GENERAL_GATES,
                                       [("BooleanConstraintGate", 1)], non_residues(155, fx["geometry"]["domain_size"]),
                                       cfg["fri_lde_factor"], cfg["merkle_tree_cap_size"])
    assert (vk.num_vars, vk.num_constant_cols, vk.lookup_width, vk.lookup_reps, vk.quotient_degree) == (155, 8, 3, 8, 8)
    proof = _golden_proof_dict(fx)
    assert OV.verify(vk, proof, verbose=True, partial_queries=True)
    assert not OV.verify(vk, proof), "all 100 queries are required without partial_queries"

    def tampered(edit):
        p = copy.deepcopy(proof)
        edit(p)
        return OV.verify(vk, p, partial_queries=True)
    def bump(lst, i, j=None):
        if j is None:
            lst[i] = (lst[i] + 1) % P
        else:
            lst[i][j] = (lst[i][j] + 1) % P
    assert not tampered(lambda p: bump(p["values_at_z"], 200, 0))                                  # quotient identity
    assert not tampered(lambda p: bump(p["values_at_0"], 3, 1))                                    # lookup sumcheck
    assert not tampered(lambda p: bump(p["public_inputs"], 1))                                     # transcript + DEEP
    assert not tampered(lambda p: bump(p["queries_per_fri_repetition"][2]["witness_query"]["leaf_elements"], 40))   # Merkle leaf
    assert not tampered(lambda p: bump(p["queries_per_fri_repetition"][4]["fri_queries"][3]["proof"][0], 2))        # FRI path
    assert not tampered(lambda p: bump(p["final_fri_monomials"][1], 5))                            # final monomials
    assert not tampered(lambda p: bump(p["quotient_oracle_cap"][7], 0))                            # a cap
    vk2 = copy.copy(vk)
    vk2.specialized_gates = []
    assert not OV.verify(vk2, proof, partial_queries=True)                                         # the boolean gate matters
