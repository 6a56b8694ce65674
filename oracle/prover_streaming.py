"""ORACLE — TEST INFRASTRUCTURE ONLY.

The commitment rounds of `prove_cpu_basic` (src/cs/implementations/prover.rs:153-1802: witness, second stage, quotient, openings
at z / z*omega / 0) restated with every low-degree extension computed ONE COSET AT A TIME and dropped again, so that the bench's
own size (2^22 rows: 25 GB of witness LDE alone) fits a test's memory and time: what is compared with the HIP prover there are
the three oracle caps, every value at z, z*omega and 0 — i.e. everything that enters the transcript before the DEEP / FRI
rounds.  The arithmetic is oracle/prover.py's (the same C bulk operations on the same conventions: leaf index = coset * n + i,
a coset's subtree ends in cap_size / fri_lde_factor cap nodes, merkle_tree.rs:112-157, proof.rs:89-91); only the order of
evaluation differs.  tests/test_oracle_prover.py checks it against oracle/prover.py on a small circuit.

Circuit class: the SHA-256 bench's (hand-written gate kinds only, specialized lookups, Poseidon2 tree + Poseidon2 / Poseidon
transcript)."""
import numpy as np

import oracle as O
from oracle import prover as OP

P = O.P


def _coset_caps(mono, n, log_L, fri_lde, cap_size, threads, cosets=None, keep_trees=None):
    """Cap fragments of the oracle over the columns with monomials `mono`: {coset: [cap_size / fri_lde][4]}.  keep_trees (a dict):
    the coset's whole subtree (leaf hashes + node layers, 64 bytes per leaf) is kept there for the query round."""
    per = cap_size // fri_lde
    assert per >= 1, "a coset's subtree must end in at least one cap node"
    shifts = O.lde_coset_shifts(n.bit_length() - 1, log_L)
    out = {}
    vals = np.empty_like(mono)                                            # one buffer for every coset (no fresh pages per transform)
    for c in (range(fri_lde) if cosets is None else cosets):
        O.fft_batch(mono, int(shifts[c]), threads, out=vals)              # this coset of every column, bit-reversed
        tree = O.merkle_construct(vals, per, threads)
        out[c] = O.merkle_cap(tree, n, per)
        if keep_trees is not None:
            keep_trees[c] = tree
        del tree
    return out


def commitments_and_openings(circuit, setup_cap, fri_lde_factor=8, cap_size=16, threads=1, transcript_kind=1,
                             check_setup_cosets=(0,), cap_cosets=None, claimed_caps=None, rest_of_the_proof=False,
                             security_level=100, pow_bits=0):
    """Returns the dict of proof fields listed above plus "setup_cap_fragments" {coset: cap nodes} for `check_setup_cosets`
    (the transcript absorbs the caller's `setup_cap`; hashing all of the setup oracle again is the caller's choice).

    `cap_cosets` (with `claimed_caps` = {"witness_oracle_cap" | "stage_2_oracle_cap" | "quotient_oracle_cap": nodes} from the
    proof under check): hash only those cosets of the witness and second-stage oracles — at 2^23 rows a coset of the witness
    oracle is 10^8 permutations — and let the transcript absorb the CLAIMED caps of those two oracles; the recomputed subtree roots come back in
    out["cap_fragments"][name][coset] for the caller to compare with the claimed nodes.  Every challenge then is the one the
    proof under check used, and every value at z / z*omega / 0 is still recomputed from the witness alone.  The quotient
    oracle (one permutation per leaf) is always hashed in full.

    `rest_of_the_proof`: go on past the openings, still one coset at a time (prover.rs:1803-2266, fri/mod.rs:49-345): the DEEP
    accumulator of every FRI-domain point from a second pass over the cosets (all four oracles extended to one coset, the
    pointwise quotening_operation of prover.rs:2523-2706 on it, dropped again — 16 bytes per point stay), then `do_fri` in
    memory (base oracle and every intermediate oracle with its cap and fold challenge, final monomials), proof of work off,
    and the queries: the FRI openings with their paths in full; for the witness / second-stage / quotient (/ setup, on
    `check_setup_cosets`) oracles the Merkle path and the LEAF HASH of every query that lands in a coset whose subtree was
    built — the caller hashes the leaf elements of the proof under check and compares (the column values of a coset are gone
    by the time the indices are drawn; equal Poseidon2 leaf hashes over equal-length inputs are equal leaves up to a hash
    collision).  Adds "fri_base_oracle_cap", "fri_intermediate_oracles_caps", "final_fri_monomials", "query_indexes",
    "queries" [{"fri_queries": [...], "<oracle>_query": {"leaf_hash", "proof"} or None}]."""
    c = circuit
    n, log_n, V, q = c.n, c.log_n, c.num_vars, c.quotient_degree
    assert all(g.kind < 5 for g in c.gates) and not getattr(c, "specialized_gates", None) and not getattr(c, "num_witness_cols", 0)
    log_q = q.bit_length() - 1
    L = max(fri_lde_factor, q)
    log_L = L.bit_length() - 1
    log_fri = fri_lde_factor.bit_length() - 1
    has_lookup = c.lookup_reps > 0
    shifts = O.lde_coset_shifts(log_n, log_L)
    t = O.Transcript(transcript_kind)
    t.absorb_cap(np.asarray(setup_cap, dtype=np.uint64))                  # prover.rs:211
    pub_vals = [v for (_, _, v) in c.public_inputs]
    t.absorb(pub_vals)
    out = {"public_inputs": pub_vals, "cap_fragments": {}}
    trees = {"witness_query": {}, "stage_2_query": {}, "quotient_query": {}, "setup_query": {}}   # per oracle: {coset: subtree}
    keep = lambda name: trees[name] if rest_of_the_proof else None

    def full_cap(mono, log_lde_used, name=None, query_name=None):
        if cap_cosets is not None and name is not None:
            out["cap_fragments"][name] = _coset_caps(mono, n, log_lde_used, fri_lde_factor, cap_size, threads, cap_cosets,
                                                     keep_trees=keep(query_name))
            return np.asarray(claimed_caps[name], dtype=np.uint64).reshape(cap_size, 4)
        frag = _coset_caps(mono, n, log_lde_used, fri_lde_factor, cap_size, threads, keep_trees=keep(query_name))
        return np.concatenate([frag[k] for k in range(fri_lde_factor)])

    # setup: sigma || constants || tables  (polynomial_storage.rs:667-676)
    nS, nC = V, c.num_constant_cols
    setup_mono = O.ifft_batch(np.concatenate([c.sigmas, c.constants] + ([c.tables] if has_lookup else []), axis=0), 1, threads)
    out["setup_cap_fragments"] = _coset_caps(setup_mono, n, log_L, fri_lde_factor, cap_size, threads, check_setup_cosets,
                                             keep_trees=keep("setup_query"))
    # round 1: witness oracle, leaf = variables || multiplicities  (prover.rs:270-353)
    wit_mono = O.ifft_batch(np.concatenate([c.variables] + ([c.multiplicities] if has_lookup else []), axis=0), 1, threads)
    wit_cap = full_cap(wit_mono, log_L, "witness_oracle_cap", "witness_query")
    t.absorb_cap(wit_cap)
    # round 2: copy permutation + lookup polynomials on the main domain  (prover.rs:360-554)
    beta, gamma = t.challenge_ext(), t.challenge_ext()
    z_nat, partials_nat = OP.copy_perm_stage2(c.variables, c.sigmas, c.non_residues, log_n, q, beta, gamma, threads)
    n_partials = partials_nat.shape[0]
    stage2 = [z_nat[0], z_nat[1]] + [partials_nat[j][k] for j in range(n_partials) for k in range(2)]
    lbeta = lgamma = (0, 0)
    if has_lookup:
        lbeta, lgamma = t.challenge_ext(), t.challenge_ext()
        A_nat, B_nat = OP.lookup_polys(c.variables[c.num_gp_vars:], OP.lookup_table_id(c, c.constants), c.tables, c.multiplicities[0],
                                       c.lookup_reps, c.lookup_width, log_n, lbeta, lgamma, threads)
        stage2 += [A_nat[i][k] for i in range(c.lookup_reps) for k in range(2)] + [B_nat[0], B_nat[1]]
    s2_mono = O.ifft_batch(np.stack(stage2), 1, threads)
    del stage2, z_nat, partials_nat
    s2_cap = full_cap(s2_mono, log_L, "stage_2_oracle_cap", "stage_2_query")
    t.absorb_cap(s2_cap)
    # round 3: the quotient on the first q cosets, one coset at a time  (prover.rs:560-1495)
    alpha = t.challenge_ext()
    n_lookup_terms = c.lookup_reps + 1 if has_lookup else 0
    n_gate_terms = sum(g.reps * g.num_terms for g in c.gates)
    n_chunks = (V + q - 1) // q
    alphas = [(1, 0)]
    while len(alphas) < n_lookup_terms + n_gate_terms + 1 + n_chunks:
        alphas.append(OP.emul(alphas[-1], alpha))
    T = np.zeros((2, q * n), dtype=np.uint64)
    none = np.zeros(0, dtype=np.uint64)
    w, s, s2 = np.empty_like(wit_mono), np.empty_like(setup_mono), np.empty_like(s2_mono)   # reused by every coset below
    for cs in range(q):
        sh = int(shifts[cs])
        O.fft_batch(wit_mono, sh, threads, out=w)
        O.fft_batch(setup_mono, sh, threads, out=s)
        O.fft_batch(s2_mono, sh, threads, out=s2)
        o = 2 + 2 * n_partials
        Tc = OP.quotient(np.ascontiguousarray(w[:V]), np.ascontiguousarray(s[nS:nS + nC]), np.ascontiguousarray(s[:nS]),
                         np.ascontiguousarray(s2[0:2]), np.ascontiguousarray(s2[2:o]),
                         np.ascontiguousarray(s2[o:o + 2 * c.lookup_reps]) if has_lookup else none,
                         np.ascontiguousarray(s2[o + 2 * c.lookup_reps:]) if has_lookup else none,
                         np.ascontiguousarray(w[V]) if has_lookup else none,
                         np.ascontiguousarray(s[nS + nC:]) if has_lookup else none,
                         c, log_q, alphas, beta, gamma, lbeta, lgamma, threads, coset_begin=cs, coset_count=1)
        T[:, cs * n:(cs + 1) * n] = Tc
        del Tc
    qmono = O.ifft_batch(np.stack([O.bitreverse(T[0]), O.bitreverse(T[1])]), 7, threads)     # prover.rs:1405-1422
    del T
    assert qmono[0][-1] == 0 and qmono[1][-1] == 0, "unsatisfied (prover.rs:1425-1438)"
    chunks = []
    for j in range(q):
        chunks += [qmono[0][j * n:(j + 1) * n], qmono[1][j * n:(j + 1) * n]]
    q_mono = np.stack(chunks)
    q_cap = full_cap(q_mono, log_fri, None, "quotient_query")             # LDE only to fri_lde_factor (prover.rs:1473-1480)
    t.absorb_cap(q_cap)
    # round 4: openings from coset 0 (shift 7) of every committed column, in the reference's order  (prover.rs:1501-1802)
    z = t.challenge_ext()
    w0, w1 = O.barycentric_weights(log_n, 7, z)
    wit0, set0 = O.fft_batch(wit_mono, 7, threads, out=w), O.fft_batch(setup_mono, 7, threads, out=s)
    s20, q0 = O.fft_batch(s2_mono, 7, threads, out=s2), O.fft_batch(q_mono, 7, threads)
    ev_base = lambda col: O.barycentric_eval_base(col, w0, w1)
    ev_ext = lambda a, b: O.barycentric_eval_ext(a, b, w0, w1)
    vz = [ev_base(wit0[i]) for i in range(V)]
    vz += [ev_base(set0[nS + i]) for i in range(nC)]
    vz += [ev_base(set0[i]) for i in range(nS)]
    vz.append(ev_ext(s20[0], s20[1]))
    vz += [ev_ext(s20[2 + 2 * j], s20[3 + 2 * j]) for j in range(n_partials)]
    if has_lookup:
        vz.append(ev_base(wit0[V]))
        o = 2 + 2 * n_partials
        vz += [ev_ext(s20[o + 2 * i], s20[o + 2 * i + 1]) for i in range(c.lookup_reps + 1)]
        vz += [ev_base(set0[nS + nC + i]) for i in range(c.lookup_width + 1)]
    vz += [ev_ext(q0[2 * j], q0[2 * j + 1]) for j in range(q)]
    for v in vz:
        t.absorb(v)
    z_omega = OP.escale(z, O.omega(log_n))
    wz0, wz1 = O.barycentric_weights(log_n, 7, z_omega)
    vzo = [O.barycentric_eval_ext(s20[0], s20[1], wz0, wz1)]
    v0 = []
    if has_lookup:
        a0, a1 = O.barycentric_weights(log_n, 7, (0, 0))
        o = 2 + 2 * n_partials
        v0 = [O.barycentric_eval_ext(s20[o + 2 * i], s20[o + 2 * i + 1], a0, a1) for i in range(c.lookup_reps + 1)]
    out.update(witness_oracle_cap=wit_cap.tolist(), stage_2_oracle_cap=s2_cap.tolist(), quotient_oracle_cap=q_cap.tolist(),
               values_at_z=[list(v) for v in vz], values_at_z_omega=[list(v) for v in vzo], values_at_0=[list(v) for v in v0],
               challenges=dict(beta=beta, gamma=gamma, lbeta=lbeta, lgamma=lgamma, alpha=alpha, z=z))
    if not rest_of_the_proof:
        return out
    del wit0, set0, s20
    su, qv = s, q0
    for v in vzo:
        t.absorb(v)
    for v in v0:
        t.absorb(v)
    # round 5a: DEEP, one coset of the FRI domain at a time (prover.rs:1803-2067; sources in the order of the values at z)
    pub_tuples = []                                                       # grouped by opening point, first-seen order
    om = O.omega(log_n)
    for (col, row, val) in c.public_inputs:
        at = pow(om, row, P)
        for tup in pub_tuples:
            if tup[0] == at:
                tup[1].append((col, val))
                break
        else:
            pub_tuples.append((at, [(col, val)]))
    cch = t.challenge_ext()
    total_ch = len(vz) + 1 + len(v0) + sum(len(items) for _, items in pub_tuples)
    chs = [(1, 0), cch]                                                   # materialize_ext_challenge_powers (prover.rs:2374)
    while len(chs) < total_ch:
        chs.append(OP.emul(chs[-1], cch))
    N = n * fri_lde_factor
    d0, d1 = np.zeros(N, dtype=np.uint64), np.zeros(N, dtype=np.uint64)
    o = 2 + 2 * n_partials
    for cs in range(fri_lde_factor):
        sh = int(shifts[cs])
        O.fft_batch(wit_mono, sh, threads, out=w), O.fft_batch(setup_mono, sh, threads, out=su)
        O.fft_batch(s2_mono, sh, threads, out=s2), O.fft_batch(q_mono, sh, threads, out=qv)
        b_ = lambda col: (col, None)
        src = [b_(w[i]) for i in range(V)]
        src += [b_(su[nS + i]) for i in range(nC)]
        src += [b_(su[i]) for i in range(nS)]
        src.append((s2[0], s2[1]))
        src += [(s2[2 + 2 * j], s2[3 + 2 * j]) for j in range(n_partials)]
        if has_lookup:
            src.append(b_(w[V]))
            src += [(s2[o + 2 * i], s2[o + 2 * i + 1]) for i in range(c.lookup_reps + 1)]
            src += [b_(su[nS + nC + i]) for i in range(c.lookup_width + 1)]
        src += [(qv[2 * j], qv[2 * j + 1]) for j in range(q)]
        assert len(src) == len(vz)
        e0, e1 = d0[cs * n:(cs + 1) * n], d1[cs * n:(cs + 1) * n]        # views: accumulated in place
        first, off = cs * n, 0
        O.deep_quotient_accumulate_range(src, vz, chs[off:off + len(src)], z, log_n, log_fri, first, e0, e1, threads)
        off += len(src)
        O.deep_quotient_accumulate_range([(s2[0], s2[1])], vzo, chs[off:off + 1], z_omega, log_n, log_fri, first, e0, e1, threads)
        off += 1
        if has_lookup:
            s0 = [(s2[o + 2 * i], s2[o + 2 * i + 1]) for i in range(c.lookup_reps + 1)]
            O.deep_quotient_accumulate_range(s0, v0, chs[off:off + len(s0)], (0, 0), log_n, log_fri, first, e0, e1, threads)
            off += len(s0)
        for at, items in pub_tuples:
            sp = [b_(w[col]) for col, _ in items]
            vp = [(val, 0) for _, val in items]
            O.deep_quotient_accumulate_range(sp, vp, chs[off:off + len(sp)], (at, 0), log_n, log_fri, first, e0, e1, threads)
            off += len(sp)
        assert off == len(chs)
        del src
    # round 5b: FRI in memory (prover.rs:2075-2105, fri/mod.rs:49-345): 16 bytes per point
    new_pow, num_queries, sched, final_degree = O.fri_schedule(security_level, cap_size, pow_bits, log_fri, log_n)
    assert new_pow == 0, "the streaming restatement runs without proof of work (as the benches do)"
    fri = O.do_fri(d0, d1, log_fri, sched, cap_size, t, threads)
    del d0, d1
    # round 6: queries (prover.rs:2161-2266)
    qi = O.QueryIndexer(log_n, log_fri)
    per = cap_size // fri_lde_factor
    queries, indexes = [], []

    def open_base(name, idx):
        tree = trees[name].get(idx >> log_n)
        if tree is None:
            return None
        leaf_hash, path = O.merkle_proof(tree, n, per, idx & (n - 1))
        return {"leaf_hash": [int(x) for x in leaf_hash], "proof": [[int(x) for x in p] for p in path]}

    for _ in range(num_queries):
        idx = qi.next(t)
        indexes.append(idx)
        qd = {name: open_base(name, idx) for name in trees}
        qd["fri_queries"] = []
        f_idx, ln = idx, N
        for i, k in enumerate(sched):
            E = 1 << k
            j = f_idx >> k
            s0, s1 = fri["sources"][i]
            leaf = np.concatenate([s0[j * E:(j + 1) * E], s1[j * E:(j + 1) * E]])
            _, path = O.merkle_proof(fri["trees"][i], ln >> k, cap_size, j)
            qd["fri_queries"].append({"leaf_elements": [int(x) for x in leaf], "proof": [[int(x) for x in p] for p in path]})
            f_idx >>= k
            ln >>= k
        queries.append(qd)
    fd = fri["final_degree"]
    out.update(fri_base_oracle_cap=fri["caps"][0].tolist(), fri_intermediate_oracles_caps=[cap.tolist() for cap in fri["caps"][1:]],
               final_fri_monomials=[fri["final_monomials"][0][:fd].tolist(), fri["final_monomials"][1][:fd].tolist()],
               query_indexes=indexes, queries=queries, deep_challenge=cch, fri_challenges=fri["challenges"])
    return out


def compare_rest_of_the_proof(proof, got):
    """The DEEP / FRI / query part of a proof (the dict of proof_format.parse / oracle/prover.py) against commitments_and_openings(...,
    rest_of_the_proof=True): FRI caps, final monomials and FRI query openings byte for byte; for the four base oracles the Merkle
    path byte for byte and the Poseidon2 hash of the opened leaf against the leaf hash of the restatement's own tree, for every
    query whose coset the restatement built.  Returns how many base-oracle openings were compared."""
    for k in ("fri_base_oracle_cap", "fri_intermediate_oracles_caps", "final_fri_monomials"):
        assert proof[k] == got[k], k
    assert len(proof["queries_per_fri_repetition"]) == len(got["queries"])
    compared = 0
    for qp, qo in zip(proof["queries_per_fri_repetition"], got["queries"]):
        assert qp["fri_queries"] == qo["fri_queries"], "FRI query openings"
        for name in ("witness_query", "stage_2_query", "quotient_query", "setup_query"):
            if qo[name] is None:
                continue
            assert qp[name]["proof"] == qo[name]["proof"], name + " path"
            assert [int(x) for x in O.hash_leaf(qp[name]["leaf_elements"])] == qo[name]["leaf_hash"], name + " leaf"
            compared += 1
    return compared
