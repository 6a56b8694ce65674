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


def _coset_caps(mono, n, log_L, fri_lde, cap_size, threads, cosets=None):
    """Cap fragments of the oracle over the columns with monomials `mono`: {coset: [cap_size / fri_lde][4]}."""
    per = cap_size // fri_lde
    assert per >= 1, "a coset's subtree must end in at least one cap node"
    shifts = O.lde_coset_shifts(n.bit_length() - 1, log_L)
    out = {}
    for c in (range(fri_lde) if cosets is None else cosets):
        vals = O.fft_batch(mono, int(shifts[c]), threads)                 # this coset of every column, bit-reversed
        tree = O.merkle_construct(vals, per, threads)
        out[c] = O.merkle_cap(tree, n, per)
        del vals, tree
    return out


def commitments_and_openings(circuit, setup_cap, fri_lde_factor=8, cap_size=16, threads=1, transcript_kind=1,
                             check_setup_cosets=(0,), cap_cosets=None, claimed_caps=None):
    """Returns the dict of proof fields listed above plus "setup_cap_fragments" {coset: cap nodes} for `check_setup_cosets`
    (the transcript absorbs the caller's `setup_cap`; hashing all of the setup oracle again is the caller's choice).

    `cap_cosets` (with `claimed_caps` = {"witness_oracle_cap" | "stage_2_oracle_cap" | "quotient_oracle_cap": nodes} from the
    proof under check): hash only those cosets of the witness and second-stage oracles — at 2^23 rows a coset of the witness
    oracle is 10^8 permutations — and let the transcript absorb the CLAIMED caps of those two oracles; the recomputed subtree roots come back in
    out["cap_fragments"][name][coset] for the caller to compare with the claimed nodes.  Every challenge then is the one the
    proof under check used, and every value at z / z*omega / 0 is still recomputed from the witness alone.  The quotient
    oracle (one permutation per leaf) is always hashed in full."""
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

    def full_cap(mono, log_lde_used, name=None):
        if cap_cosets is not None and name is not None:
            out["cap_fragments"][name] = _coset_caps(mono, n, log_lde_used, fri_lde_factor, cap_size, threads, cap_cosets)
            return np.asarray(claimed_caps[name], dtype=np.uint64).reshape(cap_size, 4)
        frag = _coset_caps(mono, n, log_lde_used, fri_lde_factor, cap_size, threads)
        return np.concatenate([frag[k] for k in range(fri_lde_factor)])

    # setup: sigma || constants || tables  (polynomial_storage.rs:667-676)
    nS, nC = V, c.num_constant_cols
    setup_mono = O.ifft_batch(np.concatenate([c.sigmas, c.constants] + ([c.tables] if has_lookup else []), axis=0), 1, threads)
    out["setup_cap_fragments"] = _coset_caps(setup_mono, n, log_L, fri_lde_factor, cap_size, threads, check_setup_cosets)
    # round 1: witness oracle, leaf = variables || multiplicities  (prover.rs:270-353)
    wit_mono = O.ifft_batch(np.concatenate([c.variables] + ([c.multiplicities] if has_lookup else []), axis=0), 1, threads)
    wit_cap = full_cap(wit_mono, log_L, "witness_oracle_cap")
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
    s2_cap = full_cap(s2_mono, log_L, "stage_2_oracle_cap")
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
    for cs in range(q):
        sh = int(shifts[cs])
        w = O.fft_batch(wit_mono, sh, threads)
        s = O.fft_batch(setup_mono, sh, threads)
        s2 = O.fft_batch(s2_mono, sh, threads)
        o = 2 + 2 * n_partials
        Tc = OP.quotient(np.ascontiguousarray(w[:V]), np.ascontiguousarray(s[nS:nS + nC]), np.ascontiguousarray(s[:nS]),
                         np.ascontiguousarray(s2[0:2]), np.ascontiguousarray(s2[2:o]),
                         np.ascontiguousarray(s2[o:o + 2 * c.lookup_reps]) if has_lookup else none,
                         np.ascontiguousarray(s2[o + 2 * c.lookup_reps:]) if has_lookup else none,
                         np.ascontiguousarray(w[V]) if has_lookup else none,
                         np.ascontiguousarray(s[nS + nC:]) if has_lookup else none,
                         c, log_q, alphas, beta, gamma, lbeta, lgamma, threads, coset_begin=cs, coset_count=1)
        T[:, cs * n:(cs + 1) * n] = Tc
        del w, s, s2, Tc
    qmono = O.ifft_batch(np.stack([O.bitreverse(T[0]), O.bitreverse(T[1])]), 7, threads)     # prover.rs:1405-1422
    del T
    assert qmono[0][-1] == 0 and qmono[1][-1] == 0, "unsatisfied (prover.rs:1425-1438)"
    chunks = []
    for j in range(q):
        chunks += [qmono[0][j * n:(j + 1) * n], qmono[1][j * n:(j + 1) * n]]
    q_mono = np.stack(chunks)
    q_cap = full_cap(q_mono, log_fri)                                     # LDE only to fri_lde_factor (prover.rs:1473-1480)
    t.absorb_cap(q_cap)
    # round 4: openings from coset 0 (shift 7) of every committed column, in the reference's order  (prover.rs:1501-1802)
    z = t.challenge_ext()
    w0, w1 = O.barycentric_weights(log_n, 7, z)
    wit0, set0 = O.fft_batch(wit_mono, 7, threads), O.fft_batch(setup_mono, 7, threads)
    s20, q0 = O.fft_batch(s2_mono, 7, threads), O.fft_batch(q_mono, 7, threads)
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
    return out
