"""ORACLE — TEST INFRASTRUCTURE ONLY.

Restatement of `Verifier::verify` (src/cs/implementations/verifier.rs:888-2524) for the circuit class handled by
oracle/prover.py (and the HIP prover): transcript replay, lookup sumcheck, quotient identity at z, and per query the four
base-oracle Merkle paths, the DEEP simulation and the FRI chain down to the final monomials.  Independent of the prover
code path (python integers + the pinned hashing/transcript primitives), so "verifier accepts" is a meaningful check of
the quotient assembly that the golden proof cannot pin."""
import numpy as np

import oracle as O
from oracle.prover import emul, eadd, esub, escale, einv, epow

P = O.P


class VerificationKey:
    def __init__(self, circuit, setup_cap, fri_lde_factor, cap_size):
        c = circuit
        self.log_n, self.n = c.log_n, c.n
        self.num_vars, self.num_gp_vars = c.num_vars, c.num_gp_vars
        self.num_witness_cols = int(getattr(c, "num_witness_cols", 0))      # CSGeometry::num_witness_columns
        self.num_constant_cols = c.num_constant_cols
        self.lookup_reps, self.lookup_width = c.lookup_reps, c.lookup_width
        self.table_id_col = c.table_id_col
        # LookupParameters::UseSpecializedColumnsWithTableIdAsVariable (verifier.rs:675-678, 1397-1470): width + 1 variable columns per
        # sub-argument, no table-id constant (vk.fixed_parameters.table_ids_column_idxes is empty)
        self.table_id_as_variable = bool(getattr(c, "table_id_as_variable", False))
        self.gates = c.gates
        self.specialized_gates = list(getattr(c, "specialized_gates", []) or [])   # (evaluator_data.rs:190-236)
        self.quotient_degree = c.quotient_degree
        self.non_residues = list(c.non_residues)
        self.public_input_locations = [(col, row) for (col, row, _) in c.public_inputs]
        self.setup_cap = np.array(setup_cap, dtype=np.uint64)
        self.fri_lde_factor, self.cap_size = fri_lde_factor, cap_size


class _Gate:
    pass


def vk_from_reference_geometry(geometry, setup_cap, general_gates, specialized_gates, non_residues, fri_lde_factor, cap_size):
    """A VerificationKey for verify() out of the reference's own layout (vk.fixed_parameters as cut into the golden fixture):
    evaluator names in gate_idx order, [(name, repetitions)] for the gates over specialized columns; paths from the
    `selectors_placement` tree (left = the constant)."""
    from oracle.gates import EVALUATORS
    from oracle.golden_quotient import _paths
    vk = VerificationKey.__new__(VerificationKey)
    n = geometry["domain_size"]
    from oracle.golden_quotient import lookup_parameters
    lk, w, reps, cps = lookup_parameters(geometry)
    vgp = geometry["num_variable_columns"]
    paths = {}
    _paths(geometry["selectors_placement"], [], paths)
    vk.gates, vk.specialized_gates = [], []
    for idx, name in enumerate(general_gates):
        width, reps_fn, n_shared, cstride, n_terms, _ = EVALUATORS[name]
        g = _Gate()
        g.name, g.kind, g.path, g.num_terms = name, 0, list(paths.get(idx, [])), n_terms
        g.reps, g.var_stride, g.const_stride = reps_fn(vgp, geometry["num_constant_columns"]), width, cstride
        vk.gates.append(g)
    for name, r in specialized_gates:
        g = _Gate()
        g.name, g.kind, g.path, g.reps = name, 0, [], r
        # per-repetition constants (share_constants = false) of an evaluator that reads them inside evaluate_once: its own columns
        # behind the table-id column (evaluator_data.rs:196-238); 0 for the BooleanConstraintGate of the golden proof
        g.var_stride, g.const_stride, g.num_terms = EVALUATORS[name][0], EVALUATORS[name][3], EVALUATORS[name][4]
        vk.specialized_gates.append(g)
    vk.log_n, vk.n = n.bit_length() - 1, n
    vk.num_gp_vars = vgp
    vk.num_vars = vgp + cps * reps + sum(g.reps * g.var_stride for g in vk.specialized_gates)
    vk.num_constant_cols = (geometry["num_constant_columns"] + geometry["extra_constant_polys_for_selectors"] + len(geometry["table_ids_column_idxes"])
                            + sum(g.reps * g.const_stride for g in vk.specialized_gates))
    vk.lookup_reps, vk.lookup_width = reps, w
    vk.table_id_as_variable = cps != w
    vk.table_id_col = geometry["table_ids_column_idxes"][0] if (lk and not vk.table_id_as_variable) else 0
    vk.quotient_degree = geometry["quotient_degree"]
    vk.non_residues = list(non_residues)
    vk.public_input_locations = [tuple(x) for x in geometry["public_inputs_locations"]]
    vk.setup_cap = np.array(setup_cap, dtype=np.uint64)
    vk.fri_lde_factor, vk.cap_size = fri_lde_factor, cap_size
    return vk


def _program_terms_ext(prog, var, con, wit):
    """An op list (relations: (op, dst, (kind, index), (kind, index)); kinds 0 variable, 1 witness, 2 constant column, 3 temporary, 4
    field constant; ops 1 add, 2 double, 3 sub, 4 negate, 5 mul, 6 square, 7 inverse — Relation / Index of gpu_synthesizer/mod.rs:113-133)
    evaluated over F_p^2 values."""
    tmp = {}

    def get(ref):
        k, i = ref
        if k == 0: return var[i]
        if k == 1: return wit[i]
        if k == 2: return con[i]
        if k == 3: return tmp[i]
        return (prog.values[i] % P, 0)
    for op, dst, a, b in prog.relations:
        x = get(a)
        if op == 1: r = eadd(x, get(b))
        elif op == 2: r = eadd(x, x)
        elif op == 3: r = esub(x, get(b))
        elif op == 4: r = esub((0, 0), x)
        elif op == 5: r = emul(x, get(b))
        elif op == 6: r = emul(x, x)
        else: r = einv(x)
        tmp[dst] = r
    return [get(w) for w in prog.writes]


def _gate_terms_at(vk, var, con, wit=()):
    """[(selector, [terms...])] with var/con/wit = F_p^2 values of the variable / constant / witness polys at z."""
    out = []
    one = (1, 0)
    for g in vk.gates:
        if g.num_terms == 0:
            continue
        sel = one
        for b, bit in enumerate(g.path):
            sel = emul(sel, con[b] if bit else esub(one, con[b]))
        d = len(g.path)
        terms = []
        for r in range(g.reps):
            vb, cb = r * g.var_stride, d + r * g.const_stride
            if g.kind == 1:      # ConstantsAllocator (constant_allocator.rs:107-126)
                terms.append(esub(var[vb], con[cb]))
            elif g.kind == 2:    # FMA without constant (fma_gate_without_constant.rs:96-126)
                t = eadd(emul(var[vb + 2], con[d + 1]), emul(con[d], emul(var[vb], var[vb + 1])))
                terms.append(esub(t, var[vb + 3]))
            elif g.kind == 3:    # Reduction<4> (reduction_gate.rs:103-126)
                t = (0, 0)
                for k in range(4):
                    t = eadd(t, emul(var[vb + k], con[d + k]))
                terms.append(esub(t, var[vb + 4]))
            else:                # any other evaluator, by name: the golden-pinned formulas of oracle/gates.py
                from oracle.gates import EVALUATORS
                if g.name not in EVALUATORS:     # a host's own evaluator: its op list over F_p^2 (the verifier's view of a capture)
                    ws = getattr(g, "wit_stride", 0)
                    terms.extend(_program_terms_ext(g.program, var[vb:], con[cb:], wit[r * ws:] if ws else ()))
                    continue
                width, fn = EVALUATORS[g.name][0], EVALUATORS[g.name][5]
                ws = getattr(g, "wit_stride", 0)
                if ws:       # the evaluator reads witness columns, relative to its repetition (per_chunk_offset.witnesses_offset)
                    terms.extend(fn(var[vb:vb + width], con[cb:], wit[r * ws:]))
                else:
                    terms.extend(fn(var[vb:vb + width], con[cb:]))
        out.append((sel, terms))
    return out


def verify(vk, proof, verbose=False, transcript_kind=1, partial_queries=False, pow_runner=1):
    """partial_queries: the proof carries only the FIRST k of the query openings (the golden fixture keeps 6 of 100);
    indices are drawn in order, so the first k can be checked on their own."""
    def fail(msg):
        if verbose:
            print("verify:", msg)
        return False

    log_n, n, V, q = vk.log_n, vk.n, vk.num_vars, vk.quotient_degree
    cfg = proof["proof_config"]
    fri_lde, cap_size = cfg["fri_lde_factor"], cfg["merkle_tree_cap_size"]
    if fri_lde != vk.fri_lde_factor or cap_size != vk.cap_size:
        return fail("proof config does not match the VK")
    log_fri = fri_lde.bit_length() - 1
    LOGN = log_n + log_fri
    has_lookup = vk.lookup_reps > 0
    n_chunks = (V + q - 1) // q
    n_partials = n_chunks - 1
    nC = vk.num_constant_cols
    # ---- transcript replay (verifier.rs:924-1076)
    from oracle.prover import hashing_layer
    H = hashing_layer({3: 2, 4: 3}.get(transcript_kind, 1))
    t = H.Transcript(transcript_kind)
    t.absorb_cap(vk.setup_cap)
    t.absorb(proof["public_inputs"])
    t.absorb_cap(np.array(proof["witness_oracle_cap"], dtype=np.uint64))
    beta, gamma = t.challenge_ext(), t.challenge_ext()
    lbeta = lgamma = (0, 0)
    if has_lookup:
        lbeta, lgamma = t.challenge_ext(), t.challenge_ext()
    t.absorb_cap(np.array(proof["stage_2_oracle_cap"], dtype=np.uint64))
    alpha = t.challenge_ext()
    t.absorb_cap(np.array(proof["quotient_oracle_cap"], dtype=np.uint64))
    z = t.challenge_ext()
    vz = [tuple(v) for v in proof["values_at_z"]]
    vzo = [tuple(v) for v in proof["values_at_z_omega"]]
    v0 = [tuple(v) for v in proof["values_at_0"]]
    for grp in (vz, vzo, v0):
        for v in grp:
            t.absorb(v)
    n_lookup_terms = vk.lookup_reps + 1 if has_lookup else 0
    n_lookup_polys = (vk.lookup_reps + 2 + vk.lookup_width + 1) if has_lookup else 0
    Wc = getattr(vk, "num_witness_cols", 0)
    if len(vz) != V + Wc + nC + V + 1 + n_partials + n_lookup_polys + q:
        return fail("unexpected number of openings at z")
    if len(vzo) != 1 or len(v0) != n_lookup_terms:
        return fail("unexpected number of openings at z*omega / 0")
    # ---- split the openings (verifier.rs:1150-1206)
    it = iter(vz)
    take = lambda k: [next(it) for _ in range(k)]
    var_z, wit_z, con_z, sig_z = take(V), take(Wc), take(nC), take(V)       # variables, witness, constants, sigmas (verifier.rs:1150-1206)
    z_at_z = next(it)
    part_z = take(n_partials)
    mult_z = take(1) if has_lookup else []
    A_z = take(vk.lookup_reps) if has_lookup else []
    B_z = take(1) if has_lookup else []
    tab_z = take(vk.lookup_width + 1) if has_lookup else []
    qch_z = take(q)
    z_at_zo = vzo[0]
    # ---- challenges for the quotient terms (prover.rs:599-625 / verifier.rs:1000-1060)
    n_gate_terms = sum(g.reps * g.num_terms for g in vk.gates)
    n_spec_terms = sum(g.reps * g.num_terms for g in vk.specialized_gates)
    total_terms = n_lookup_terms + n_spec_terms + n_gate_terms + 1 + n_chunks      # lookup | specialized | general | L1 | chunks
    alphas = [(1, 0)]
    while len(alphas) < total_terms:
        alphas.append(emul(alphas[-1], alpha))
    a_lookup, a_spec = alphas[:n_lookup_terms], alphas[n_lookup_terms:n_lookup_terms + n_spec_terms]
    a_gates = alphas[n_lookup_terms + n_spec_terms:n_lookup_terms + n_spec_terms + n_gate_terms]
    a_rest = alphas[n_lookup_terms + n_spec_terms + n_gate_terms:]
    one = (1, 0)
    T = (0, 0)
    if has_lookup:
        # sumcheck (verifier.rs:1236-1256)
        sa = (0, 0)
        for a in v0[:vk.lookup_reps]:
            sa = eadd(sa, a)
        if sa != v0[vk.lookup_reps]:
            return fail("lookup sumcheck is invalid")
        gp = [one]
        for _ in range(vk.lookup_width):
            gp.append(emul(gp[-1], lgamma))
        tid_var = getattr(vk, "table_id_as_variable", False)
        cps = vk.lookup_width + (1 if tid_var else 0)     # specialized_columns_per_subargument (verifier.rs:1402-1409)
        for i in range(vk.lookup_reps):
            d = lbeta
            for j in range(cps):
                d = eadd(d, emul(gp[j], var_z[vk.num_gp_vars + i * cps + j]))
            if not tid_var:                               # witness_columns.chain(table_id) (verifier.rs:1447-1464)
                d = eadd(d, emul(gp[vk.lookup_width], con_z[vk.table_id_col]))
            T = eadd(T, emul(esub(emul(A_z[i], d), one), a_lookup[i]))
        d = lbeta
        for j in range(vk.lookup_width + 1):
            d = eadd(d, emul(gp[j], tab_z[j]))
        T = eadd(T, emul(esub(emul(B_z[0], d), mult_z[0]), a_lookup[vk.lookup_reps]))
    # gates over specialized columns: no selector, their own variable columns after the lookup ones (verifier.rs:1560-1638)
    from oracle.gates import EVALUATORS
    col, off = vk.num_gp_vars + vk.lookup_reps * (vk.lookup_width + (1 if getattr(vk, "table_id_as_variable", False) else 0)), 0
    ccol = vk.num_constant_cols - sum(g.reps * g.const_stride for g in vk.specialized_gates)   # their constants: the last columns
    for g in vk.specialized_gates:
        width, fn = EVALUATORS[g.name][0], EVALUATORS[g.name][5]
        for r in range(g.reps):
            cons = con_z[ccol + r * g.const_stride: ccol + (r + 1) * g.const_stride]   # its own per repetition (verifier.rs:1609-1633)
            for term in fn(var_z[col + r * g.var_stride: col + r * g.var_stride + width], cons):
                T = eadd(T, emul(term, a_spec[off]))
                off += 1
        col += g.reps * g.var_stride
        ccol += g.reps * g.const_stride
    if off != n_spec_terms or col != vk.num_vars or ccol != vk.num_constant_cols:
        return fail("specialized gate bookkeeping")
    # gates (verifier.rs:1640-1720)
    off = 0
    for sel, terms in _gate_terms_at(vk, var_z, con_z, wit_z):
        acc = (0, 0)
        for term in terms:
            acc = eadd(acc, emul(term, a_gates[off]))
            off += 1
        T = eadd(T, emul(acc, sel))
    if off != n_gate_terms:
        return fail("gate challenge bookkeeping")
    # (z(x) - 1) * L1~ and the copy-permutation chain (verifier.rs:1722-1790)
    z_n = epow(z, n)
    vanishing = esub(z_n, one)
    l1 = emul(vanishing, einv(esub(z, one)))
    T = eadd(T, emul(emul(esub(z_at_z, one), l1), a_rest[0]))
    lhs_list = part_z + [z_at_zo]
    rhs_list = [z_at_z] + part_z
    for j in range(n_chunks):
        lhs, rhs = lhs_list[j], rhs_list[j]
        for cidx in range(j * q, min((j + 1) * q, V)):
            lhs = emul(lhs, eadd(eadd(emul(sig_z[cidx], beta), var_z[cidx]), gamma))
            rhs = emul(rhs, eadd(eadd(emul(escale(z, vk.non_residues[cidx]), beta), var_z[cidx]), gamma))
        T = eadd(T, emul(esub(lhs, rhs), a_rest[1 + j]))
    t_chunks, pw = (0, 0), one
    for el in qch_z:
        t_chunks = eadd(t_chunks, emul(el, pw))
        pw = emul(pw, z_n)
    if T != emul(t_chunks, vanishing):
        return fail("invalid quotient at z")
    # ---- DEEP / FRI challenges (verifier.rs:1819-1983)
    cch = t.challenge_ext()
    new_pow, num_queries, sched, final_degree = O.fri_schedule(cfg["security_level"], cap_size, cfg["pow_bits"], log_fri, log_n)
    caps = [proof["fri_base_oracle_cap"]] + proof["fri_intermediate_oracles_caps"]
    if len(caps) != len(sched):
        return fail("unexpected number of FRI oracles")
    fri_ch = []
    for cap in caps:
        t.absorb_cap(np.array(cap, dtype=np.uint64))
        fri_ch.append(t.challenge_ext())
    fm = proof["final_fri_monomials"]
    if len(fm[0]) != final_degree or len(fm[1]) != final_degree:
        return fail("unexpected final monomials length")
    t.absorb(fm[0])
    t.absorb(fm[1])
    if new_pow:          # verifier.rs:1957-1983: the nonce must solve the puzzle seeded by the transcript, then it is absorbed
        from oracle.prover import pow_seed, pow_ok
        nonce = int(proof["pow_challenge"])
        if not pow_ok(pow_seed(t), new_pow, nonce, pow_runner):     # POW::verify_from_field_elements (pow.rs:16-31)
            return fail("invalid proof of work")
        t.absorb([nonce & 0xFFFFFFFF, nonce >> 32])
    if len(proof["queries_per_fri_repetition"]) != num_queries and not (
            partial_queries and 0 < len(proof["queries_per_fri_repetition"]) < num_queries):
        return fail("unexpected number of queries")
    om = O.omega(log_n)
    pub_tuples = []
    for (col, row), val in zip(vk.public_input_locations, proof["public_inputs"]):
        at = pow(om, row, P)
        for tup in pub_tuples:
            if tup[0] == at:
                tup[1].append((col, val))
                break
        else:
            pub_tuples.append((at, [(col, val)]))
    total_ch = len(vz) + 1 + len(v0) + sum(len(s) for _, s in pub_tuples)
    chs = [(1, 0), cch]
    while len(chs) < total_ch:
        chs.append(emul(chs[-1], cch))
    roots = O.twiddles(LOGN, inverse=True)
    qi = H.QueryIndexer(log_n, log_fri)
    z_omega = escale(z, om)
    base = lambda l: [(e, 0) for e in l]
    ext = lambda l: [(l[i], l[i + 1]) for i in range(0, len(l), 2)]
    wl = V + Wc + (1 if has_lookup else 0)
    s2l = 2 * (1 + n_partials) + (2 * (vk.lookup_reps + 1) if has_lookup else 0)
    sul = V + nC + ((vk.lookup_width + 1) if has_lookup else 0)
    named_caps = {"witness_query": proof["witness_oracle_cap"], "stage_2_query": proof["stage_2_oracle_cap"],
                  "quotient_query": proof["quotient_oracle_cap"], "setup_query": vk.setup_cap.tolist()}
    widths = {"witness_query": wl, "stage_2_query": s2l, "quotient_query": 2 * q, "setup_query": sul}
    depth = (n * fri_lde // cap_size).bit_length() - 1
    for query in proof["queries_per_fri_repetition"]:
        idx = qi.next(t)
        for name, cap in named_caps.items():
            le, path = query[name]["leaf_elements"], query[name]["proof"]
            if len(le) != widths[name] or len(path) != depth:
                return fail("bad opening shape for %s" % name)
            if not H.merkle_verify(np.array(path, dtype=np.uint64).reshape(-1, 4), np.array(cap, dtype=np.uint64),
                                   H.hash_leaf(le), idx):
                return fail("Merkle path of %s does not verify" % name)
        W, S2, Q_, SU = (query[k]["leaf_elements"] for k in ("witness_query", "stage_2_query", "quotient_query", "setup_query"))
        # source order of verifier.rs:2233-2290 == opening order: vars, constants, sigmas, z, partials, mult, A, B, tables, quotient
        src = base(W[:V + Wc]) + base(SU[V:V + nC]) + base(SU[:V]) + ext(S2[0:2]) + ext(S2[2:2 + 2 * n_partials])
        if has_lookup:
            o = 2 + 2 * n_partials
            src += base(W[V + Wc:V + Wc + 1]) + ext(S2[o:o + 2 * vk.lookup_reps]) + ext(S2[o + 2 * vk.lookup_reps:]) + base(SU[V + nC:])
        src += ext(Q_)
        x = pow(O.omega(LOGN), O.bitrev(idx, LOGN), P) * 7 % P

        def quot(srcs, vals, at, ws):
            acc = (0, 0)
            for s_, v_, w_ in zip(srcs, vals, ws):
                acc = eadd(acc, emul(w_, esub(s_, v_)))
            return emul(acc, einv(esub((x, 0), at)))
        o = 0
        h = quot(src, vz, z, chs[o:o + len(vz)]); o += len(vz)
        h = eadd(h, quot(ext(S2[0:2]), vzo, z_omega, chs[o:o + 1])); o += 1
        if has_lookup:
            oo = 2 + 2 * n_partials
            h = eadd(h, quot(ext(S2[oo:]), v0, (0, 0), chs[o:o + len(v0)])); o += len(v0)
        for at, items in pub_tuples:
            h = eadd(h, quot([(W[col], 0) for col, _ in items], [(val, 0) for _, val in items], (at, 0), chs[o:o + len(items)]))
            o += len(items)
        # FRI chain (verifier.rs:2387-2519)
        cur, fidx, kappa, ln = h, idx, O.inv(7), n * fri_lde
        if len(query["fri_queries"]) != len(sched):
            return fail("unexpected number of FRI openings")
        for layer, (k, fq) in enumerate(zip(sched, query["fri_queries"])):
            m = 1 << k
            sub_, tree_idx = fidx % m, fidx >> k
            le = fq["leaf_elements"]
            if len(le) != 2 * m:
                return fail("bad FRI leaf size")
            c0, c1 = np.array(le[:m], dtype=np.uint64), np.array(le[m:], dtype=np.uint64)
            if (int(c0[sub_]), int(c1[sub_])) != cur:
                return fail("FRI layer %d: carried value is not in the leaf" % layer)
            path = np.array(fq["proof"], dtype=np.uint64).reshape(-1, 4)
            if path.shape[0] != ((ln >> k) // cap_size).bit_length() - 1:
                return fail("bad FRI path length")
            if not H.merkle_verify(path, np.array(caps[layer], dtype=np.uint64), H.hash_leaf(le), tree_idx):
                return fail("FRI layer %d: Merkle path does not verify" % layer)
            chal, start = fri_ch[layer], tree_idx * m
            for _ in range(k):
                half = c0.size // 2
                c0, c1 = O.fri_fold(c0, c1, roots[(start >> 1):(start >> 1) + half], kappa, chal)
                start >>= 1
                chal = emul(chal, chal)
                kappa = kappa * kappa % P
            cur, fidx, ln = (int(c0[0]), int(c1[0])), tree_idx, ln >> k
        xx = x
        for _ in range(sum(sched)):
            xx = xx * xx % P
        acc = (0, 0)
        for cc in reversed(list(zip(fm[0], fm[1]))):
            acc = eadd(escale(acc, xx), cc)
        if acc != cur:
            return fail("final FRI fold does not match the final monomials")
    return True
