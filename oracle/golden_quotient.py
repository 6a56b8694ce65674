"""ORACLE — TEST INFRASTRUCTURE ONLY.

The quotient identity at z of `Verifier::verify` (verifier.rs:1090-1810) for an arbitrary gate set described by a
VerificationKey in the reference's own layout (vk.json): lookup terms, gates over specialized columns, gates over
general-purpose columns behind their selector paths, (z(x)-1)*L_1 and the copy-permutation chain, against
sum_i t_i(z) z^{n i} * (z^n - 1).  Used to turn the reference's golden proof into a known-answer test of exactly the
conventions the fixture's Merkle/FRI checks cannot reach (alpha order, selector paths, copy-permutation / lookup terms,
gate formulas) — the same conventions oracle/prover.py, oracle/verifier.py and the HIP prover implement.
"""
from oracle.gates import EVALUATORS
from oracle.prover import P, eadd, emul, esub, escale, einv, epow

ONE, ZERO = (1, 0), (0, 0)


def _paths(node, prefix, out):
    if "GateOnly" in node:
        out[node["GateOnly"]["gate_idx"]] = list(prefix)
    elif "Fork" in node:
        _paths(node["Fork"]["left"], prefix + [True], out)      # setup.rs:1467-1480: left = the constant itself
        _paths(node["Fork"]["right"], prefix + [False], out)    # right = 1 - constant


def geometry_from_vk_json(vk):
    """vk.json layout (VerificationKey, verifier.rs:52-79) -> the `geometry` dict used here and in the golden fixture."""
    fp = vk["fixed_parameters"]
    return {"domain_size": fp["domain_size"],
            "num_variable_columns": fp["parameters"]["num_columns_under_copy_permutation"],
            "num_witness_columns": fp["parameters"]["num_witness_columns"],
            "num_constant_columns": fp["parameters"]["num_constant_columns"],
            "extra_constant_polys_for_selectors": fp["extra_constant_polys_for_selectors"],
            "lookup": fp["lookup_parameters"], "quotient_degree": fp["quotient_degree"],
            "public_inputs_locations": fp["public_inputs_locations"],
            "table_ids_column_idxes": fp["table_ids_column_idxes"], "total_tables_len": fp["total_tables_len"],
            "selectors_placement": fp["selectors_placement"]}


def lookup_parameters(geometry):
    """(parameters dict or None, width, num_repetitions, variable columns per sub-argument) of vk.fixed_parameters.lookup_parameters:
    the two specialized-columns modes of cs/mod.rs:237-246.  With the table id as a VARIABLE a sub-argument owns width + 1 variable
    columns and table_ids_column_idxes is empty (setup.rs:970-971; verifier.rs:675-678, 1402-1409)."""
    lookup = geometry["lookup"] if isinstance(geometry["lookup"], dict) else {}
    lk = lookup.get("UseSpecializedColumnsWithTableIdAsConstant")
    if lk:
        return lk, lk["width"], lk["num_repetitions"], lk["width"]
    lk = lookup.get("UseSpecializedColumnsWithTableIdAsVariable")
    if lk:
        assert not geometry["table_ids_column_idxes"], "table id as a variable: no table-id constant column"
        return lk, lk["width"], lk["num_repetitions"], lk["width"] + 1
    return None, 0, 0, 0


def quotient_identity(geometry, general_gates, specialized_gates, non_residues, challenges, values_at_z, value_z_omega,
                      verbose=False):
    """geometry: the fixture's `geometry` dict (vk.fixed_parameters); general_gates: evaluator names in gate_idx order;
    specialized_gates: [(name, num_repetitions)] after the lookup columns; challenges: dict beta, gamma, lookup_beta,
    lookup_gamma, alpha, z.  Returns (lhs, rhs) of the identity."""
    n = geometry["domain_size"]
    Vgp = geometry["num_variable_columns"]
    q = geometry["quotient_degree"]
    lk, w, reps, cps = lookup_parameters(geometry)
    n_spec_vars = sum(EVALUATORS[name][0] * r for name, r in specialized_gates)
    V = Vgp + cps * reps + n_spec_vars
    # constant columns: general-purpose gates' | table id | per-repetition constants of the gates over specialized columns
    # (share_constants = false, evaluator_data.rs:196-238; EVALUATORS[name][3] = constants one repetition reads inside evaluate_once)
    n_spec_consts = sum(EVALUATORS[name][3] * r for name, r in specialized_gates)
    nC = geometry["num_constant_columns"] + geometry["extra_constant_polys_for_selectors"] + len(geometry["table_ids_column_idxes"]) + n_spec_consts
    n_chunks = (V + q - 1) // q
    it = iter([tuple(v) for v in values_at_z])
    take = lambda k: [next(it) for _ in range(k)]
    var_z, con_z, sig_z = take(V), take(nC), take(V)
    z_at_z = next(it)
    part_z = take(n_chunks - 1)
    mult_z = take(1) if lk else []
    A_z, B_z = (take(reps), take(1)) if lk else ([], [])
    tab_z = take(w + 1) if lk else []
    qch_z = take(q)
    assert next(it, None) is None, "unexpected number of openings at z"
    paths = {}
    _paths(geometry["selectors_placement"], [], paths)
    consts_for_gp_gates = geometry["num_constant_columns"] + geometry["extra_constant_polys_for_selectors"]
    # number of alpha powers: lookup | specialized | general | L1 | copy-permutation chunks   (prover.rs:599-625)
    n_lookup_terms = reps + 1 if lk else 0
    n_spec_terms = sum(EVALUATORS[name][4] * r for name, r in specialized_gates)
    gp_reps = [EVALUATORS[name][1](Vgp, geometry["num_constant_columns"]) for name in general_gates]
    n_gp_terms = sum(EVALUATORS[name][4] * r for name, r in zip(general_gates, gp_reps))
    total = n_lookup_terms + n_spec_terms + n_gp_terms + 1 + n_chunks
    alpha = challenges["alpha"]
    alphas = [ONE]
    while len(alphas) < total:
        alphas.append(emul(alphas[-1], alpha))
    pos = 0
    T = ZERO
    beta, gamma, z = challenges["beta"], challenges["gamma"], challenges["z"]
    if lk:
        lb, lg = challenges["lookup_beta"], challenges["lookup_gamma"]
        gp = [ONE]
        for _ in range(w):
            gp.append(emul(gp[-1], lg))
        for i in range(reps):
            d = lb
            for j in range(cps):
                d = eadd(d, emul(gp[j], var_z[Vgp + i * cps + j]))
            if cps == w:                                   # table id in a constant column (verifier.rs:1447-1453)
                d = eadd(d, emul(gp[w], con_z[geometry["table_ids_column_idxes"][0]]))
            T = eadd(T, emul(esub(emul(A_z[i], d), ONE), alphas[pos]))
            pos += 1
        d = lb
        for j in range(w + 1):
            d = eadd(d, emul(gp[j], tab_z[j]))
        T = eadd(T, emul(esub(emul(B_z[0], d), mult_z[0]), alphas[pos]))
        pos += 1
    # gates over specialized columns: no selector, own variable columns after the lookup ones (evaluator_data.rs:190-236)
    off, coff = Vgp + cps * reps, nC - n_spec_consts
    for name, r in specialized_gates:
        width, _, n_shared, cstride, n_terms, fn = EVALUATORS[name]
        for k in range(r):
            for term in fn(var_z[off + k * width: off + (k + 1) * width], con_z[coff + k * cstride: coff + (k + 1) * cstride]):
                T = eadd(T, emul(term, alphas[pos]))
                pos += 1
        off += width * r
        coff += cstride * r
    # gates over general-purpose columns (verifier.rs:1640-1720)
    for idx, (name, r) in enumerate(zip(general_gates, gp_reps)):
        width, _, n_shared, cstride, n_terms, fn = EVALUATORS[name]
        if n_terms == 0:
            continue
        path = paths[idx]
        sel = ONE
        for b, bit in enumerate(path):
            sel = emul(sel, con_z[b] if bit else esub(ONE, con_z[b]))
        d = len(path)
        acc = ZERO
        for k in range(r):
            cons = con_z[d:d + n_shared] if n_shared else con_z[d + k * cstride: consts_for_gp_gates]
            for term in fn(var_z[k * width:(k + 1) * width], cons):
                acc = eadd(acc, emul(term, alphas[pos]))
                pos += 1
        T = eadd(T, emul(acc, sel))
    z_n = epow(z, n)
    vanishing = esub(z_n, ONE)
    l1 = emul(vanishing, einv(esub(z, ONE)))
    T = eadd(T, emul(emul(esub(z_at_z, ONE), l1), alphas[pos]))
    pos += 1
    lhs_list = part_z + [tuple(value_z_omega)]
    rhs_list = [z_at_z] + part_z
    for j in range(n_chunks):
        lhs, rhs = lhs_list[j], rhs_list[j]
        for c in range(j * q, min((j + 1) * q, V)):
            lhs = emul(lhs, eadd(eadd(emul(sig_z[c], beta), var_z[c]), gamma))
            rhs = emul(rhs, eadd(eadd(emul(escale(z, non_residues[c]), beta), var_z[c]), gamma))
        T = eadd(T, emul(esub(lhs, rhs), alphas[pos]))
        pos += 1
    assert pos == total
    t_chunks, pw = ZERO, ONE
    for el in qch_z:
        t_chunks = eadd(t_chunks, emul(el, pw))
        pw = emul(pw, z_n)
    if verbose:
        print("terms:", dict(lookup=n_lookup_terms, specialized=n_spec_terms, general=n_gp_terms, chunks=n_chunks))
    return T, emul(t_chunks, vanishing)
