"""serde-JSON wire formats of the reference's `Proof` and `VerificationKey`, so a Rust host (or the reference's own
verifier) can consume what this prover emits without linking anything.

Layouts follow the derives in the reference (field order = declaration order):
  Proof                      src/cs/implementations/proof.rs:121-136   (golden instance: proof.json)
  OracleQuery                src/cs/implementations/proof.rs:12-17
  SingleRoundQueries         src/cs/implementations/proof.rs:38-45
  ExtensionField             src/field/traits/field.rs (coeffs + PhantomData `_marker`)
  VerificationKey            src/cs/implementations/verifier.rs:52-79, 176-181  (golden instance: vk.json)
  TreeNode / GateDescription src/cs/implementations/setup.rs:1346-1396
All field elements are canonical u64 (JSON numbers, as serde emits them).
"""
import json


def _ext(pair):
    return {"coeffs": [int(pair[0]), int(pair[1])], "_marker": None}


def _query(q):
    return {"leaf_elements": [int(x) for x in q["leaf_elements"]], "proof": [[int(x) for x in d] for d in q["proof"]]}


def proof_to_reference_json(proof):
    """`proof`: the dict of proof_format.parse (or of the oracle prover).  Returns the serde layout of `Proof`."""
    cfg = proof["proof_config"]
    return {
        "proof_config": {"fri_lde_factor": int(cfg["fri_lde_factor"]), "merkle_tree_cap_size": int(cfg["merkle_tree_cap_size"]),
                         "fri_folding_schedule": cfg.get("fri_folding_schedule"), "security_level": cfg["security_level"],
                         "pow_bits": int(cfg.get("pow_bits", 0))},
        "public_inputs": [int(x) for x in proof["public_inputs"]],
        "witness_oracle_cap": [[int(x) for x in d] for d in proof["witness_oracle_cap"]],
        "stage_2_oracle_cap": [[int(x) for x in d] for d in proof["stage_2_oracle_cap"]],
        "quotient_oracle_cap": [[int(x) for x in d] for d in proof["quotient_oracle_cap"]],
        "final_fri_monomials": [[int(x) for x in proof["final_fri_monomials"][0]], [int(x) for x in proof["final_fri_monomials"][1]]],
        "values_at_z": [_ext(e) for e in proof["values_at_z"]],
        "values_at_z_omega": [_ext(e) for e in proof["values_at_z_omega"]],
        "values_at_0": [_ext(e) for e in proof["values_at_0"]],
        "fri_base_oracle_cap": [[int(x) for x in d] for d in proof["fri_base_oracle_cap"]],
        "fri_intermediate_oracles_caps": [[[int(x) for x in d] for d in c] for c in proof["fri_intermediate_oracles_caps"]],
        "queries_per_fri_repetition": [
            {"witness_query": _query(q["witness_query"]), "stage_2_query": _query(q["stage_2_query"]),
             "quotient_query": _query(q["quotient_query"]), "setup_query": _query(q["setup_query"]),
             "fri_queries": [_query(f) for f in q["fri_queries"]]}
            for q in proof["queries_per_fri_repetition"]],
        "pow_challenge": int(proof.get("pow_challenge", 0)),
        "_marker": None,
    }


def proof_from_reference_json(obj):
    """Inverse of proof_to_reference_json: serde `Proof` JSON object -> the flat dict the parity tests compare."""
    co = lambda e: [int(e["coeffs"][0]), int(e["coeffs"][1])]
    out = {k: obj[k] for k in ("proof_config", "public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap",
                               "final_fri_monomials", "fri_base_oracle_cap", "fri_intermediate_oracles_caps", "pow_challenge")}
    for k in ("values_at_z", "values_at_z_omega", "values_at_0"):
        out[k] = [co(e) for e in obj[k]]
    out["queries_per_fri_repetition"] = [
        {"witness_query": _query(q["witness_query"]), "stage_2_query": _query(q["stage_2_query"]),
         "quotient_query": _query(q["quotient_query"]), "setup_query": _query(q["setup_query"]),
         "fri_queries": [_query(f) for f in q["fri_queries"]]} for q in obj["queries_per_fri_repetition"]]
    return out


def _tree(node, gate_index):
    if node is None:
        return "Empty"
    if node[0] == "gate":
        g = node[1]
        return {"GateOnly": {"gate_idx": gate_index[id(g)], "num_constants": g.num_constants, "degree": g.degree,
                             "needs_selector": bool(g.needs_selector), "is_lookup": False}}
    return {"Fork": {"left": _tree(node[1], gate_index), "right": _tree(node[2], gate_index)}}


def vk_to_reference_json(circuit, setup_cap, fri_lde_factor, cap_size):
    """serde layout of `VerificationKey` for a circuit of era_boojum_amd.synthetic.Circuit shape."""
    c = circuit
    tid_var = bool(getattr(c, "table_id_as_variable", False))
    if c.lookup_reps and tid_var:     # cs/mod.rs:237-241; share_table_id is not read by the prover in this mode (lookup_argument_in_ext.rs:357)
        lookup = {"UseSpecializedColumnsWithTableIdAsVariable": {"width": c.lookup_width, "num_repetitions": c.lookup_reps,
                                                                 "share_table_id": False}}
    elif c.lookup_reps:
        lookup = {"UseSpecializedColumnsWithTableIdAsConstant": {"width": c.lookup_width, "num_repetitions": c.lookup_reps,
                                                                 "share_table_id": True}}
    else:
        lookup = "NoLookup"
    gate_index = {id(g): i for i, g in enumerate(c.gates)}
    return {
        "fixed_parameters": {
            "parameters": {"num_columns_under_copy_permutation": c.num_gp_vars, "num_witness_columns": int(getattr(c, "num_witness_cols", 0)),
                           "num_constant_columns": c.geometry_constant_cols,
                           "max_allowed_constraint_degree": c.max_allowed_constraint_degree},
            "lookup_parameters": lookup,
            "domain_size": c.n,
            "total_tables_len": c.total_tables_len if c.lookup_reps else 0,
            "public_inputs_locations": [[int(col), int(row)] for col, row, _ in c.public_inputs],
            "extra_constant_polys_for_selectors": c.num_constants_for_gates - c.geometry_constant_cols,
            "table_ids_column_idxes": [c.table_id_col] if (c.lookup_reps and not tid_var) else [],
            "quotient_degree": c.quotient_degree,
            "selectors_placement": _tree(c.selector_tree, gate_index),
            "fri_lde_factor": fri_lde_factor,
            "cap_size": cap_size,
        },
        "setup_merkle_tree_cap": [[int(x) for x in d] for d in setup_cap],
    }


def dumps(obj):
    return json.dumps(obj, separators=(",", ":"))
