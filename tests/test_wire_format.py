"""serde-JSON wire formats (era_boojum_amd/wire_format.py): the layout is the reference's own (`Proof`, proof.rs:121-136;
`VerificationKey`, verifier.rs:52-79) — checked against the golden fixture cut from proof.json / vk.json — and a proof
of the oracle prover survives the round trip."""
import json

import numpy as np

from era_boojum_amd import synthetic as S, wire_format as W
from oracle import prover as OP

PROOF_KEYS = ["proof_config", "public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap",
              "final_fri_monomials", "values_at_z", "values_at_z_omega", "values_at_0", "fri_base_oracle_cap",
              "fri_intermediate_oracles_caps", "queries_per_fri_repetition", "pow_challenge", "_marker"]


def test_golden_proof_pieces_round_trip(fixture_json):
    fx = fixture_json
    flat = {k: fx[k] for k in ("proof_config", "public_inputs", "witness_oracle_cap", "stage_2_oracle_cap", "quotient_oracle_cap",
                               "final_fri_monomials", "values_at_z", "values_at_z_omega", "values_at_0", "fri_base_oracle_cap",
                               "fri_intermediate_oracles_caps", "pow_challenge")}
    flat["queries_per_fri_repetition"] = fx["queries"]          # kept in the reference's layout by make_fixture.py
    obj = W.proof_to_reference_json(flat)
    assert list(obj.keys()) == PROOF_KEYS                        # serde field order of proof.json
    assert obj["queries_per_fri_repetition"] == fx["queries"]    # OracleQuery / SingleRoundQueries layout unchanged
    assert obj["values_at_z"][0] == {"coeffs": fx["values_at_z"][0], "_marker": None}
    assert obj["proof_config"] == fx["proof_config"]
    back = W.proof_from_reference_json(json.loads(W.dumps(obj)))
    for k in flat:
        assert back[k] == flat[k], k


def test_oracle_proof_round_trip_and_vk_layout():
    c = S.sha_shaped_circuit(8, seed=3, table_bits=2)
    setup = OP.Setup(c, 4, 8, threads=2)
    proof = OP.prove(c, setup, 4, 8, security_level=20, threads=2)
    obj = json.loads(W.dumps(W.proof_to_reference_json(proof)))
    back = W.proof_from_reference_json(obj)
    for k in ("public_inputs", "witness_oracle_cap", "values_at_z", "values_at_0", "final_fri_monomials",
              "fri_intermediate_oracles_caps", "queries_per_fri_repetition"):
        assert back[k] == json.loads(json.dumps(proof[k])), k
    vk = W.vk_to_reference_json(c, np.asarray(setup.cap), 4, 8)
    fp = vk["fixed_parameters"]
    assert list(fp.keys()) == ["parameters", "lookup_parameters", "domain_size", "total_tables_len", "public_inputs_locations",
                               "extra_constant_polys_for_selectors", "table_ids_column_idxes", "quotient_degree",
                               "selectors_placement", "fri_lde_factor", "cap_size"]           # vk.json order
    assert fp["parameters"]["num_constant_columns"] + fp["extra_constant_polys_for_selectors"] == fp["table_ids_column_idxes"][0]
    # every gate appears exactly once in the placement tree, at the depth of its selector path
    seen = {}

    def walk(node, depth):
        if "GateOnly" in node:
            seen[node["GateOnly"]["gate_idx"]] = depth
        else:
            walk(node["Fork"]["left"], depth + 1)
            walk(node["Fork"]["right"], depth + 1)

    walk(fp["selectors_placement"], 0)
    assert seen == {i: len(g.path) for i, g in enumerate(c.gates)}
