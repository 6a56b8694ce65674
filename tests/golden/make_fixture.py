#!/usr/bin/env python3
"""Cut a small known-answer fixture out of the reference's own golden proof (proof.json / vk.json at the root of
matter-labs/era-boojum @ 2024_08_07; exercised by the reference test `test_recursive_verification`,
src/gadgets/recursion/recursive_verifier.rs:2213).  Pure data; nothing is recomputed here.

Kept: everything the Fiat–Shamir replay needs (all caps, public inputs, openings, final FRI monomials), the geometry
constants, and the first NUM_QUERIES query openings (leaf elements + Merkle paths for the 4 base oracles and the 6 FRI
oracles).  Output: tests/golden/boojum_fixture.json (~300 kB).

    python tests/golden/make_fixture.py [/root/reference]
"""
import json, os, sys
ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
NUM_QUERIES = 6
p = json.load(open(os.path.join(ref, 'proof.json')))
vk = json.load(open(os.path.join(ref, 'vk.json')))
fp = vk['fixed_parameters']
co = lambda e: e['coeffs']
out = {
    'source': 'matter-labs/era-boojum proof.json + vk.json (tag 2024_08_07)',
    'proof_config': p['proof_config'],
    'geometry': {
        'domain_size': fp['domain_size'],
        'num_variable_columns': fp['parameters']['num_columns_under_copy_permutation'],
        'num_witness_columns': fp['parameters']['num_witness_columns'],
        'num_constant_columns': fp['parameters']['num_constant_columns'],
        'extra_constant_polys_for_selectors': fp['extra_constant_polys_for_selectors'],
        'lookup': fp['lookup_parameters'],
        'quotient_degree': fp['quotient_degree'],
        'public_inputs_locations': fp['public_inputs_locations'],
        'table_ids_column_idxes': fp['table_ids_column_idxes'],
        'total_tables_len': fp['total_tables_len'],
        'selectors_placement': fp['selectors_placement'],
        'max_allowed_constraint_degree': fp['parameters']['max_allowed_constraint_degree'],
    },
    'setup_merkle_tree_cap': vk['setup_merkle_tree_cap'],
    'public_inputs': p['public_inputs'],
    'witness_oracle_cap': p['witness_oracle_cap'],
    'stage_2_oracle_cap': p['stage_2_oracle_cap'],
    'quotient_oracle_cap': p['quotient_oracle_cap'],
    'fri_base_oracle_cap': p['fri_base_oracle_cap'],
    'fri_intermediate_oracles_caps': p['fri_intermediate_oracles_caps'],
    'final_fri_monomials': p['final_fri_monomials'],
    'values_at_z': [co(e) for e in p['values_at_z']],
    'values_at_z_omega': [co(e) for e in p['values_at_z_omega']],
    'values_at_0': [co(e) for e in p['values_at_0']],
    'pow_challenge': p['pow_challenge'],
    'num_queries_total': len(p['queries_per_fri_repetition']),
    'queries': p['queries_per_fri_repetition'][:NUM_QUERIES],
}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'boojum_fixture.json')
json.dump(out, open(dst, 'w'), separators=(',', ':'))
print('wrote', dst, os.path.getsize(dst), 'bytes')
