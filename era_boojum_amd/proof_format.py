"""Parser of the flat u64 proof serialisation produced by bj_proof_serialize (csrc/prover.hip) into a dict with the field
names of the reference's `Proof` (src/cs/implementations/proof.rs:121-136).

Layout (all u64, little endian):
  header[19] = magic 'BJPF', version (2), n_public, cap_size, n_values_at_z, n_values_at_z_omega, n_values_at_0, n_fri_oracles,
               final_degree, n_queries, witness leaf width, stage-2 leaf width, quotient leaf width, setup leaf width,
               base-oracle path depth, log_n, fri_lde_factor, pow_bits, pow_challenge      (version 1: the first 17 only)
  schedule[n_fri_oracles] | public inputs | witness cap | stage-2 cap | quotient cap (cap_size*4 each)
  values_at_z (2 each) | values_at_z_omega | values_at_0 | FRI caps (n_fri_oracles * cap_size*4)
  final monomials c0[final_degree], c1[final_degree]
  per query: index | for each of {witness, stage-2, quotient, setup}: leaf elements, path (depth*4)
             | for each FRI oracle i: 2*2^schedule[i] leaf elements, path
"""
import numpy as np

MAGIC = 0x424A5046


def parse(buf, security_level=None, pow_bits=0):
    a = np.asarray(buf, dtype=np.uint64)
    pos = 0

    def take(k):
        nonlocal pos
        out = a[pos:pos + k]
        if out.size != k:
            raise ValueError("truncated proof buffer")
        pos += k
        return out

    h = [int(x) for x in take(17)]
    if h[0] != MAGIC or h[1] not in (1, 2):
        raise ValueError("not a BJPF v1/v2 proof")
    (_, version, n_pub, cap, nz, nzo, n0, n_fri, final_degree, n_queries, w_wit, w_s2, w_q, w_su, depth, log_n, fri_lde) = h
    pow_challenge = 0
    if version == 2:
        pow_bits, pow_challenge = (int(x) for x in take(2))
    sched = [int(x) for x in take(n_fri)]
    caps4 = lambda: take(cap * 4).reshape(cap, 4).tolist()
    pairs = lambda k: take(2 * k).reshape(k, 2).tolist()
    proof = {"proof_config": {"fri_lde_factor": fri_lde, "merkle_tree_cap_size": cap, "fri_folding_schedule": None,
                              "security_level": security_level, "pow_bits": pow_bits}}
    proof["public_inputs"] = take(n_pub).tolist()
    proof["witness_oracle_cap"] = caps4()
    proof["stage_2_oracle_cap"] = caps4()
    proof["quotient_oracle_cap"] = caps4()
    proof["values_at_z"] = pairs(nz)
    proof["values_at_z_omega"] = pairs(nzo)
    proof["values_at_0"] = pairs(n0)
    fri_caps = [caps4() for _ in range(n_fri)]
    proof["fri_base_oracle_cap"] = fri_caps[0]
    proof["fri_intermediate_oracles_caps"] = fri_caps[1:]
    proof["final_fri_monomials"] = [take(final_degree).tolist(), take(final_degree).tolist()]
    queries, indices = [], []
    n_leaves = (1 << log_n) * fri_lde
    for _ in range(n_queries):
        indices.append(int(take(1)[0]))
        qd = {}
        for name, w in (("witness_query", w_wit), ("stage_2_query", w_s2), ("quotient_query", w_q), ("setup_query", w_su)):
            qd[name] = {"leaf_elements": take(w).tolist(), "proof": take(depth * 4).reshape(depth, 4).tolist()}
        qd["fri_queries"] = []
        ln = n_leaves
        for k in sched:
            d = ((ln >> k) // cap).bit_length() - 1
            qd["fri_queries"].append({"leaf_elements": take(2 << k).tolist(), "proof": take(d * 4).reshape(d, 4).tolist()})
            ln >>= k
        queries.append(qd)
    if pos != a.size:
        raise ValueError("trailing data in proof buffer")
    proof["queries_per_fri_repetition"] = queries
    proof["pow_challenge"] = pow_challenge
    proof["_query_indices"] = indices
    proof["_schedule"] = sched
    return proof
