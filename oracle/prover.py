"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of `prove_cpu_basic` (src/cs/implementations/prover.rs:153-2266) for the circuit class of the SHA-256
bench (general-purpose gates ConstantsAllocator / FMA / Reduction<4> / Nop, specialized lookups with the table id as a
shared constant, Poseidon2 tree hasher + Poseidon2 transcript, no PoW), orchestrated in Python over the C bulk operations
of liboracle.so.  Round structure, Fiat–Shamir order, leaf layouts, opening order and query construction follow the
reference line by line (cited inline); the transcript/Merkle/FRI/DEEP building blocks are the ones pinned by the golden
proof.  The proof is returned as a dict with the same field names as the reference's `Proof` (proof.rs:121-136).
"""
import ctypes as C

import numpy as np

import oracle as O

P = O.P
_p, _arr, lib = O._p, O._arr, O.lib


# ---------------- F_p^2 scalars as python int pairs ----------------
def emul(a, b): return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def eadd(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def esub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def escale(a, s): return (a[0] * s % P, a[1] * s % P)
def einv(a):
    ni = O.inv((a[0] * a[0] - 7 * a[1] * a[1]) % P)
    return (a[0] * ni % P, (-a[1] * ni) % P)
def epow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = emul(r, a)
        a = emul(a, a)
        e >>= 1
    return r


def gates_flat(gates):
    out = []
    for g in gates:
        path = [1 if b else 0 for b in g.path] + [0] * (6 - len(g.path))
        out += [g.kind, len(g.path), g.reps, g.var_stride, g.const_stride, g.num_terms] + path
    return np.array(out, dtype=np.int32)


def copy_perm_stage2(variables, sigmas, non_res, log_n, chunk, beta, gamma, threads=1):
    V, n = variables.shape
    n_chunks = (V + chunk - 1) // chunk
    z = np.zeros((2, n), dtype=np.uint64)
    partials = np.zeros((max(n_chunks - 1, 1), 2, n), dtype=np.uint64)
    lib().orc_copy_perm_stage2(_p(_arr(variables)), _p(_arr(sigmas)), _p(_arr(non_res)), C.c_size_t(V), C.c_uint(log_n),
                               C.c_size_t(chunk), _p(_arr(beta)), _p(_arr(gamma)), _p(z), _p(partials), C.c_int(threads))
    return z, partials[:n_chunks - 1]


def lookup_cols_per_sub(c):
    """specialized_columns_per_subargument (cs/mod.rs:300-312): width, + 1 when the table id is a variable column."""
    return c.lookup_width + (1 if getattr(c, "table_id_as_variable", False) else 0)


def lookup_table_id(c, constants):
    """The table-id constant column, or None in the UseSpecializedColumnsWithTableIdAsVariable mode."""
    return None if getattr(c, "table_id_as_variable", False) else constants[c.table_id_col]


def lookup_polys(lookup_vars, table_id, tables, mult, reps, w, log_n, beta, gamma, threads=1):
    """table_id None: UseSpecializedColumnsWithTableIdAsVariable — lookup_vars holds w + 1 columns per sub-argument."""
    n = 1 << log_n
    A = np.zeros((reps, 2, n), dtype=np.uint64)
    B = np.zeros((2, n), dtype=np.uint64)
    lib().orc_lookup_polys(_p(_arr(lookup_vars)), _p(_arr(table_id)) if table_id is not None else None, _p(_arr(tables)), _p(_arr(mult)), C.c_size_t(reps),
                           C.c_size_t(w), C.c_uint(log_n), _p(_arr(beta)), _p(_arr(gamma)), _p(A), _p(B), C.c_int(threads))
    return A, B


def quotient(vars_q, consts_q, sigmas_q, z_q, partials_q, A_q, B_q, mult_q, tables_q, circuit, log_q, alphas, beta, gamma,
             lbeta, lgamma, threads=1, coset_begin=0, coset_count=0):
    """All inputs are [cols][q*n] restrictions of the LDEs to the first q cosets; returns T = numerator / (x^n - 1).  With
    coset_count: only the cosets [coset_begin, coset_begin + coset_count) of those q, inputs [cols][coset_count*n]."""
    V, Q = vars_q.shape
    gf = gates_flat(circuit.gates)
    out = np.zeros((2, Q), dtype=np.uint64)
    npart = partials_q.shape[0] // 2 if partials_q.size else 0
    dummy = np.zeros(2, dtype=np.uint64)
    lib().orc_quotient(_p(vars_q), C.c_size_t(V), _p(consts_q), C.c_size_t(consts_q.shape[0]), _p(sigmas_q), _p(z_q),
                       _p(partials_q if partials_q.size else dummy), C.c_size_t(npart),
                       _p(A_q if A_q.size else dummy), _p(B_q if B_q.size else dummy), _p(mult_q if mult_q.size else dummy),
                       _p(tables_q if tables_q.size else dummy), C.c_size_t(circuit.lookup_reps),
                       C.c_size_t(circuit.lookup_width), C.c_size_t(circuit.num_gp_vars),
                       C.c_size_t(-1 if getattr(circuit, "table_id_as_variable", False) else circuit.table_id_col),
                       gf.ctypes.data_as(C.POINTER(C.c_int)), C.c_size_t(len(circuit.gates)),
                       _p(_arr(circuit.non_residues)), C.c_size_t(circuit.quotient_degree), C.c_uint(circuit.log_n),
                       C.c_uint(log_q), C.c_uint(coset_begin), _p(_arr(alphas).reshape(-1)), C.c_size_t(len(alphas)),
                       _p(_arr(beta)), _p(_arr(gamma)), _p(_arr(lbeta)), _p(_arr(lgamma)), _p(out), C.c_int(threads),
                       C.c_size_t(coset_count))
    return out


def _F():
    from era_boojum_amd import field_np      # numpy Goldilocks vector arithmetic (host-side helper, not the HIP path)
    return field_np


def _op_list_gate_terms(c, spec_gates, vars_q, con_q, a_gates, a_spec, wits_q=None):
    """sum over the op-list gates of selector * sum alpha * term on the quotient domain, from the programs' own semantics
    (GateProgram.evaluate_columns: plain numpy field arithmetic).  General-purpose gates of kind >= 5 with their selector
    path; gates over specialized columns without one, on their own columns after the lookup ones."""
    F = _F()
    Qn = vars_q.shape[1]
    acc0, acc1 = np.zeros(Qn, dtype=np.uint64), np.zeros(Qn, dtype=np.uint64)

    def program_of(g):
        if getattr(g, "program", None) is not None:
            return g.program
        from era_boojum_amd.gate_program import poseidon2_flattened_compact_program
        assert g.name == "Poseidon2FlattenedGate"
        return poseidon2_flattened_compact_program()      # the same 118 terms in 2.4 k column operations instead of 9.6 k

    def weighted(prog, vcols, ccols, alphas, aoff, wcols=()):
        s0, s1 = np.zeros(Qn, dtype=np.uint64), np.zeros(Qn, dtype=np.uint64)
        for term in prog.evaluate_columns(vcols, ccols, wcols):
            a = alphas[aoff]
            s0 = F.add(s0, F.mul(term, np.uint64(a[0])))
            s1 = F.add(s1, F.mul(term, np.uint64(a[1])))
            aoff += 1
        return s0, s1, aoff
    aoff = 0
    for g in c.gates:
        if g.kind < 5:
            aoff += g.reps * g.num_terms
            continue
        prog, d = program_of(g), len(g.path)
        sel = np.ones(Qn, dtype=np.uint64)
        for b, bit in enumerate(g.path):
            sel = F.mul(sel, con_q[b] if bit else F.sub(np.ones(Qn, dtype=np.uint64), con_q[b]))
        for r in range(g.reps):
            vcols = [vars_q[r * g.var_stride + k] for k in range(g.principal_width)]
            ccols = [con_q[k] for k in range(d + r * g.const_stride, con_q.shape[0])]
            ws = getattr(g, "wit_stride", 0)
            wcols = [wits_q[k] for k in range(r * ws, wits_q.shape[0])] if (ws and wits_q is not None) else ()
            s0, s1, aoff = weighted(prog, vcols, ccols, a_gates, aoff, wcols)
            acc0, acc1 = F.add(acc0, F.mul(s0, sel)), F.add(acc1, F.mul(s1, sel))
    col, aoff = c.num_gp_vars + c.lookup_reps * lookup_cols_per_sub(c), 0
    # constants of the gates over specialized columns: the LAST constant columns (behind the general-purpose gates' ones and the
    # table-id column), reps * const_stride per gate — every repetition its own (prover.rs:748-772, evaluator_data.rs:196-238)
    ccol = c.num_constant_cols - sum(g.reps * g.const_stride for g in spec_gates)
    for g in spec_gates:
        for r in range(g.reps):
            vcols = [vars_q[col + r * g.var_stride + k] for k in range(g.var_stride)]
            ccols = [con_q[ccol + r * g.const_stride + k] for k in range(g.const_stride)]
            s0, s1, aoff = weighted(g.program, vcols, ccols, a_spec, aoff)
            acc0, acc1 = F.add(acc0, s0), F.add(acc1, s1)
        col += g.reps * g.var_stride
        ccol += g.reps * g.const_stride
    return acc0, acc1


def pow_seed(t):
    """256 / CHAR_BITS = 4 challenges, +1 because 4 is not a multiple of CHAR_BITS (prover.rs:2114-2119): 40 seed bytes."""
    return b"".join(int(t.challenge()).to_bytes(8, "little") for _ in range(5))


def _pow_first_words(seed, nonces, runner):
    """First 8 digest bytes (little-endian u64) of H(seed || le64(nonce)) for every nonce; runner 1 = Blake2s256 (pow.rs:50-133,
    hashlib), 2 = Keccak256 (pow.rs:139-230: the original Keccak padding, oracle/keccak.py — pinned through hashlib.sha3_256)."""
    if runner == 2:
        from oracle import keccak as K
        words = np.empty((len(nonces), 6), dtype=np.uint64)
        words[:, :5] = np.frombuffer(seed, dtype="<u8")
        words[:, 5] = np.asarray(nonces, dtype=np.uint64)
        return [int(x) for x in K.hash_words(words)[:, 0]]
    import hashlib
    return [int.from_bytes(hashlib.blake2s(seed + int(nonce).to_bytes(8, "little")).digest()[:8], "little") for nonce in nonces]


def pow_ok(seed, pow_bits, nonce, runner=1):
    first = _pow_first_words(seed, [nonce], runner)[0]
    tz = 64 if first == 0 else (first & -first).bit_length() - 1
    return tz >= pow_bits


def pow_search(seed, pow_bits, runner=1):
    """The serial search of pow.rs:60-73 / 149-162: the smallest valid nonce (in batches: the first hit of the first batch with one)."""
    base, batch = 0, 1 << 12
    while True:
        for i, first in enumerate(_pow_first_words(seed, range(base, base + batch), runner)):
            if (64 if first == 0 else (first & -first).bit_length() - 1) >= pow_bits:
                return base + i
        base += batch


def hashing_layer(hasher):
    """The module providing merkle_* / Transcript / QueryIndexer / do_fri for a tree hasher."""
    if hasher == 2:
        from oracle import blake
        return blake
    if hasher == 3:
        from oracle import keccak
        return keccak.layer()
    return O


class Setup:
    """get_full_setup's prover-side outputs (setup.rs:1273-1300): setup LDEs, the setup tree and the VK cap."""

    def __init__(self, circuit, fri_lde_factor, cap_size, threads=1, hasher=1):
        H = hashing_layer(hasher)            # 1: Poseidon2 sponge, 2: Blake2s-256 (oracle/blake.py)
        c = self.circuit = circuit
        self.fri_lde = fri_lde_factor
        self.cap_size = cap_size
        self.L = max(fri_lde_factor, c.quotient_degree)
        self.log_L = self.L.bit_length() - 1
        n = c.n
        # leaf order of the setup oracle: sigma || constants || tables   (polynomial_storage.rs:667-676)
        parts = [c.sigmas, c.constants] + ([c.tables] if c.lookup_reps else [])
        self.cols_nat = np.concatenate(parts, axis=0)
        self.mono = O.ifft_batch(self.cols_nat, 1, threads)
        self.lde = O.lde_batch(self.mono, self.log_L, threads)            # [cols][L][n]
        N = n * fri_lde_factor
        self.leaves_view = np.ascontiguousarray(self.lde[:, :fri_lde_factor, :].reshape(-1, N))
        self.tree = H.merkle_construct(self.leaves_view, cap_size, threads)
        self.cap = H.merkle_cap(self.tree, N, cap_size)


def prove(circuit, setup, fri_lde_factor=8, cap_size=16, security_level=100, pow_bits=0, threads=1, return_aux=False,
          transcript_kind=1, pow_runner=1):
    c = circuit
    n, log_n = c.n, c.log_n
    V = c.num_vars
    q = c.quotient_degree
    log_q = q.bit_length() - 1
    L, log_L = setup.L, setup.log_L                        # used_lde_degree (prover.rs:313)
    log_fri = fri_lde_factor.bit_length() - 1
    N = n * fri_lde_factor
    Q = n * q
    has_lookup = c.lookup_reps > 0
    # transcript 1: Poseidon2 (golden proof), 2: Poseidon v1 (SHA-256 bench script), 3: Blake2s (non-recursive config, with the
    # Blake2s tree hasher: Transcript::CompatibleCap = TreeHasher::Output)
    H = hashing_layer({3: 2, 4: 3}.get(transcript_kind, 1))
    t = H.Transcript(transcript_kind)
    t.absorb_cap(setup.cap)                                # prover.rs:211
    pub_vals = [v for (_, _, v) in c.public_inputs]
    t.absorb(pub_vals)                                     # prover.rs:257-259

    def commit(cols_nat):
        mono = O.ifft_batch(cols_nat, 1, threads)
        lde = O.lde_batch(mono, log_L, threads)
        view = np.ascontiguousarray(lde[:, :fri_lde_factor, :].reshape(-1, N))
        tree = H.merkle_construct(view, cap_size, threads)
        return lde, view, tree, H.merkle_cap(tree, N, cap_size)

    # ---- round 1: witness (prover.rs:270-353); leaf = variables || witness || multiplicities
    Wc = int(getattr(c, "num_witness_cols", 0))                          # non-copiable witness columns sit behind the variables
    parts = [c.variables] + ([c.witness] if Wc else []) + ([c.multiplicities] if has_lookup else [])
    wit_nat = np.concatenate(parts, axis=0) if len(parts) > 1 else c.variables
    wit_lde, wit_view, wit_tree, wit_cap = commit(wit_nat)
    t.absorb_cap(wit_cap)
    # ---- round 2: copy-permutation + lookup (prover.rs:360-554)
    beta, gamma = t.challenge_ext(), t.challenge_ext()
    z_nat, partials_nat = copy_perm_stage2(c.variables, c.sigmas, c.non_residues, log_n, q, beta, gamma, threads)
    n_partials = partials_nat.shape[0]
    stage2 = [z_nat[0], z_nat[1]] + [partials_nat[j][k] for j in range(n_partials) for k in range(2)]
    lbeta = lgamma = (0, 0)
    if has_lookup:
        lbeta, lgamma = t.challenge_ext(), t.challenge_ext()
        A_nat, B_nat = lookup_polys(c.variables[c.num_gp_vars:], lookup_table_id(c, c.constants), c.tables, c.multiplicities[0],
                                    c.lookup_reps, c.lookup_width, log_n, lbeta, lgamma, threads)
        stage2 += [A_nat[i][k] for i in range(c.lookup_reps) for k in range(2)] + [B_nat[0], B_nat[1]]
    s2_lde, s2_view, s2_tree, s2_cap = commit(np.stack(stage2))
    t.absorb_cap(s2_cap)
    # ---- round 3: quotient (prover.rs:560-1495)
    alpha = t.challenge_ext()
    n_lookup_terms = c.lookup_reps + 1 if has_lookup else 0
    spec_gates = list(getattr(c, "specialized_gates", None) or [])
    n_spec_terms = sum(g.reps * g.num_terms for g in spec_gates)
    n_gate_terms = sum(g.reps * g.num_terms for g in c.gates)
    n_chunks = (V + q - 1) // q
    total_terms = n_lookup_terms + n_spec_terms + n_gate_terms + 1 + n_chunks   # lookup | specialized | general | L1 | chunks (prover.rs:599-625)
    alphas_all = [(1, 0)]
    while len(alphas_all) < total_terms:
        alphas_all.append(emul(alphas_all[-1], alpha))                  # materialize_powers_serial (utils.rs:31)
    a_spec = alphas_all[n_lookup_terms:n_lookup_terms + n_spec_terms]
    a_gates = alphas_all[n_lookup_terms + n_spec_terms:n_lookup_terms + n_spec_terms + n_gate_terms]
    alphas = alphas_all[:n_lookup_terms] + alphas_all[n_lookup_terms + n_spec_terms:]     # what the C quotient consumes
    sub = lambda lde: np.ascontiguousarray(lde[:, :q, :].reshape(lde.shape[0], Q))
    vars_q = sub(wit_lde[:V])
    mult_q = sub(wit_lde[V + Wc:V + Wc + 1]).reshape(-1) if has_lookup else np.zeros(0, dtype=np.uint64)
    wits_q = sub(wit_lde[V:V + Wc]) if Wc else None
    nS, nC = V, c.num_constant_cols
    sig_q, con_q = sub(setup.lde[:nS]), sub(setup.lde[nS:nS + nC])
    tab_q = sub(setup.lde[nS + nC:]) if has_lookup else np.zeros(0, dtype=np.uint64)
    s2_q = sub(s2_lde)
    z_q, part_q = s2_q[0:2], s2_q[2:2 + 2 * n_partials]
    A_q = s2_q[2 + 2 * n_partials:2 + 2 * n_partials + 2 * c.lookup_reps] if has_lookup else np.zeros(0, dtype=np.uint64)
    B_q = s2_q[2 + 2 * n_partials + 2 * c.lookup_reps:] if has_lookup else np.zeros(0, dtype=np.uint64)
    T = quotient(vars_q, con_q, sig_q, np.ascontiguousarray(z_q), np.ascontiguousarray(part_q), np.ascontiguousarray(A_q),
                 np.ascontiguousarray(B_q), mult_q, tab_q, c, log_q, alphas, beta, gamma, lbeta, lgamma, threads)
    if spec_gates or any(g.kind >= 5 for g in c.gates):
        e0, e1 = _op_list_gate_terms(c, spec_gates, vars_q, con_q, a_gates, a_spec, wits_q)
        for cs in range(q):   # divide by x^n - 1, constant on a coset: x^n = (7 * w_{nL}^{bitrev_L(cs)})^n
            xn = pow(7 * pow(O.omega(log_n + log_L), O.bitrev(cs, log_L), P) % P, n, P)
            inv = np.uint64(O.inv((xn - 1) % P))
            sl = slice(cs * n, (cs + 1) * n)
            T[0][sl] = _F().add(T[0][sl], _F().mul(e0[sl], inv))
            T[1][sl] = _F().add(T[1][sl], _F().mul(e1[sl], inv))
    # flatten_presumably_bitreversed == bit-reversal of the size-qn array; iNTT on coset g (prover.rs:1405-1422)
    qmono = O.ifft_batch(np.stack([O.bitreverse(T[0]), O.bitreverse(T[1])]), 7, threads)
    assert qmono[0][-1] == 0 and qmono[1][-1] == 0, "unsatisfied (prover.rs:1425-1438)"
    chunks = []
    for j in range(q):                                                  # chunk_into_subpolys_of_degree (polynomial/mod.rs:267)
        chunks += [qmono[0][j * n:(j + 1) * n], qmono[1][j * n:(j + 1) * n]]
    q_lde = O.lde_batch(np.stack(chunks), log_fri, threads)             # LDE only to fri_lde_factor (prover.rs:1473-1480)
    q_view = np.ascontiguousarray(q_lde.reshape(-1, N))
    q_tree = H.merkle_construct(q_view, cap_size, threads)
    q_cap = H.merkle_cap(q_tree, N, cap_size)
    t.absorb_cap(q_cap)
    # ---- round 4: openings (prover.rs:1501-1802)
    z = t.challenge_ext()
    w0, w1 = O.barycentric_weights(log_n, 7, z)
    ev_base = lambda lde_col: O.barycentric_eval_base(lde_col[0], w0, w1)
    ev_ext = lambda a, b: O.barycentric_eval_ext(a[0], b[0], w0, w1)
    values_at_z = [ev_base(wit_lde[i]) for i in range(V + Wc)]                              # variables, then witness columns
    values_at_z += [ev_base(setup.lde[nS + i]) for i in range(nC)]                          # constants
    values_at_z += [ev_base(setup.lde[i]) for i in range(nS)]                               # sigmas
    values_at_z.append(ev_ext(s2_lde[0], s2_lde[1]))                                        # z
    values_at_z += [ev_ext(s2_lde[2 + 2 * j], s2_lde[3 + 2 * j]) for j in range(n_partials)]
    if has_lookup:
        values_at_z.append(ev_base(wit_lde[V + Wc]))                                        # multiplicities
        o = 2 + 2 * n_partials
        values_at_z += [ev_ext(s2_lde[o + 2 * i], s2_lde[o + 2 * i + 1]) for i in range(c.lookup_reps + 1)]   # A_i, B
        values_at_z += [ev_base(setup.lde[nS + nC + i]) for i in range(c.lookup_width + 1)]  # tables
    values_at_z += [ev_ext(q_lde[2 * j], q_lde[2 * j + 1]) for j in range(q)]               # quotient chunks
    for v in values_at_z:
        t.absorb(v)
    z_omega = escale(z, O.omega(log_n))
    wz0, wz1 = O.barycentric_weights(log_n, 7, z_omega)
    values_at_z_omega = [O.barycentric_eval_ext(s2_lde[0][0], s2_lde[1][0], wz0, wz1)]
    for v in values_at_z_omega:
        t.absorb(v)
    values_at_0 = []
    if has_lookup:
        v0w0, v0w1 = O.barycentric_weights(log_n, 7, (0, 0))
        o = 2 + 2 * n_partials
        values_at_0 = [O.barycentric_eval_ext(s2_lde[o + 2 * i][0], s2_lde[o + 2 * i + 1][0], v0w0, v0w1)
                       for i in range(c.lookup_reps + 1)]
    for v in values_at_0:
        t.absorb(v)
    # ---- round 5a: DEEP (prover.rs:1803-2067)
    pub_tuples = []                                                     # grouped by opening point, first-seen order
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
    total_ch = len(values_at_z) + 1 + len(values_at_0) + sum(len(s) for _, s in pub_tuples)
    chs = [(1, 0), cch]                                                 # materialize_ext_challenge_powers (prover.rs:2374)
    while len(chs) < total_ch:
        chs.append(emul(chs[-1], cch))
    fri_sub = lambda lde_col: np.ascontiguousarray(lde_col[:fri_lde_factor].reshape(-1))
    b_ = lambda col: (fri_sub(col), None)
    e_ = lambda a, b: (fri_sub(a), fri_sub(b))
    src = [b_(wit_lde[i]) for i in range(V + Wc)]
    src += [b_(setup.lde[nS + i]) for i in range(nC)]
    src += [b_(setup.lde[i]) for i in range(nS)]
    src.append(e_(s2_lde[0], s2_lde[1]))
    src += [e_(s2_lde[2 + 2 * j], s2_lde[3 + 2 * j]) for j in range(n_partials)]
    if has_lookup:
        src.append(b_(wit_lde[V + Wc]))
        o = 2 + 2 * n_partials
        src += [e_(s2_lde[o + 2 * i], s2_lde[o + 2 * i + 1]) for i in range(c.lookup_reps + 1)]
        src += [b_(setup.lde[nS + nC + i]) for i in range(c.lookup_width + 1)]
    src += [e_(q_lde[2 * j], q_lde[2 * j + 1]) for j in range(q)]
    assert len(src) == len(values_at_z)
    d0, d1 = np.zeros(N, dtype=np.uint64), np.zeros(N, dtype=np.uint64)
    off = 0
    O.deep_quotient_accumulate(src, values_at_z, chs[off:off + len(src)], z, log_n, log_fri, d0, d1, threads)
    off += len(src)
    O.deep_quotient_accumulate([e_(s2_lde[0], s2_lde[1])], values_at_z_omega, chs[off:off + 1], z_omega, log_n, log_fri, d0, d1, threads)
    off += 1
    if has_lookup:
        o = 2 + 2 * n_partials
        s0 = [e_(s2_lde[o + 2 * i], s2_lde[o + 2 * i + 1]) for i in range(c.lookup_reps + 1)]
        O.deep_quotient_accumulate(s0, values_at_0, chs[off:off + len(s0)], (0, 0), log_n, log_fri, d0, d1, threads)
        off += len(s0)
    for at, items in pub_tuples:
        sp = [b_(wit_lde[col]) for col, _ in items]
        vp = [(val, 0) for _, val in items]
        O.deep_quotient_accumulate(sp, vp, chs[off:off + len(sp)], (at, 0), log_n, log_fri, d0, d1, threads)
        off += len(sp)
    assert off == len(chs)
    # ---- round 5b: FRI (prover.rs:2075-2105)
    new_pow, num_queries, sched, final_degree = O.fri_schedule(security_level, cap_size, pow_bits, log_fri, log_n)
    fri = H.do_fri(d0, d1, log_fri, sched, cap_size, t, threads)
    # ---- proof of work (prover.rs:2107-2131; the POW type parameter: 1 = Blake2s256, pow.rs:50-133, hashlib is the hash;
    # 2 = Keccak256, pow.rs:139-230)
    pow_challenge = 0
    if new_pow:
        pow_challenge = pow_search(pow_seed(t), new_pow, pow_runner)
        t.absorb([pow_challenge & 0xFFFFFFFF, pow_challenge >> 32])
    # ---- round 6: queries (prover.rs:2161-2266)
    qi = H.QueryIndexer(log_n, log_fri)
    queries = []

    def open_base(view, tree, idx):
        lh, path = H.merkle_proof(tree, N, cap_size, idx)
        return {"leaf_elements": [int(x) for x in view[:, idx]], "proof": [[int(x) for x in p] for p in path]}

    for _ in range(num_queries):
        idx = qi.next(t)
        qd = {"witness_query": open_base(wit_view, wit_tree, idx), "stage_2_query": open_base(s2_view, s2_tree, idx),
              "quotient_query": open_base(q_view, q_tree, idx), "setup_query": open_base(setup.leaves_view, setup.tree, idx),
              "fri_queries": []}
        f_idx, ln = idx, N
        srcs = fri["sources"]
        for i, k in enumerate(sched):
            E = 1 << k
            j = f_idx >> k
            s0, s1 = srcs[i]
            leaf = np.concatenate([s0[j * E:(j + 1) * E], s1[j * E:(j + 1) * E]])
            lh, path = H.merkle_proof(fri["trees"][i], ln >> k, cap_size, j)
            qd["fri_queries"].append({"leaf_elements": [int(x) for x in leaf], "proof": [[int(x) for x in p] for p in path]})
            f_idx >>= k
            ln >>= k
        queries.append(qd)
    fd = fri["final_degree"]
    proof = {
        "proof_config": {"fri_lde_factor": fri_lde_factor, "merkle_tree_cap_size": cap_size, "fri_folding_schedule": None,
                         "security_level": security_level, "pow_bits": pow_bits},
        "public_inputs": pub_vals,
        "witness_oracle_cap": wit_cap.tolist(), "stage_2_oracle_cap": s2_cap.tolist(), "quotient_oracle_cap": q_cap.tolist(),
        "final_fri_monomials": [fri["final_monomials"][0][:fd].tolist(), fri["final_monomials"][1][:fd].tolist()],
        "values_at_z": [list(v) for v in values_at_z], "values_at_z_omega": [list(v) for v in values_at_z_omega],
        "values_at_0": [list(v) for v in values_at_0],
        "fri_base_oracle_cap": fri["caps"][0].tolist(),
        "fri_intermediate_oracles_caps": [cap.tolist() for cap in fri["caps"][1:]],
        "queries_per_fri_repetition": queries, "pow_challenge": pow_challenge,
    }
    if return_aux:
        aux = dict(beta=beta, gamma=gamma, lbeta=lbeta, lgamma=lgamma, alpha=alpha, z=z, deep_challenge=cch,
                   z_nat=z_nat, partials_nat=partials_nat, stage2_nat=np.stack(stage2), T=T, qmono=qmono,
                   wit_lde=wit_lde, s2_lde=s2_lde, q_lde=q_lde, deep=(d0, d1), schedule=sched, alphas=alphas,
                   fri_challenges=fri["challenges"])
        return proof, aux
    return proof
