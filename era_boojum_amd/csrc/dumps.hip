// Seam S1 without linking: the prover's inputs as the reference's own `MemcopySerializable` dumps.  A Rust host
// writes `SetupBaseStorage`, `WitnessVec` and `DenseVariablesCopyHint` with `write_into_buffer` and hands the bytes over; what
// is code on the Rust side (geometry, gate list in gate_idx order, public input locations, proof config) comes in a bj_circuit.
//
// Byte layouts restated from the reference (all integers little-endian; era_boojum_amd/memcopy_format.py holds the Python
// reader / writer the tests cross-check against):
//   Vec<F> / Polynomial          u64 length in field elements, then the raw u64 words
//                                (src/cs/implementations/fast_serialization.rs:139-207, polynomial/mod.rs:95-118)
//   Vec<Arc<Polynomial>>         u64 count, then each polynomial                         (fast_serialization.rs:17-47)
//   SetupBaseStorage             copy_permutation_polys, constant_columns, lookup_tables_columns as above, then bincode of
//                                `table_ids_column_idxes: Vec<usize>` and of `selectors_placement: TreeNode`
//                                (polynomial_storage.rs:77-126); bincode default: u64 lengths, u32 variant tags, usize as u64,
//                                bool as one byte; TreeNode::{Empty, GateOnly(GateDescription), Fork{left, right}},
//                                GateDescription {gate_idx, num_constants, degree, needs_selector, is_lookup}: setup.rs:1378-1396
//   WitnessVec                   public_inputs_locations: u64 count + (u64 column, u64 row) pairs; all_values: Vec<F>;
//                                multiplicities: u64 count + u32 each                    (witness.rs:29-71)
//   DenseVariablesCopyHint       u64 columns, each: u64 length + u64 per cell; bit 63 set = no variable in the cell, the low 48
//                                bits index all_values (hints/mod.rs:10-61, 104-118; src/cs/mod.rs:44-46, 151-181)
// The reference holds no golden bytes for these formats (parity unpinned by vectors): the layouts follow the code cited.
#include "ctx.h"

#include <cstring>
#include <vector>

using gl::u64;

namespace {

struct Reader {
    const unsigned char *p, *end;
    bool ok = true;
    Reader(const void *d, size_t n) : p((const unsigned char *)d), end((const unsigned char *)d + n) {}
    bool take(void *dst, size_t n) {
        if (!ok || (size_t)(end - p) < n) return ok = false;
        if (dst) memcpy(dst, p, n);
        p += n;
        return true;
    }
    uint64_t u64v() {
        uint64_t v = 0;
        take(&v, 8);
        return v;
    }
    uint32_t u32v() {
        uint32_t v = 0;
        take(&v, 4);
        return v;
    }
    unsigned char u8v() {
        unsigned char v = 0;
        take(&v, 1);
        return v;
    }
    const unsigned char *skip(size_t n) {
        const unsigned char *q = p;
        return take(nullptr, n) ? q : nullptr;
    }
};

struct PolyVec {
    std::vector<const unsigned char *> cols;   // raw words of each polynomial inside the dump
    size_t n = 0;                              // their common length
};
// Vec<Arc<Polynomial>>: every polynomial must have the same power-of-two length (0 columns: n stays 0)
bool read_poly_vec(Reader &r, PolyVec *out) {
    const uint64_t count = r.u64v();
    if (!r.ok || count > (1u << 20)) return false;
    for (uint64_t i = 0; i < count; i++) {
        const uint64_t len = r.u64v();
        if (!r.ok || len > ((uint64_t)1 << 32)) return false;
        if (i == 0) out->n = (size_t)len;
        if (len != out->n) return false;
        const unsigned char *w = r.skip((size_t)len * 8);
        if (!w) return false;
        out->cols.push_back(w);
    }
    return true;
}

// TreeNode in bincode: paths of the gates (left = multiply by the constant column, right = by 1 - constant: the convention
// the reference's own proof pins, compute_selector_subpath prover.rs:2775-2916) and the maximum over the leaves of depth + degree
// (TreeNode::compute_stats, setup.rs:1398-1453)
struct TreeWalk {
    std::vector<std::vector<unsigned char>> path;   // per gate_idx
    std::vector<char> seen;
    uint64_t max_degree = 0;
    std::string err;
};
bool read_tree(Reader &r, std::vector<unsigned char> &prefix, TreeWalk *w, int depth) {
    if (depth > 16) return (w->err = "selector tree deeper than 16"), false;
    const uint32_t tag = r.u32v();
    if (!r.ok) return (w->err = "truncated TreeNode"), false;
    if (tag == 0) return true;
    if (tag == 1) {
        const uint64_t idx = r.u64v();
        const uint64_t num_constants = r.u64v();
        const uint64_t degree = r.u64v();
        r.u8v();    // needs_selector
        const unsigned char is_lookup = r.u8v();
        if (!r.ok) return (w->err = "truncated GateDescription"), false;
        // untrusted sizes: a gate of degree 2^16 or with 2^16 constants does not exist (the quotient degree derived from them
        // below is at most 16 here); without the bound a crafted degree wrapped the sum / never ended the doubling loop
        if (degree > (1u << 16) || num_constants > (1u << 16))
            return (w->err = "GateDescription with a degree or constant count above 2^16"), false;
        if (is_lookup) return (w->err = "a lookup placed through the selector tree (general-purpose-column lookups) is not supported"), false;
        if (idx >= w->path.size() || w->seen[idx]) return (w->err = "selector tree names a gate index the circuit does not have (or twice)"), false;
        w->seen[idx] = 1;
        w->path[idx] = prefix;
        if (prefix.size() + degree > w->max_degree) w->max_degree = prefix.size() + degree;
        return true;
    }
    if (tag == 2) {
        prefix.push_back(1);
        if (!read_tree(r, prefix, w, depth + 1)) return false;
        prefix.back() = 0;
        if (!read_tree(r, prefix, w, depth + 1)) return false;
        prefix.pop_back();
        return true;
    }
    return (w->err = "bad TreeNode variant"), false;
}

// make_non_residues (utils.rs:636-688): 1, then the smallest quadratic non-residues whose n-th powers differ from 1 and from
// each other — the cosets k_c * <omega> of the copy permutation must be disjoint
u64 pow_mod(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl::mul(r, b);
        b = gl::mul(b, b);
        e >>= 1;
    }
    return r;
}
std::vector<u64> make_non_residues(size_t count, u64 domain_size) {
    std::vector<u64> out{1}, seen;
    u64 cur = 1;
    while (out.size() < count) {
        cur++;
        if (pow_mod(cur, (gl::P - 1) / 2) != gl::P - 1) continue;
        const u64 t = pow_mod(cur, domain_size);
        bool dup = t == 1;
        for (u64 s : seen) dup |= s == t;
        if (dup) continue;
        seen.push_back(t);
        out.push_back(cur);
    }
    return out;
}

constexpr u64 PLACEHOLDER_BIT = 1ull << 63, LOW_U48 = (1ull << 48) - 1;
// witness_set_from_witness_vec (witness.rs:386-443): cell = value of its variable, 0 for a placeholder
__global__ void materialize_cells_kernel(const u64 *hint, const u64 *values, size_t n_values, u64 *cells, size_t count, unsigned *bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const u64 h = hint[i];
    u64 v = 0;
    if (!(h & PLACEHOLDER_BIT)) {
        const u64 idx = h & LOW_U48;
        if (idx < n_values)
            v = values[idx];
        else
            atomicOr(bad, 1u);
    }
    cells[i] = v;
}

struct DeviceBlock {
    void *p = nullptr;
    ~DeviceBlock() {
        if (p) (void)hipFree(p);
    }
};
}  // namespace

extern "C" {

// SetupBaseStorage bytes -> views into the dump (no copies); num_gates: how many gate indices the selector tree may name
struct ParsedSetup {
    PolyVec sig, con, tab;
    std::vector<uint64_t> ids;
    TreeWalk walk;
};
static bool parse_setup_dump(const void *setup_base, size_t setup_base_len, size_t num_gates, ParsedSetup *out, std::string *err) {
    Reader r(setup_base, setup_base_len);
    if (!read_poly_vec(r, &out->sig) || !read_poly_vec(r, &out->con) || !read_poly_vec(r, &out->tab))
        return (*err = "SetupBaseStorage dump: truncated or ragged polynomial vectors"), false;
    const uint64_t n_ids = r.u64v();
    for (uint64_t i = 0; r.ok && i < n_ids && i < 64; i++) out->ids.push_back(r.u64v());
    if (!r.ok || n_ids > 64) return (*err = "SetupBaseStorage dump: bad table_ids_column_idxes"), false;
    out->walk.path.resize(num_gates);
    out->walk.seen.assign(num_gates, 0);
    std::vector<unsigned char> prefix;
    if (!read_tree(r, prefix, &out->walk, 0)) return (*err = "SetupBaseStorage dump: " + out->walk.err), false;
    if (r.p != r.end) return (*err = "SetupBaseStorage dump: " + std::to_string((size_t)(r.end - r.p)) + " trailing bytes"), false;
    return true;
}

// quotient degree from the selector tree: the power of two covering max (depth + degree) - 1 (setup.rs:560-600).  64-bit with a
// cap: `max_degree` comes from the dump (read_tree bounds every term, this loop must end whatever it is handed)
static bool quotient_degree_from_tree(uint64_t max_degree, unsigned *q_out) {
    uint64_t q = 1;
    while (q + 1 < max_degree && q <= 64) q *= 2;
    if (q > 64) return false;
    *q_out = (unsigned)q;
    return true;
}

extern "C" int bj_setup_dump_info(const void *setup_base, size_t setup_base_len, uint64_t *info8) {
    if (!setup_base || !info8) return BJ_ERR_INVALID_ARG;
    ParsedSetup P;
    std::string err;
    if (!parse_setup_dump(setup_base, setup_base_len, 4096, &P, &err)) return BJ_ERR_INVALID_ARG;
    unsigned q = 0;      // the derived quotient degree must exist: the same host-side steps bj_setup_create_from_dump runs
    if (!quotient_degree_from_tree(P.walk.max_degree, &q)) return BJ_ERR_INVALID_ARG;
    size_t gates = 0, longest = 0;
    for (size_t g = 0; g < P.walk.seen.size(); g++)
        if (P.walk.seen[g]) {
            gates = g + 1;
            if (P.walk.path[g].size() > longest) longest = P.walk.path[g].size();
        }
    const uint64_t v[8] = {P.sig.n, P.sig.cols.size(), P.con.cols.size(), P.tab.cols.size(), P.ids.size(), P.ids.empty() ? 0 : P.ids[0],
                           gates, P.walk.max_degree | ((uint64_t)longest << 32)};
    memcpy(info8, v, sizeof v);
    return BJ_OK;
}

int bj_setup_create_from_dump(bj_ctx *ctx, const bj_circuit *circuit, const void *setup_base, size_t setup_base_len,
                              const bj_proof_config *config, bj_setup **out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!circuit || !setup_base || !config || !out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_setup_create_from_dump: null argument");
    ParsedSetup P;
    {
        std::string err;
        if (!parse_setup_dump(setup_base, setup_base_len, circuit->num_gates, &P, &err)) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "%s", err.c_str());
    }
    PolyVec &sig = P.sig, &con = P.con, &tab = P.tab;
    std::vector<uint64_t> &ids = P.ids;
    TreeWalk &walk = P.walk;
    const size_t n = sig.n;
    if (!bj::is_pow2(n) || (size_t)1 << circuit->log_n != n || (!con.cols.empty() && con.n != n) || (!tab.cols.empty() && tab.n != n))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "SetupBaseStorage dump: columns of %zu rows, the circuit says 2^%u", n, circuit->log_n);
    if (sig.cols.size() != circuit->num_vars)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "SetupBaseStorage dump: %zu copy-permutation polynomials, the circuit has %u variable columns",
                        sig.cols.size(), circuit->num_vars);
    const bool lookups = circuit->lookup_reps != 0;
    // one table-id column: UseSpecializedColumnsWithTableIdAsConstant { share_table_id }; none: ..AsVariable (setup.rs:970-990)
    if (lookups && (ids.size() > 1 || tab.cols.size() != circuit->lookup_width + 1))
        return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "SetupBaseStorage dump: specialized lookups need at most one (shared) table-id "
                        "column and width + 1 table columns (%zu ids, %zu table columns)", ids.size(), tab.cols.size());
    // what the dump knows better than the caller: constant columns, the table-id column, selector paths, the quotient degree
    bj_circuit c = *circuit;
    c.num_constant_cols = (unsigned)con.cols.size();
    if (lookups) c.table_id_col = ids.empty() ? BJ_TABLE_ID_AS_VARIABLE : (unsigned)ids[0];
    std::vector<bj_gate_desc> gates(circuit->gates, circuit->gates + circuit->num_gates);
    for (unsigned g = 0; g < circuit->num_gates; g++) {
        if (!walk.seen[g] && gates[g].kind != BJ_GATE_NOP && gates[g].num_terms)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "SetupBaseStorage dump: gate %u is not placed in the selector tree", g);
        const auto &p = walk.path[g];
        if (p.size() > 8) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "selector path of %zu bits", p.size());
        gates[g].path_len = (unsigned)p.size();
        memset(gates[g].path, 0, sizeof(gates[g].path));
        for (size_t b = 0; b < p.size(); b++) gates[g].path[b] = p[b];
    }
    c.gates = gates.data();
    if (c.quotient_degree == 0) {   // quotient degree: the power of two covering max (depth + degree) - 1 (setup.rs:560-600)
        if (!quotient_degree_from_tree(walk.max_degree, &c.quotient_degree))
            return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "SetupBaseStorage dump: selector depth + gate degree %llu asks for a quotient "
                                                     "degree above 64", (unsigned long long)walk.max_degree);
    }
    std::vector<u64> nr;
    if (!c.non_residues) {
        nr = make_non_residues(c.num_vars, n);
        c.non_residues = nr.data();
    }
    // the setup call takes contiguous [cols][n] arrays: gather the dump's polynomials (each has its own length prefix)
    std::vector<u64> hs(sig.cols.size() * n), hc(con.cols.size() * n), ht(tab.cols.size() * n);
    for (size_t i = 0; i < sig.cols.size(); i++) memcpy(&hs[i * n], sig.cols[i], n * 8);
    for (size_t i = 0; i < con.cols.size(); i++) memcpy(&hc[i * n], con.cols[i], n * 8);
    for (size_t i = 0; i < tab.cols.size(); i++) memcpy(&ht[i * n], tab.cols[i], n * 8);
    return bj_setup_create(ctx, &c, hs.data(), hc.data(), lookups ? ht.data() : nullptr, config, out);
}

int bj_prove_from_dumps(bj_ctx *ctx, const bj_setup *setup, const void *witness_vec, size_t witness_vec_len,
                        const void *variables_hint, size_t variables_hint_len, const void *witness_hint, size_t witness_hint_len,
                        bj_proof **out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!setup || !witness_vec || !variables_hint || !out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_from_dumps: null argument");
    unsigned log_n = 0, num_vars = 0, num_witness_cols = 0, num_public = 0;
    if (int rc = bj_setup_shape(setup, &log_n, &num_vars, &num_witness_cols, &num_public)) return rc;
    if ((num_witness_cols != 0) != (witness_hint != nullptr))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_from_dumps: the setup has %u non-copiable witness columns: a DenseWitnessCopyHint "
                        "dump is %s", num_witness_cols, num_witness_cols ? "required" : "not expected");
    const size_t n = (size_t)1 << log_n;
    Reader w(witness_vec, witness_vec_len);
    const uint64_t n_pub = w.u64v();
    if (!w.ok || n_pub != num_public)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "WitnessVec dump: %llu public input locations, the setup was made with %u",
                        (unsigned long long)n_pub, num_public);
    std::vector<uint64_t> pub_col(n_pub), pub_row(n_pub);
    for (uint64_t i = 0; i < n_pub; i++) {
        pub_col[i] = w.u64v();
        pub_row[i] = w.u64v();
    }
    const uint64_t n_values = w.u64v();      // lengths come from the dump: bound them by what is left of it before multiplying
    const unsigned char *values = (w.ok && n_values <= (size_t)(w.end - w.p) / 8) ? w.skip((size_t)n_values * 8) : nullptr;
    if (!values) w.ok = false;
    const uint64_t n_mult = w.u64v();
    const unsigned char *mult = (w.ok && n_mult <= (size_t)(w.end - w.p) / 4) ? w.skip((size_t)n_mult * 4) : nullptr;
    if (!mult) w.ok = false;
    if (!w.ok || !values || (n_mult && !mult) || w.p != w.end)
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "WitnessVec dump: truncated or trailing bytes");
    if (n_mult > n) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "WitnessVec dump: %llu multiplicities for %zu rows", (unsigned long long)n_mult, n);
    // DenseVariablesCopyHint, then DenseWitnessCopyHint (same layout, same indexing of all_values: witness.rs:445-490): the witness
    // columns travel right behind the variable columns, as bj_prove_dev takes them
    const unsigned total_cols = num_vars + num_witness_cols;
    std::vector<const unsigned char *> hint_col(total_cols);
    for (int which = 0; which < (num_witness_cols ? 2 : 1); which++) {
        Reader h(which ? witness_hint : variables_hint, which ? witness_hint_len : variables_hint_len);
        const char *what = which ? "DenseWitnessCopyHint" : "DenseVariablesCopyHint";
        const unsigned want = which ? num_witness_cols : num_vars, first = which ? num_vars : 0;
        const uint64_t hint_cols = h.u64v();
        if (!h.ok || hint_cols != want)
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "%s dump: %llu columns, the circuit has %u", what, (unsigned long long)hint_cols, want);
        for (unsigned c = 0; c < want; c++) {
            const uint64_t len = h.u64v();
            if (!h.ok || len != n) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "%s dump: column %u has %llu cells, not %zu", what, c, (unsigned long long)len, n);
            hint_col[first + c] = h.skip(n * 8);
            if (!hint_col[first + c]) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "%s dump: truncated", what);
        }
        if (h.p != h.end) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "%s dump: trailing bytes", what);
    }
    // cells on the device: all_values once, the hint in groups of columns, one gather per group
    DeviceBlock d_values, d_hint, d_cells, d_mult, d_bad;
    const unsigned G = 8;
    BJ_HIP(ctx, hipMalloc(&d_values.p, (n_values ? n_values : 1) * 8));
    BJ_HIP(ctx, hipMalloc(&d_hint.p, (size_t)G * n * 8));
    BJ_HIP(ctx, hipMalloc(&d_cells.p, (size_t)total_cols * n * 8));
    BJ_HIP(ctx, hipMalloc(&d_mult.p, n * 8));
    BJ_HIP(ctx, hipMalloc(&d_bad.p, 4));
    BJ_HIP(ctx, hipMemsetAsync(d_bad.p, 0, 4, ctx->stream));
    if (n_values) BJ_HIP(ctx, hipMemcpyAsync(d_values.p, values, n_values * 8, hipMemcpyHostToDevice, ctx->stream));
    for (unsigned c0 = 0; c0 < total_cols; c0 += G) {
        const unsigned g = total_cols - c0 < G ? total_cols - c0 : G;
        for (unsigned k = 0; k < g; k++)
            BJ_HIP(ctx, hipMemcpyAsync((u64 *)d_hint.p + (size_t)k * n, hint_col[c0 + k], n * 8, hipMemcpyHostToDevice, ctx->stream));
        const size_t count = (size_t)g * n;
        hipLaunchKernelGGL(materialize_cells_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, (const u64 *)d_hint.p,
                           (const u64 *)d_values.p, (size_t)n_values, (u64 *)d_cells.p + (size_t)c0 * n, count, (unsigned *)d_bad.p);
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the pageable source of the next group's copies is reused storage only for us
    }
    BJ_CHECK_LAUNCH(ctx);
    unsigned bad = 0;
    BJ_HIP(ctx, hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "copy hint dump: a cell names a variable beyond all_values");
    // materialize_multiplicities_polynomials (witness.rs:225-272): the per-table counters, concatenated, zero-extended
    std::vector<u64> hm(n, 0);
    for (uint64_t i = 0; i < n_mult; i++) {
        uint32_t v;
        memcpy(&v, mult + 4 * i, 4);
        hm[i] = v;
    }
    BJ_HIP(ctx, hipMemcpy(d_mult.p, hm.data(), n * 8, hipMemcpyHostToDevice));
    // public input values in location order: the cells themselves (witness.rs:400-410)
    std::vector<u64> pub(n_pub ? n_pub : 1, 0);
    for (uint64_t i = 0; i < n_pub; i++) {
        if (pub_col[i] >= num_vars || pub_row[i] >= n) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "WitnessVec dump: public input %llu outside the trace", (unsigned long long)i);
        BJ_HIP(ctx, hipMemcpy(&pub[i], (u64 *)d_cells.p + pub_col[i] * n + pub_row[i], 8, hipMemcpyDeviceToHost));
    }
    return bj_prove_dev(ctx, setup, (const u64 *)d_cells.p, (const u64 *)d_mult.p, pub.data(), out);
}

}  // extern "C"
