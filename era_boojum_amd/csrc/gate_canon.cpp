// Canonical form of gate op lists: see gate_canon.h.  Pure host C++17.
#include "gate_canon.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <queue>
#include <unordered_map>

namespace bj {
namespace canon {
namespace {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
inline uint64_t f_canon(uint64_t a) { return a >= P ? a - P : a; }
inline uint64_t f_add(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }
inline uint64_t f_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (P - b); }
inline uint64_t f_mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
inline uint64_t f_inv(uint64_t a) {   // a^(p-2); 0 -> 0 like the device's inv_pow
    uint64_t r = 1, b = a, e = P - 2;
    while (e) {
        if (e & 1) r = f_mul(r, b);
        b = f_mul(b, b);
        e >>= 1;
    }
    return r;
}

struct H128 {
    uint64_t a, b;
    bool operator==(const H128 &o) const { return a == o.a && b == o.b; }
    bool operator<(const H128 &o) const { return a != o.a ? a < o.a : b < o.b; }
};
inline uint64_t fmix(uint64_t x) {   // the 64-bit finaliser of MurmurHash3
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDULL; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ULL; x ^= x >> 33;
    return x;
}
inline uint64_t smix(uint64_t x) {   // SplitMix64's output function
    x += 0x9E3779B97F4A7C15ULL; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL; x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
inline H128 absorb(H128 h, uint64_t w) {
    h.a = fmix(h.a ^ w) + 0x165667B19E3779F9ULL * (h.b | 1);
    h.b = smix(h.b + w) ^ (h.a >> 29);
    return h;
}
inline H128 absorb(H128 h, H128 w) { return absorb(absorb(h, w.a), w.b); }

// an operand while the DAG is built: a column, a constant (the canonical residue itself), or a DAG node
enum : uint32_t { K_VAR = BJ_IDX_VARIABLE_POLY, K_WIT = BJ_IDX_WITNESS_POLY, K_CON = BJ_IDX_CONSTANT_POLY, K_NODE = BJ_IDX_TEMPORARY,
                  K_VAL = BJ_IDX_CONSTANT_VALUE, K_NONE = 7 };
struct Op {
    uint32_t kind = K_NONE;
    uint64_t payload = 0;
    bool operator==(const Op &o) const { return kind == o.kind && payload == o.payload; }
};
struct RawNode {
    uint32_t op;
    Op a, b;
    H128 h;
};
struct Key {
    uint32_t op;
    Op a, b;
    bool operator==(const Key &o) const { return op == o.op && a == o.a && b == o.b; }
};
struct KeyHash {
    size_t operator()(const Key &k) const {
        uint64_t h = fmix(k.op * 0x9E3779B97F4A7C15ULL ^ k.a.kind);
        h = fmix(h ^ k.a.payload) + k.b.kind;
        return (size_t)fmix(h ^ (k.b.payload * 0xC2B2AE3D27D4EB4FULL));
    }
};

struct Builder {
    std::vector<RawNode> nodes;
    std::unordered_map<Key, uint32_t, KeyHash> cons;

    H128 hash_of(const Op &x) const {
        if (x.kind == K_NODE) return nodes[x.payload].h;
        return absorb(H128{0x243F6A8885A308D3ULL + x.kind, 0x13198A2E03707344ULL ^ ((uint64_t)x.kind << 56)}, x.payload);
    }
    static Op val(uint64_t v) { return Op{K_VAL, v}; }
    static bool is_val(const Op &x, uint64_t v) { return x.kind == K_VAL && x.payload == v; }

    Op node(uint32_t op, Op a, Op b) {
        const bool comm = op == BJ_OP_ADD || op == BJ_OP_MUL;
        H128 ha = hash_of(a), hb = b.kind == K_NONE ? H128{0, 0} : hash_of(b);
        if (comm && hb < ha) {
            std::swap(a, b);
            std::swap(ha, hb);
        }
        const Key k{op, a, b};
        auto it = cons.find(k);
        if (it != cons.end()) return Op{K_NODE, it->second};
        H128 h = absorb(H128{0xA4093822299F31D0ULL, 0x082EFA98EC4E6C89ULL}, (uint64_t)op);
        h = absorb(h, ha);
        if (b.kind != K_NONE) h = absorb(h, hb);
        nodes.push_back(RawNode{op, a, b, h});
        const uint32_t id = (uint32_t)nodes.size() - 1;
        cons.emplace(k, id);
        return Op{K_NODE, id};
    }
    // one recorded relation with the rewrites of gate_canon.h
    Op relation(uint32_t op, Op a, Op b) {
        const bool va = a.kind == K_VAL, vb = b.kind == K_VAL;
        switch (op) {
            case BJ_OP_ADD:
                if (va && vb) return val(f_add(a.payload, b.payload));
                if (is_val(a, 0)) return b;
                if (is_val(b, 0)) return a;
                if (a == b) return node(BJ_OP_DOUBLE, a, Op{});
                return node(BJ_OP_ADD, a, b);
            case BJ_OP_DOUBLE:
                if (va) return val(f_add(a.payload, a.payload));
                return node(BJ_OP_DOUBLE, a, Op{});
            case BJ_OP_SUB:
                if (va && vb) return val(f_sub(a.payload, b.payload));
                if (is_val(b, 0)) return a;
                if (a == b) return val(0);
                if (is_val(a, 0)) return node(BJ_OP_NEGATE, b, Op{});
                return node(BJ_OP_SUB, a, b);
            case BJ_OP_NEGATE:
                if (va) return val(f_sub(0, a.payload));
                return node(BJ_OP_NEGATE, a, Op{});
            case BJ_OP_MUL:
                if (va && vb) return val(f_mul(a.payload, b.payload));
                if (is_val(a, 0) || is_val(b, 0)) return val(0);
                if (is_val(a, 1)) return b;
                if (is_val(b, 1)) return a;
                if (a == b) return node(BJ_OP_SQUARE, a, Op{});
                return node(BJ_OP_MUL, a, b);
            case BJ_OP_SQUARE:
                if (va) return val(f_mul(a.payload, a.payload));
                return node(BJ_OP_SQUARE, a, Op{});
            default:
                if (va) return val(f_inv(a.payload));
                return node(BJ_OP_INVERSE, a, Op{});
        }
    }
};

int bad(std::string *err, const char *fmt, unsigned x = 0, unsigned y = 0) {
    if (err) {
        char buf[160];
        snprintf(buf, sizeof buf, fmt, x, y);
        *err = buf;
    }
    return BJ_ERR_INVALID_ARG;
}

}  // namespace

int canonicalize(const bj_gate_program *p, Program *out, std::string *err) {
    if (!p || !p->writes || p->num_writes == 0 || (p->num_relations && !p->relations))
        return bad(err, "gate program: null / empty program");
    if (p->num_relations > (1u << 24) || p->num_writes > (1u << 20))
        return bad(err, "gate program: %u relations / %u terms: too large", p->num_relations, p->num_writes);
    Builder B;
    B.nodes.reserve(p->num_relations);
    // the value each temporary holds right now; a map, not an array: the reference's process-wide counter makes the numbers of a
    // capture start anywhere (num_temporaries may be millions for a list of ten relations)
    std::unordered_map<uint32_t, Op> cur;
    cur.reserve(p->num_relations);
    auto operand = [&](const bj_gate_index &ix, Op *o) -> bool {
        switch (ix.kind) {
            case BJ_IDX_VARIABLE_POLY:
            case BJ_IDX_WITNESS_POLY:
            case BJ_IDX_CONSTANT_POLY:
                if (ix.index >= (1u << 20)) return false;
                *o = Op{ix.kind, ix.index};
                return true;
            case BJ_IDX_TEMPORARY: {
                if (ix.index >= p->num_temporaries) return false;
                const auto it = cur.find(ix.index);
                if (it == cur.end()) return false;       // read before any relation wrote it
                *o = it->second;
                return true;
            }
            case BJ_IDX_CONSTANT_VALUE:
                if (!p->values || ix.index >= p->num_values) return false;
                *o = Builder::val(f_canon(p->values[ix.index]));
                return true;
            default: return false;
        }
    };
    for (uint32_t i = 0; i < p->num_relations; i++) {
        const bj_gate_relation &R = p->relations[i];
        if (R.op < BJ_OP_ADD || R.op > BJ_OP_INVERSE || R.dst >= p->num_temporaries) return bad(err, "gate program: bad relation %u", i);
        const bool binary = R.op == BJ_OP_ADD || R.op == BJ_OP_SUB || R.op == BJ_OP_MUL;
        Op a, b;
        if (!operand(R.a, &a)) return bad(err, "gate program: bad first operand in relation %u", i);
        if (binary && !operand(R.b, &b)) return bad(err, "gate program: bad second operand in relation %u", i);
        cur[R.dst] = B.relation(R.op, a, b);
    }
    std::vector<Op> terms(p->num_writes);
    for (uint32_t t = 0; t < p->num_writes; t++)
        if (!operand(p->writes[t], &terms[t])) return bad(err, "gate program: bad write %u", t);

    // fingerprint: the ordered terms' structural hashes
    H128 fp = absorb(H128{0x452821E638D01377ULL, 0xBE5466CF34E90C6CULL}, (uint64_t)p->num_writes);
    for (const Op &t : terms) fp = absorb(fp, B.hash_of(t));

    // schedule: iterative depth-first post-order from the terms, a term is written as soon as its value exists
    Program &O = *out;
    O = Program{};
    O.num_terms = p->num_writes;
    O.fp[0] = fp.a;
    O.fp[1] = fp.b;
    std::vector<uint32_t> sched_of(B.nodes.size(), UINT32_MAX);   // DAG node -> position in O.nodes
    std::unordered_map<uint64_t, uint32_t> value_index;
    auto see_extent = [&](const Op &x) {
        if (x.kind == K_VAR) O.var_extent = std::max(O.var_extent, (uint32_t)x.payload + 1);
        if (x.kind == K_CON) O.const_extent = std::max(O.const_extent, (uint32_t)x.payload + 1);
        if (x.kind == K_WIT) O.wit_extent = std::max(O.wit_extent, (uint32_t)x.payload + 1);
    };
    auto final_operand = [&](const Op &x) -> Operand {
        see_extent(x);
        if (x.kind == K_NODE) return Operand{BJ_IDX_TEMPORARY, sched_of[x.payload]};
        if (x.kind == K_VAL) {
            auto it = value_index.find(x.payload);
            if (it == value_index.end()) {
                it = value_index.emplace(x.payload, (uint32_t)O.values.size()).first;
                O.values.push_back(x.payload);
            }
            return Operand{BJ_IDX_CONSTANT_VALUE, it->second};
        }
        if (x.kind == K_NONE) return Operand{0, 0};
        return Operand{x.kind, (uint32_t)x.payload};
    };
    // Sethi-Ullman numbers: how many values a subexpression keeps alive while it is computed.  The operand that needs more goes
    // first, so a long accumulation (contribution = ((c0 + c1 s) + c2 s') + ...) is walked along its spine with one value
    // waiting instead of one per level.  Nodes are created after their operands, so one forward sweep does it.
    std::vector<uint32_t> need(B.nodes.size(), 1);
    for (size_t i = 0; i < B.nodes.size(); i++) {
        const RawNode &N = B.nodes[i];
        const uint32_t la = N.a.kind == K_NODE ? need[N.a.payload] : 0, lb = N.b.kind == K_NODE ? need[N.b.payload] : 0;
        need[i] = std::max<uint32_t>(1, la == lb ? la + (lb ? 1 : 0) : std::max(la, lb));
    }
    struct Frame {
        uint32_t node;
        int stage;
    };
    std::vector<Frame> stack;
    for (uint32_t t = 0; t < p->num_writes; t++) {
        if (terms[t].kind == K_NODE && sched_of[terms[t].payload] == UINT32_MAX) stack.push_back(Frame{(uint32_t)terms[t].payload, 0});
        while (!stack.empty()) {
            Frame &f = stack.back();
            const RawNode &N = B.nodes[f.node];
            if (sched_of[f.node] != UINT32_MAX) {   // reached again through another path while it waited on the stack
                stack.pop_back();
                continue;
            }
            const bool b_first = N.a.kind == K_NODE && N.b.kind == K_NODE && need[N.b.payload] > need[N.a.payload];
            const Op &first = b_first ? N.b : N.a, &second = b_first ? N.a : N.b;
            if (f.stage == 0) {
                f.stage = 1;
                if (first.kind == K_NODE && sched_of[first.payload] == UINT32_MAX) {
                    stack.push_back(Frame{(uint32_t)first.payload, 0});
                    continue;
                }
            }
            if (f.stage == 1) {
                f.stage = 2;
                if (second.kind == K_NODE && sched_of[second.payload] == UINT32_MAX) {
                    stack.push_back(Frame{(uint32_t)second.payload, 0});
                    continue;
                }
            }
            const uint32_t id = f.node;
            stack.pop_back();
            Node n{N.op, 0, final_operand(N.a), final_operand(N.b)};
            sched_of[id] = (uint32_t)O.nodes.size();
            O.nodes.push_back(n);
            O.num_ops++;
        }
        O.nodes.push_back(Node{OP_WRITE, t, final_operand(terms[t]), Operand{0, 0}});
    }

    // slots: linear scan over the schedule; an operand read for the last time frees its slot before the result takes one
    const size_t S = O.nodes.size();
    std::vector<uint32_t> last_use(S, 0);
    for (size_t i = 0; i < S; i++) {
        const Node &n = O.nodes[i];
        if (n.a.kind == BJ_IDX_TEMPORARY) last_use[n.a.index] = (uint32_t)i;
        const bool binary = n.op == BJ_OP_ADD || n.op == BJ_OP_SUB || n.op == BJ_OP_MUL;
        if (binary && n.b.kind == BJ_IDX_TEMPORARY) last_use[n.b.index] = (uint32_t)i;
    }
    O.slot_of.assign(S, 0);
    std::priority_queue<uint32_t, std::vector<uint32_t>, std::greater<uint32_t>> free_slots;
    for (size_t i = 0; i < S; i++) {
        Node &n = O.nodes[i];
        const bool binary = n.op == BJ_OP_ADD || n.op == BJ_OP_SUB || n.op == BJ_OP_MUL;
        if (n.a.kind == BJ_IDX_TEMPORARY && last_use[n.a.index] == i) free_slots.push(O.slot_of[n.a.index]);
        if (binary && n.b.kind == BJ_IDX_TEMPORARY && last_use[n.b.index] == i && !(n.a.kind == BJ_IDX_TEMPORARY && n.a.index == n.b.index))
            free_slots.push(O.slot_of[n.b.index]);
        if (n.op == OP_WRITE) continue;
        uint32_t s;
        if (!free_slots.empty()) {
            s = free_slots.top();
            free_slots.pop();
        } else {
            s = O.num_slots++;
        }
        O.slot_of[i] = s;
        n.dst = s;
    }
    return BJ_OK;
}

std::string emit_body(const Program &P, const char *indent) {
    std::string out;
    char buf[256];
    auto name = [&](const Operand &x) -> std::string {
        char b[48];
        switch (x.kind) {
            case BJ_IDX_VARIABLE_POLY: snprintf(b, sizeof b, "VAR(%u)", x.index); break;
            case BJ_IDX_WITNESS_POLY: snprintf(b, sizeof b, "WIT(%u)", x.index); break;
            case BJ_IDX_CONSTANT_POLY: snprintf(b, sizeof b, "CON(%u)", x.index); break;
            case BJ_IDX_TEMPORARY: snprintf(b, sizeof b, "n%u", x.index); break;
            default: snprintf(b, sizeof b, "0x%016llxULL", (unsigned long long)P.values[x.index]); break;
        }
        return b;
    };
    for (size_t i = 0; i < P.nodes.size(); i++) {
        const Node &n = P.nodes[i];
        const std::string a = name(n.a);
        switch (n.op) {
            case BJ_OP_ADD: snprintf(buf, sizeof buf, "%sconst u64 n%zu = gl::add(%s, %s);\n", indent, i, a.c_str(), name(n.b).c_str()); break;
            case BJ_OP_SUB: snprintf(buf, sizeof buf, "%sconst u64 n%zu = gl::sub(%s, %s);\n", indent, i, a.c_str(), name(n.b).c_str()); break;
            case BJ_OP_MUL: snprintf(buf, sizeof buf, "%sconst u64 n%zu = gl::mul(%s, %s);\n", indent, i, a.c_str(), name(n.b).c_str()); break;
            case BJ_OP_DOUBLE: snprintf(buf, sizeof buf, "%sconst u64 n%zu = gl::add(%s, %s);\n", indent, i, a.c_str(), a.c_str()); break;
            case BJ_OP_NEGATE: snprintf(buf, sizeof buf, "%sconst u64 n%zu = gl::neg(%s);\n", indent, i, a.c_str()); break;
            case BJ_OP_SQUARE: snprintf(buf, sizeof buf, "%sconst u64 n%zu = gl::sqr(%s);\n", indent, i, a.c_str()); break;
            case BJ_OP_INVERSE: snprintf(buf, sizeof buf, "%sconst u64 n%zu = inv_pow(%s);\n", indent, i, a.c_str()); break;
            default: snprintf(buf, sizeof buf, "%sterm[%u] = %s;\n", indent, n.dst, a.c_str()); break;
        }
        out += buf;
    }
    return out;
}

}  // namespace canon
}  // namespace bj

extern "C" int bj_gate_program_canonical_info(const bj_gate_program *program, uint64_t fp[2], uint32_t *num_slots, uint32_t *num_ops,
                                              uint32_t *extents3) {
    bj::canon::Program P;
    std::string err;
    if (int rc = bj::canon::canonicalize(program, &P, &err)) return rc;
    if (fp) {
        fp[0] = P.fp[0];
        fp[1] = P.fp[1];
    }
    if (num_slots) *num_slots = P.num_slots;
    if (num_ops) *num_ops = P.num_ops;
    if (extents3) {
        extents3[0] = P.var_extent;
        extents3[1] = P.const_extent;
        extents3[2] = P.wit_extent;
    }
    return BJ_OK;
}

extern "C" size_t bj_gate_program_emit_body(const bj_gate_program *program, char *out, size_t cap) {
    bj::canon::Program P;
    std::string err;
    if (bj::canon::canonicalize(program, &P, &err)) return 0;
    const std::string s = bj::canon::emit_body(P, "    ");
    if (out && cap > s.size()) memcpy(out, s.c_str(), s.size() + 1);
    return s.size() + 1;
}
