// Host-side Fiat–Shamir for the prover (tiny data, order-critical): Poseidon2 permutation on the CPU side of the
// product plus the algebraic sponge transcript and the query-bit buffer.  This is product code (it is linked into
// libboojum_hip.so) and deliberately independent of oracle/ — the tests compare the two.
//   permutation   src/implementations/poseidon2/state_generic_impl.rs:128-233, suggested_mds.rs:21-103
//   transcript    src/cs/implementations/transcript.rs:48-131 (AlgebraicSpongeBasedTranscript), :144-151 (Poseidon2)
//   BoolsBuffer   src/cs/implementations/transcript.rs:369-417; index split prover.rs:2161-2182
#pragma once
#include "gl.cuh"
#include "poseidon_rc.inc"
#include <vector>

namespace bj {
namespace host {

using gl::u64;

inline const u64 *rc_table() {
    static const u64 RC[BJ_POSEIDON_NUM_RC] = BJ_POSEIDON_RC_TABLE;
    return RC;
}

inline u64 pow7(u64 x) {
    u64 x2 = gl::sqr(x), x3 = gl::mul(x2, x), x4 = gl::sqr(x2);
    return gl::mul(x4, x3);
}
inline void m4(u64 *x) {
    u64 t0 = gl::add(x[0], x[1]), t1 = gl::add(x[2], x[3]);
    u64 t2 = gl::add(gl::dbl(x[1]), t1), t3 = gl::add(gl::dbl(x[3]), t0);
    u64 t4 = gl::add(gl::dbl(gl::dbl(t1)), t3), t5 = gl::add(gl::dbl(gl::dbl(t0)), t2);
    x[0] = gl::add(t3, t5); x[1] = t5; x[2] = gl::add(t2, t4); x[3] = t4;
}
inline void ext_mds(u64 *s) {
    m4(s); m4(s + 4); m4(s + 8);
    for (int j = 0; j < 4; j++) {
        u64 sum = gl::add(gl::add(s[j], s[4 + j]), s[8 + j]);
        s[j] = gl::add(s[j], sum); s[4 + j] = gl::add(s[4 + j], sum); s[8 + j] = gl::add(s[8 + j], sum);
    }
}
inline void poseidon2_permutation(u64 *s) {
    static const unsigned SH[12] = {4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12};
    const u64 *RC = rc_table();
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
    ext_mds(s);
    int r = 0;
    for (int i = 0; i < 4; i++, r++) {
        for (int k = 0; k < 12; k++) s[k] = pow7(gl::add(s[k], RC[12 * r + k]));
        ext_mds(s);
    }
    for (int i = 0; i < 22; i++, r++) {
        s[0] = pow7(gl::add(s[0], RC[12 * r]));
        u64 sum = 0;
        for (int k = 0; k < 12; k++) sum = gl::add(sum, s[k]);
        for (int k = 0; k < 12; k++) s[k] = gl::add(gl::mul_pow2(s[k], SH[k]), sum);
    }
    for (int i = 0; i < 4; i++, r++) {
        for (int k = 0; k < 12; k++) s[k] = pow7(gl::add(s[k], RC[12 * r + k]));
        ext_mds(s);
    }
}

// Poseidon (v1), the round function of GoldilocksPoisedonTranscript (transcript.rs:133-141) which the reference's SHA-256
// bench script uses next to the Poseidon2 tree hasher (gadgets/sha256/mod.rs:289-293):
// implementations/poseidon_goldilocks_naive.rs:10-160 — every round adds its 12 constants, x^7 on all (full) or on
// element 0 (partial), then the circulant MDS with power-of-two entries, here as shift-and-add on 128-bit integers.
// The reference has no known-answer vector for it: checked against two independent restatements only.
inline void poseidon1_permutation(u64 *s) {
    static const unsigned EXPS[12] = {0, 0, 1, 0, 3, 5, 1, 8, 12, 3, 16, 10};
    const u64 *RC = rc_table();
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        for (int k = 0; k < 12; k++) s[k] = gl::add(s[k], RC[12 * r + k]);
        for (int k = 0; k < (full ? 12 : 1); k++) s[k] = pow7(s[k]);
        u64 out[12];
        for (int row = 0; row < 12; row++) {
            u64 lo = 0, hi = 0;   // 128-bit accumulator; the sum stays below 2^81
            for (int col = 0; col < 12; col++) {
                const unsigned e = EXPS[(col + 12 - row) % 12];
                const u64 add_lo = s[col] << e, add_hi = e ? s[col] >> (64 - e) : 0;
                const u64 t = lo + add_lo;
                hi += add_hi + (t < lo ? 1 : 0);
                lo = t;
            }
            out[row] = gl::add(gl::canon(lo), gl::mul(hi, 0xFFFFFFFFULL));   // 2^64 = 2^32 - 1
        }
        for (int k = 0; k < 12; k++) s[k] = out[k];
    }
}

struct Transcript {
    int kind = 1;                 // BJ_TRANSCRIPT_POSEIDON2 = 1, BJ_TRANSCRIPT_POSEIDON = 2
    void permute() {
        if (kind == 2) poseidon1_permutation(state);
        else poseidon2_permutation(state);
    }
    u64 state[12] = {0};
    std::vector<u64> buffer;
    u64 avail[8];
    size_t avail_pos = 0, avail_len = 0;

    void absorb(const u64 *els, size_t n) {
        for (size_t i = 0; i < n; i++) buffer.push_back(gl::canon(els[i]));
    }
    u64 challenge() {
        if (buffer.empty()) {
            if (avail_pos < avail_len) return avail[avail_pos++];
            permute();
        } else {
            buffer.push_back(1);
            while (buffer.size() % 8) buffer.push_back(0);
            for (size_t i = 0; i < buffer.size(); i += 8) {
                for (int k = 0; k < 8; k++) state[k] = buffer[i + k];
                permute();
            }
            buffer.clear();
        }
        for (int k = 0; k < 8; k++) avail[k] = state[k];
        avail_len = 8;
        avail_pos = 0;
        return avail[avail_pos++];
    }
};

struct BoolsBuffer {
    std::vector<unsigned char> bits;
    size_t pos = 0;
    unsigned max_needed = 0;
    u64 query_index(Transcript &t, unsigned log_n, unsigned log_lde) {
        unsigned need = log_n + log_lde;
        while (bits.size() - pos < need) {
            bits.erase(bits.begin(), bits.begin() + pos);
            pos = 0;
            u64 x = gl::canon(t.challenge());
            for (unsigned i = 0; i < 64 - max_needed; i++) bits.push_back((x >> i) & 1);
        }
        u64 inner = 0, coset = 0;
        for (unsigned i = 0; i < log_n; i++) inner |= (u64)bits[pos + i] << i;
        for (unsigned i = 0; i < log_lde; i++) coset |= (u64)bits[pos + log_n + i] << i;
        pos += need;
        return (coset << log_n) + inner;
    }
};

}  // namespace host
}  // namespace bj

// the object behind the opaque `bj_transcript` of include/boojum_hip.h (shared by fri_prover.hip and prover.hip)
struct bj_transcript {
    bj::host::Transcript t;
    bj::host::BoolsBuffer bools;
};
