// Host-side Fiat–Shamir for the prover (tiny data, order-critical): Poseidon2 permutation on the CPU side of the
// product plus the algebraic sponge transcript and the query-bit buffer.  This is product code (it is linked into
// libboojum_hip.so) and deliberately independent of oracle/ — the tests compare the two.
//   permutation   src/implementations/poseidon2/state_generic_impl.rs:128-233, suggested_mds.rs:21-103
//   transcript    src/cs/implementations/transcript.rs:48-131 (AlgebraicSpongeBasedTranscript), :144-151 (Poseidon2)
//   BoolsBuffer   src/cs/implementations/transcript.rs:369-417; index split prover.rs:2161-2182
#pragma once
#include "gl.h"
#include "poseidon_rc.inc"
#include <cstdint>
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
        // out[row] = sum_col s[col] << EXPS[(col - row) mod 12] = sum_d s[(row + d) mod 12] << EXPS[d]: for a fixed d
        // the twelve rows read a rotated copy of the state and shift by one constant, so the loops over rows vectorise.
        // Low and high words accumulate apart (each sum < 12 * 2^48), no 128-bit arithmetic.
        u64 lo2[24], hi2[24], acc_lo[12] = {0}, acc_hi[12] = {0};
        for (int k = 0; k < 12; k++) {
            lo2[k] = lo2[k + 12] = s[k] & 0xFFFFFFFFULL;
            hi2[k] = hi2[k + 12] = s[k] >> 32;
        }
        for (int d = 0; d < 12; d++) {
            const unsigned e = EXPS[d];
            for (int row = 0; row < 12; row++) {
                acc_lo[row] += lo2[row + d] << e;
                acc_hi[row] += hi2[row + d] << e;
            }
        }
        for (int row = 0; row < 12; row++) {
            // acc_lo + acc_hi * 2^32 = acc_lo + (hi part of acc_hi) * 2^64 + (lo part of acc_hi) * 2^32
            const unsigned __int128 t = (unsigned __int128)acc_lo[row] + ((unsigned __int128)(acc_hi[row] & 0xFFFFFFFFULL) << 32) +
                                        (unsigned __int128)(acc_hi[row] >> 32) * 0xFFFFFFFFULL;
            s[row] = gl::reduce128((u64)(t >> 64), (u64)t);
        }
    }
}

// Blake2s-256 (RFC 7693, unkeyed) with the update / finalize_reset interface of the `blake2` crate the reference uses.
struct Blake2s {
    uint32_t h[8];
    unsigned char buf[64];
    size_t buf_len = 0;
    uint64_t t = 0;
    Blake2s() { reset(); }
    void reset() {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        for (int i = 0; i < 8; i++) h[i] = IV[i];
        h[0] ^= 0x01010020u;
        buf_len = 0;
        t = 0;
    }
    static uint32_t rotr(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }
    void compress(const unsigned char *block, bool last) {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        static const unsigned char SIGMA[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++)
            m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) | ((uint32_t)block[4 * i + 3] << 24);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[8 + i] = IV[i];
        }
        v[12] ^= (uint32_t)t;
        v[13] ^= (uint32_t)(t >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 12);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const unsigned char *s = SIGMA[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[8 + i];
    }
    void update(const unsigned char *data, size_t n) {
        while (n) {
            if (buf_len == 64) {   // a full buffer is only compressed when more input follows (the last block is special)
                t += 64;
                compress(buf, false);
                buf_len = 0;
            }
            size_t take = 64 - buf_len < n ? 64 - buf_len : n;
            for (size_t i = 0; i < take; i++) buf[buf_len + i] = data[i];
            buf_len += take;
            data += take;
            n -= take;
        }
    }
    void finalize_reset(unsigned char out[32]) {
        t += buf_len;
        for (size_t i = buf_len; i < 64; i++) buf[i] = 0;
        compress(buf, true);
        for (int i = 0; i < 8; i++)
            for (int b = 0; b < 4; b++) out[4 * i + b] = (unsigned char)(h[i] >> (8 * b));
        reset();
    }
};

// Keccak-256 with the original padding (sha3::Keccak256), same interface.
struct Keccak256 {
    uint64_t a[25];
    unsigned char buf[136];
    size_t buf_len = 0;
    Keccak256() { reset(); }
    void reset() {
        for (int i = 0; i < 25; i++) a[i] = 0;
        buf_len = 0;
    }
    static uint64_t rotl(uint64_t x, unsigned n) { return n ? (x << n) | (x >> (64 - n)) : x; }
    void permute() {
        static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        static const unsigned ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
        for (int r = 0; r < 24; r++) {
            uint64_t c[5], b[25];
            for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
            for (int x = 0; x < 5; x++) {
                const uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
                for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
            }
            for (int x = 0; x < 5; x++)
                for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(a[x + 5 * y], ROT[x][y]);
            for (int y = 0; y < 25; y += 5)
                for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
            a[0] ^= RC[r];
        }
    }
    void absorb_block() {
        for (int i = 0; i < 17; i++) {
            uint64_t w = 0;
            for (int b = 0; b < 8; b++) w |= (uint64_t)buf[8 * i + b] << (8 * b);
            a[i] ^= w;
        }
        permute();
        buf_len = 0;
    }
    void update(const unsigned char *data, size_t n) {
        for (size_t i = 0; i < n; i++) {
            buf[buf_len++] = data[i];
            if (buf_len == 136) absorb_block();
        }
    }
    void finalize_reset(unsigned char out[32]) {
        for (size_t i = buf_len; i < 136; i++) buf[i] = 0;
        buf[buf_len] ^= 0x01;
        buf[135] ^= 0x80;
        absorb_block();
        for (int i = 0; i < 4; i++)
            for (int b = 0; b < 8; b++) out[8 * i + b] = (unsigned char)(a[i] >> (8 * b));
        reset();
    }
};

struct Transcript {
    int kind = 1;   // BJ_TRANSCRIPT_POSEIDON2 = 1, BJ_TRANSCRIPT_POSEIDON = 2, BJ_TRANSCRIPT_BLAKE2S = 3, BJ_TRANSCRIPT_KECCAK256 = 4
    bool is_bytes() const { return kind == 3 || kind == 4; }
    // --- byte transcripts (Blake2sTranscript / Keccak256Transcript, transcript.rs:155-372)
    Blake2s inner;
    Keccak256 inner_k;
    std::vector<unsigned char> bytes, avail_bytes;
    void reseed() {
        unsigned char out[32];
        if (kind == 4) {
            inner_k.finalize_reset(out);
            inner_k.update(out, 32);
        } else {
            inner.finalize_reset(out);
            inner.update(out, 32);
        }
        avail_bytes.assign(out, out + 32);
    }
    void flush_bytes() {
        if (!bytes.empty()) {
            if (kind == 4)
                inner_k.update(bytes.data(), bytes.size());
            else
                inner.update(bytes.data(), bytes.size());
            bytes.clear();
            reseed();
        }
    }
    void challenge_bytes(unsigned char *out, size_t n) {   // get_challenge_bytes
        flush_bytes();
        while (avail_bytes.size() < n) reseed();
        for (size_t i = 0; i < n; i++) out[i] = avail_bytes[i];
        avail_bytes.erase(avail_bytes.begin(), avail_bytes.begin() + n);
    }
    // Merkle caps: digests of the tree hasher.  Algebraic transcripts take them as field elements, the byte transcript
    // as their 32 raw bytes each (witness_merkle_tree_cap)
    void absorb_cap(const u64 *digest_words, size_t n_words) {
        if (!is_bytes()) {
            absorb(digest_words, n_words);
            return;
        }
        for (size_t i = 0; i < n_words; i++)
            for (int b = 0; b < 8; b++) bytes.push_back((unsigned char)(digest_words[i] >> (8 * b)));
    }
    void permute() {
        if (kind == 2) poseidon1_permutation(state);
        else poseidon2_permutation(state);
    }
    u64 state[12] = {0};
    std::vector<u64> buffer;
    u64 avail[8];
    size_t avail_pos = 0, avail_len = 0;

    void absorb(const u64 *els, size_t n) {
        if (is_bytes()) {   // witness_field_elements: as_u64_reduced().to_le_bytes()
            for (size_t i = 0; i < n; i++) {
                const u64 v = gl::canon(els[i]);
                for (int b = 0; b < 8; b++) bytes.push_back((unsigned char)(v >> (8 * b)));
            }
            return;
        }
        for (size_t i = 0; i < n; i++) buffer.push_back(gl::canon(els[i]));
    }
    u64 challenge() {
        if (is_bytes()) {   // get_challenge: 8 bytes, little endian, from_u64_with_reduction
            unsigned char b8[8];
            flush_bytes();
            if (avail_bytes.empty()) reseed();
            for (int i = 0; i < 8; i++) b8[i] = avail_bytes[i];
            avail_bytes.erase(avail_bytes.begin(), avail_bytes.begin() + 8);
            u64 x = 0;
            for (int i = 0; i < 8; i++) x |= (u64)b8[i] << (8 * i);
            return gl::canon(x);
        }
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
            if (t.is_bytes()) {   // non-algebraic transcripts hand out 8 uniform bytes, all 64 bits are used (transcript.rs:398-411)
                unsigned char b8[8];
                t.challenge_bytes(b8, 8);
                u64 x = 0;
                for (int i = 0; i < 8; i++) x |= (u64)b8[i] << (8 * i);
                for (unsigned i = 0; i < 64; i++) bits.push_back((x >> i) & 1);
                continue;
            }
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
