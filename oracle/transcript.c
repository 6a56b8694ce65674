/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Algebraic Fiat–Shamir transcript over the Poseidon2 sponge and query-bit extraction, restated from
 *   AlgebraicSpongeBasedTranscript   cs/implementations/transcript.rs:48-131
 *   GoldilocksPoisedon2Transcript    cs/implementations/transcript.rs:144-151 (the fixture's transcript)
 *   BoolsBuffer::get_bits            cs/implementations/transcript.rs:369-417
 *   query index from bits            cs/implementations/prover.rs:2161-2182
 * Pinned by replaying the golden proof.json (query 0 index must be 1192677).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

struct orc_transcript {
    int kind;                    /* 2: Poseidon (v1) round function (GoldilocksPoisedonTranscript, transcript.rs:133-141), else Poseidon2 */
    gl_t state[12];
    gl_t *buf; size_t buf_len, buf_cap;
    gl_t avail[8]; size_t avail_pos, avail_len;
};
orc_transcript *orc_transcript_new(void) { return (orc_transcript *)calloc(1, sizeof(orc_transcript)); }
orc_transcript *orc_transcript_new_kind(int kind) {
    orc_transcript *t = orc_transcript_new();
    if (t) t->kind = kind;
    return t;
}
static void round_function(orc_transcript *t) {
    if (t->kind == 2) orc_poseidon_permutation(t->state);
    else orc_poseidon2_permutation(t->state);
}
void orc_transcript_free(orc_transcript *t) { if (t) { free(t->buf); free(t); } }
void orc_transcript_absorb(orc_transcript *t, const uint64_t *els, size_t n) {
    if (t->buf_len + n + 9 > t->buf_cap) {
        t->buf_cap = 2 * (t->buf_len + n) + 16;
        t->buf = (gl_t *)realloc(t->buf, t->buf_cap * sizeof(gl_t));
    }
    for (size_t i = 0; i < n; i++) t->buf[t->buf_len++] = gl_canon(els[i]);
}
uint64_t orc_transcript_challenge(orc_transcript *t) {
    if (t->buf_len == 0) {
        if (t->avail_pos < t->avail_len) return t->avail[t->avail_pos++];
        round_function(t);                                   /* run_round_function + try_get_commitment */
    } else {
        t->buf[t->buf_len++] = 1;                            /* rescue-prime style padding */
        while (t->buf_len % 8) t->buf[t->buf_len++] = 0;
        for (size_t i = 0; i < t->buf_len; i += 8) {         /* overwrite absorption, one permutation per block */
            memcpy(t->state, t->buf + i, 8 * sizeof(gl_t));
            round_function(t);
        }
        t->buf_len = 0;
    }
    memcpy(t->avail, t->state, 8 * sizeof(gl_t));
    t->avail_len = 8; t->avail_pos = 0;
    return t->avail[t->avail_pos++];
}

struct orc_bools { unsigned char bits[4096]; size_t pos, len; unsigned max_needed; };
orc_bools *orc_bools_new(unsigned max_needed_bits) {
    orc_bools *b = (orc_bools *)calloc(1, sizeof(orc_bools));
    b->max_needed = max_needed_bits;
    return b;
}
void orc_bools_free(orc_bools *b) { free(b); }
uint64_t orc_query_index(orc_bools *b, orc_transcript *t, unsigned log_n, unsigned log_lde) {
    unsigned need = log_n + log_lde;
    while (b->len - b->pos < need) {
        if (b->pos) { memmove(b->bits, b->bits + b->pos, b->len - b->pos); b->len -= b->pos; b->pos = 0; }
        uint64_t x = gl_canon(orc_transcript_challenge(t));
        for (unsigned i = 0; i < 64 - b->max_needed; i++) b->bits[b->len++] = (x >> i) & 1;
    }
    uint64_t inner = 0, coset = 0;
    for (unsigned i = 0; i < log_n; i++) inner |= (uint64_t)b->bits[b->pos + i] << i;
    for (unsigned i = 0; i < log_lde; i++) coset |= (uint64_t)b->bits[b->pos + log_n + i] << i;
    b->pos += need;
    return (coset << log_n) + inner;
}
