/* ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h).
 * Pointwise helpers restated from cs/implementations/utils.rs:
 *   batch_inverse                      utils.rs:405-470   (Montgomery trick; result = elementwise inverse)
 *   batch_inverse_inplace_in_extension utils.rs:472-634
 */
#include "oracle.h"
#include <stdlib.h>

void orc_batch_inverse(const uint64_t *in, uint64_t *out, size_t n) {
    if (!n) return;
    gl_t *pre = (gl_t *)malloc(n * sizeof(gl_t));
    gl_t acc = 1;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = gl_mul(acc, gl_canon(in[i])); }
    gl_t inv = gl_inv(acc);
    for (size_t i = n; i-- > 0;) { gl_t x = gl_canon(in[i]); out[i] = gl_mul(inv, pre[i]); inv = gl_mul(inv, x); }
    free(pre);
}
void orc_ext_batch_inverse(const uint64_t *in0, const uint64_t *in1, uint64_t *o0, uint64_t *o1, size_t n) {
    for (size_t i = 0; i < n; i++) {
        gl2_t r = gl2_inv(gl2_make(gl_canon(in0[i]), gl_canon(in1[i])));
        o0[i] = r.c0; o1[i] = r.c1;
    }
}
