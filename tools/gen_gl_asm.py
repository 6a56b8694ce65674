#!/usr/bin/env python3
"""Generates era_boojum_amd/csrc/gl_asm.inc: hand-scheduled gfx950 instruction sequences for the lazy ("weak residue")
Goldilocks arithmetic of the NTT and Poseidon2 kernels, as inline-asm device functions.

Why a generator: the sequences write HALVES of 64-bit register pairs (v_mov into the high word of a multiply-add addend,
carry chains word by word), which compiler-allocated inline-asm operands cannot express on this target (no sub-register
operand modifier) — temporaries are therefore fixed physical VGPRs listed as clobbers; and gfx950 needs 2 wait states between
a VALU write of an SGPR pair (carry / borrow masks) and a VALU read of it, which the compiler cannot insert inside an asm
string.  The generator interleaves independent chains so that every such dependency has two other instructions in
between, checks that, and pads with s_nop only where it must.

    python tools/gen_gl_asm.py          # rewrites era_boojum_amd/csrc/gl_asm.inc (also run by era_boojum_amd/build.py)
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "era_boojum_amd", "csrc", "gl_asm.inc")

# first physical VGPR of the temporaries; low enough that kernels with ~64..80 live registers keep their occupancy
TEMP_BASE = 48


class Chain:
    """One dependent instruction sequence.  Instructions are (text, sgpr_defs, sgpr_uses)."""

    def __init__(self, name):
        self.name = name
        self.ins = []

    def add(self, text, defs=(), uses=()):
        self.ins.append((text, tuple(defs), tuple(uses)))


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def mul_chain(ch, a0, a1, b0, b1, T, X, H, Z, cm, c, bq, rare):
    """X <- weak(a * b).  T, X, H, Z: even physical VGPR numbers of pairs (Z.hi must hold 0).  cm, c, bq, rare: SGPR pair
    operand names.  `rare` receives bq & ~c (lanes whose result needs the slow path)."""
    ch.add("v_mad_u64_u32 %s, vcc, %s, %s, 0" % (pair(T), a0, b0))
    ch.add("v_mov_b32 v%d, v%d" % (Z, T + 1))
    ch.add("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (pair(X), a0, b1, pair(Z)))
    ch.add("v_mad_u64_u32 %s, %s, %s, %s, %s" % (pair(X), cm, a1, b0, pair(X)), defs=[cm])
    ch.add("v_mov_b32 v%d, v%d" % (Z, X + 1))
    ch.add("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (pair(H), a1, b1, pair(Z)))
    ch.add("v_mov_b32 v%d, v%d" % (T + 1, X))
    ch.add("v_mad_u64_u32 %s, %s, v%d, -1, %s" % (pair(X), c, H, pair(T)), defs=[c])
    ch.add("v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (X, bq, X, H + 1, cm), defs=[bq], uses=[cm])
    ch.add("v_cndmask_b32 v%d, 0, 1, %s" % (H, c), uses=[c])
    ch.add("v_subb_co_u32 v%d, %s, v%d, 0, %s" % (X + 1, bq, X + 1, bq), defs=[bq], uses=[bq])
    ch.add("v_mad_u64_u32 %s, vcc, v%d, -1, %s" % (pair(X), H, pair(X)))
    # SALU (own issue port): the borrow without a carry is the one case the sequence does not finish itself
    ch.add("s_andn2_b64 %s, %s, %s" % (rare, bq, c), defs=[], uses=[])


def addsub_chain(ch, u0, u1, t0, t1, S, D, k0, k1, out_s, out_d, sA, sB, sC, sD):
    """out_s <- weak(u + t), out_d <- weak(u - t); S, D: physical pairs for the raw sum / difference, k0 / k1: physical
    32-bit temporaries for the 0/1 carry and borrow; sC / sD receive the lanes that wrapped twice (slow path)."""
    ch.add("v_add_co_u32 v%d, %s, %s, %s" % (S, sA, u0, t0), defs=[sA])
    ch.add("v_sub_co_u32 v%d, %s, %s, %s" % (D, sB, u0, t0), defs=[sB])
    ch.add("v_addc_co_u32 v%d, %s, %s, %s, %s" % (S + 1, sA, u1, t1, sA), defs=[sA], uses=[sA])
    ch.add("v_subb_co_u32 v%d, %s, %s, %s, %s" % (D + 1, sB, u1, t1, sB), defs=[sB], uses=[sB])
    ch.add("v_cndmask_b32 v%d, 0, 1, %s" % (k0, sA), uses=[sA])
    ch.add("v_cndmask_b32 v%d, 0, 1, %s" % (k1, sB), uses=[sB])
    ch.add("v_mad_u64_u32 %s, %s, v%d, -1, %s" % (out_s, sC, k0, pair(S)), defs=[sC])       # + EPS where the sum wrapped
    ch.add("v_sub_co_u32 v%d, %s, v%d, v%d" % (D + 1, sD, D + 1, k1), defs=[sD])            # - EPS = - 2^32 + 1 where it borrowed
    ch.add("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (out_d, k1, pair(D)))


def interleave(chains):
    """Round-robin merge, then make sure every VALU read of an SGPR pair written by a VALU instruction has at least two
    instructions in between (s_nop otherwise)."""
    merged = []
    idx = [0] * len(chains)
    while any(i < len(c.ins) for i, c in zip(idx, chains)):
        for k, c in enumerate(chains):
            if idx[k] < len(c.ins):
                merged.append(c.ins[idx[k]])
                idx[k] += 1
    out = []
    last_def = {}
    for text, defs, uses in merged:
        need = 0
        for s in uses:
            if s in last_def:
                gap = len(out) - last_def[s] - 1
                need = max(need, 2 - gap)
        if need > 0:
            out.append(("s_nop %d" % (need - 1), (), ()))
            if need == 2:   # s_nop 1 = two wait states but one slot in our position count: count it twice
                out.append(("", (), ()))
        out.append((text, defs, uses))
        for s in defs:
            last_def[s] = len(out) - 1
    return [t for t, _, _ in out if t]


def asm_block(lines):
    return "\n".join('        "%s\\n\\t"' % l for l in lines[:-1]) + '\n        "%s"' % lines[-1]


def clobbers(lo, hi):
    return ", ".join('"v%d"' % r for r in range(lo, hi))


def gen_butterfly2():
    B = TEMP_BASE
    # chain A: T=B, X=B+2, H=B+4, Z=B+6 ; chain B: T=B+8, X=B+10, H=B+12, Z=B+14
    chains = []
    for k, base in enumerate((B, B + 8)):
        ch = Chain("bf%d" % k)
        T, X, H, Z = base, base + 2, base + 4, base + 6
        sfx = "ab"[k]
        mul_chain(ch, "%%[v%s0]" % sfx, "%%[v%s1]" % sfx, "%%[w%s0]" % sfx, "%%[w%s1]" % sfx, T, X, H, Z,
                  "%%[p%s]" % sfx, "%%[q%s]" % sfx, "%%[r%s]" % sfx, "%%[x%s]" % sfx)
        # the product sits in X; T and H are free again: raw sum in T, raw difference in H, 0/1 flags in Z.lo and X.lo/X.hi
        addsub_chain(ch, "%%[u%s0]" % sfx, "%%[u%s1]" % sfx, "v%d" % X, "v%d" % (X + 1), T, H, Z, X, "%%[s%s]" % sfx, "%%[d%s]" % sfx,
                     "%%[p%s]" % sfx, "%%[q%s]" % sfx, "%%[r%s]" % sfx, "%%[y%s]" % sfx)
        chains.append(ch)
    # NB addsub_chain uses k1 = X (v X.lo) as the borrow flag while t0 = X.lo is read by its first two instructions only
    lines = ["v_mov_b32 v%d, 0" % (B + 7), "v_mov_b32 v%d, 0" % (B + 15)] + interleave(chains)
    # the flag in Z.lo (k0) must not disturb Z.hi = 0; restore nothing: Z is re-zeroed at the top of every block
    lines += ["s_or_b64 %[xa], %[xa], %[ra]", "s_or_b64 %[xb], %[xb], %[rb]", "s_or_b64 %[xa], %[xa], %[ya]", "s_or_b64 %[xb], %[xb], %[yb]",
              "s_or_b64 %[xa], %[xa], %[xb]"]
    return lines, clobbers(B, B + 16)


def gen_addsub2():
    B = TEMP_BASE
    chains = []
    for k, base in enumerate((B, B + 6)):
        ch = Chain("as%d" % k)
        sfx = "ab"[k]
        addsub_chain(ch, "%%[u%s0]" % sfx, "%%[u%s1]" % sfx, "%%[v%s0]" % sfx, "%%[v%s1]" % sfx, base, base + 2, base + 4, base + 5,
                     "%%[s%s]" % sfx, "%%[d%s]" % sfx, "%%[p%s]" % sfx, "%%[q%s]" % sfx, "%%[r%s]" % sfx, "%%[y%s]" % sfx)
        chains.append(ch)
    lines = interleave(chains)
    lines += ["s_or_b64 %[ra], %[ra], %[ya]", "s_or_b64 %[rb], %[rb], %[yb]", "s_or_b64 %[ra], %[ra], %[rb]"]
    return lines, clobbers(B, B + 12)


HEADER = '''// GENERATED by tools/gen_gl_asm.py — do not edit.  Hand-scheduled gfx950 sequences for lazy Goldilocks arithmetic.
// Values are "weak" residues: any u64 congruent to the field element.  Every function returns a wave-uniform mask of the
// lanes (probability ~2^-32 per operation on random data) whose result the straight-line sequence did NOT finish: a
// second wrap of an addition / subtraction, or the borrow-without-carry case of the product (gl.h, mul_weak); the
// caller recomputes those through the canonical operators.  Temporaries are the fixed VGPRs v%d.. (clobbers).
#pragma once
'''


def main():
    b2, b2c = gen_butterfly2()
    a2, a2c = gen_addsub2()
    src = HEADER % TEMP_BASE
    src += '''
namespace gl {
#if defined(__HIP_DEVICE_COMPILE__)
// two radix-2 butterflies side by side:  (u, v) <- (u + v * w, u - v * w)  on weak residues
__device__ __forceinline__ u64 butterfly2_weak_asm(u64 ua, u64 va, u64 wa, u64 ub, u64 vb, u64 wb, u64 &sa, u64 &da, u64 &sb, u64 &db) {
    u64 pa, qa, ra, xa, ya, pb, qb, rb, xb, yb;
    asm(
%s
        : [sa] "=&v"(sa), [da] "=&v"(da), [sb] "=&v"(sb), [db] "=&v"(db), [pa] "=&s"(pa), [qa] "=&s"(qa), [ra] "=&s"(ra),
          [xa] "=&s"(xa), [ya] "=&s"(ya), [pb] "=&s"(pb), [qb] "=&s"(qb), [rb] "=&s"(rb), [xb] "=&s"(xb), [yb] "=&s"(yb)
        : [ua0] "v"(lo32(ua)), [ua1] "v"(hi32(ua)), [va0] "v"(lo32(va)), [va1] "v"(hi32(va)), [wa0] "v"(lo32(wa)), [wa1] "v"(hi32(wa)),
          [ub0] "v"(lo32(ub)), [ub1] "v"(hi32(ub)), [vb0] "v"(lo32(vb)), [vb1] "v"(hi32(vb)), [wb0] "v"(lo32(wb)), [wb1] "v"(hi32(wb))
        : "vcc", "scc", %s);
    return xa;
}
// the same without the multiplication (twiddle 1):  (u, v) <- (u + v, u - v)
__device__ __forceinline__ u64 addsub2_weak_asm(u64 ua, u64 va, u64 ub, u64 vb, u64 &sa, u64 &da, u64 &sb, u64 &db) {
    u64 pa, qa, ra, ya, pb, qb, rb, yb;
    asm(
%s
        : [sa] "=&v"(sa), [da] "=&v"(da), [sb] "=&v"(sb), [db] "=&v"(db), [pa] "=&s"(pa), [qa] "=&s"(qa), [ra] "=&s"(ra),
          [ya] "=&s"(ya), [pb] "=&s"(pb), [qb] "=&s"(qb), [rb] "=&s"(rb), [yb] "=&s"(yb)
        : [ua0] "v"(lo32(ua)), [ua1] "v"(hi32(ua)), [va0] "v"(lo32(va)), [va1] "v"(hi32(va)),
          [ub0] "v"(lo32(ub)), [ub1] "v"(hi32(ub)), [vb0] "v"(lo32(vb)), [vb1] "v"(hi32(vb))
        : "vcc", "scc", %s);
    return ra;
}
#endif
}  // namespace gl
''' % (asm_block(b2), b2c, asm_block(a2), a2c)
    old = open(OUT).read() if os.path.exists(OUT) else None
    if old != src:
        with open(OUT, "w") as f:
            f.write(src)
    return OUT


if __name__ == "__main__":
    print(main())
