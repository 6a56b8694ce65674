#!/usr/bin/env python3
"""Generates era_boojum_amd/csrc/p2_asm.inc: the Poseidon2 permutation (Goldilocks, t = 12) as ONE hand-scheduled gfx950
instruction stream with fixed registers, emitted as an inline-asm device function.

Why: the permutation is 67 % of a proof and integer-VALU-bound; what is left to win is instruction count and issue slots.
The compiler's version pays (a) v_mov pairs to zero-extend 32-bit words into 64-bit multiply-add addends, (b) s_nop between
dependent inline-asm statements, (c) carry chains in the linear layers, (d) a separate weak addition per round constant.
Here
  * a product is the 12-instruction chained multiply-add sequence of gl::mul_weak with a persistent zero register,
    two S-boxes interleaved so that every SGPR wait state is covered by the other chain (no s_nop in full rounds);
  * the external linear layer circ(2*M4, M4, M4) runs on the low and the high 32-bit words separately as carry-free 64-bit
    sums (out = M4 * (x_b + sum_b x_b): 24 + 24 multiply-adds give the zero-extension for free, 48 v_lshl_add_u64 do M4),
    the NEXT round's constants are folded through the matrix at generation time (M x + c = M (x + M^-1 c): block 0's share rides
    on the first multiply-add of the column sums, 16 multiply-adds per layer for the rest), and each output is folded back to a
    weak 64-bit word by one multiply-add and one add whose rare carry is fixed out of line;
  * partial rounds: the S-box chain of word 0 is interleaved with the word sums of the other eleven;
  * rounds are loops around ONE copy of the full-round body (8.5 KB of code instead of 36 KB).
State words live in v[0:23] (operands tied to physical registers), temporaries in v[24:71], scalars in s[36:101].

    python tools/gen_p2_asm.py      # rewrites era_boojum_amd/csrc/p2_asm.inc (also run by era_boojum_amd/build.py)
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "era_boojum_amd", "csrc", "p2_asm.inc")
RC_INC = os.path.join(ROOT, "era_boojum_amd", "csrc", "poseidon_rc.inc")

WAYS = int(os.environ.get("BJ_P2_WAYS", "3"))            # S-boxes interleaved in a full round (2, 3 or 4; v[24:71] holds four sets): 3 measured best by ~1 %
assert WAYS in (2, 3, 4)
COMBINE_INLINE = os.environ.get("BJ_P2_COMBINE", "") == "inline"
# On since round 4 (validated on the GPU in round 3; "=0" rebuilds the round-3 stream for A/B):
#   BJ_P2_ZERO_HOIST=1: the zero high halves of the S-box addend pairs are written once per full round (3 moves instead of 12)
#                       and once before the partial-round loop (v35 is not touched by the partial rounds' linear layer): -93 VALU;
#   BJ_P2_LATE_CONST=1: the twelve constants of the first closing full round ride on the LAST partial round's linear layer
#                       (two multiply-adds per word, as word 0's constant always does) instead of a weak addition each: -26 VALU.
ZERO_HOIST = os.environ.get("BJ_P2_ZERO_HOIST", "1") == "1"
LATE_CONST = os.environ.get("BJ_P2_LATE_CONST", "1") == "1"
SH = [4, 14, 11, 8, 0, 5, 2, 9, 13, 6, 3, 12]           # internal matrix 1 + diag(2^SH)  (poseidon2/params.rs:38-39)
P = (1 << 64) - (1 << 32) + 1

# ---- scalar registers (all clobbered) ----
S_RC = 40            # s[40:71]: the constants of the layer being applied — an external layer's 32 words (see fold_ext_constants),
                     # 12 (lo, hi) pairs for the one weak addition, one pair in a partial round
S_PTR = 36           # s[36:37]: running pointer into P2_ASM_RC
S_CNT, S_PHASE = 38, 39
S_SHIFT = 72         # s[72:77]: 2^14, 2^11, 2^8, 2^9, 2^13, 2^12 (powers above 64 are not inline constants)
S_MASK = 78          # s[78:101]: carry / borrow masks (pairs)
S_TOP = 101          # last scalar register the stream touches
SHIFT_REG = {14: S_SHIFT, 11: S_SHIFT + 1, 8: S_SHIFT + 2, 9: S_SHIFT + 3, 13: S_SHIFT + 4, 12: S_SHIFT + 5}


def sp(r):
    return "s[%d:%d]" % (r, r + 1)


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def S(k):
    return 2 * k


class Ins:
    __slots__ = ("text", "defs", "uses", "glue")

    def __init__(self, text, defs=(), uses=(), glue=False):
        self.text, self.defs, self.uses, self.glue = text, tuple(defs), tuple(uses), glue   # glue: stays right behind its predecessor


def interleave(chains):
    """Round-robin merge of instruction lists (glued instructions travel with their predecessor)."""
    groups = []
    for ch in chains:
        g = []
        for ins in ch:
            if ins.glue and g:
                g[-1].append(ins)
            else:
                g.append([ins])
        groups.append(g)
    out, idx = [], [0] * len(groups)
    while any(i < len(g) for i, g in zip(idx, groups)):
        for k, g in enumerate(groups):
            if idx[k] < len(g):
                out += g[idx[k]]
                idx[k] += 1
    return out


def hazard_pass(seq):
    """gfx950: 2 wait states between a VALU write of an SGPR pair and a VALU read of it."""
    out, last_def, pos = [], {}, 0          # pos counts issued instructions (labels take no slot)
    for ins in seq:
        need = 0
        for s in ins.uses:
            if s in last_def:
                need = max(need, 2 - (pos - last_def[s] - 1))
        if need > 0:
            out.append(Ins("s_nop %d" % (need - 1)))
            pos += need
        out.append(ins)
        if not ins.text.endswith(":"):
            for s in ins.defs:
                last_def[s] = pos
            pos += 1
    return out


class Gen:
    def __init__(self):
        self.lines = []
        self.stubs = []
        self.n_label = 0
        self.nops = 0

    def label(self, stem):
        self.n_label += 1
        return ".Lp2%s%d_%%=" % (stem, self.n_label)

    def emit(self, seq):
        for ins in hazard_pass(seq):
            if ins.text.startswith("s_nop"):
                self.nops += 1
            self.lines.append(ins.text)

    def raw(self, text):
        self.lines.append(text)

    # ------------------------------------------------------------------------------------------------ product
    def mulw(self, a, b, T, X, H, Z, out, cm, c, bq):
        """out <- weak(a * b); a, b = (lo, hi) VGPR numbers; T, X, H, Z even VGPRs of pairs (v[Z+1] = 0); out: pair (may be X)."""
        a0, a1 = a
        b0, b1 = b
        back, stub = self.label("b"), self.label("f")
        ch = [
            Ins("v_mad_u64_u32 %s, vcc, v%d, v%d, 0" % (vp(T), a0, b0)),
            Ins("v_mov_b32 v%d, v%d" % (Z, T + 1)),
            Ins("v_mad_u64_u32 %s, vcc, v%d, v%d, %s" % (vp(X), a0, b1, vp(Z))),
            Ins("v_mad_u64_u32 %s, %s, v%d, v%d, %s" % (vp(X), sp(cm), a1, b0, vp(X)), defs=[cm]),
            Ins("v_mov_b32 v%d, v%d" % (Z, X + 1)),
            Ins("v_mad_u64_u32 %s, vcc, v%d, v%d, %s" % (vp(H), a1, b1, vp(Z))),
            Ins("v_mov_b32 v%d, v%d" % (T + 1, X)),
            Ins("v_mad_u64_u32 %s, %s, v%d, -1, %s" % (vp(X), sp(c), H, vp(T)), defs=[c]),
            Ins("v_subb_co_u32 v%d, %s, v%d, v%d, %s" % (X, sp(bq), X, H + 1, sp(cm)), defs=[bq], uses=[cm]),
            Ins("v_cndmask_b32 v%d, 0, 1, %s" % (H, sp(c)), uses=[c]),
            Ins("v_subb_co_u32 v%d, %s, v%d, 0, %s" % (X + 1, sp(bq), X + 1, sp(bq)), defs=[bq], uses=[bq]),
            Ins("s_cmp_lg_u64 %s, 0" % sp(bq)),
            Ins("s_cbranch_scc1 %s" % stub, glue=True),
            Ins("%s:" % back, glue=True),
            Ins("v_mad_u64_u32 %s, vcc, v%d, -1, %s" % (vp(out), H, vp(X))),
        ]
        # the borrow without a carry (R < 2^32: ~2^-32 per product): subtract EPS once more, out of line
        self.stubs += ["%s:" % stub, "s_andn2_b64 vcc, %s, %s" % (sp(bq), sp(c)), "v_cndmask_b32 v%d, 0, -1, vcc" % (H + 1),
                       "s_nop 1", "v_sub_co_u32 v%d, vcc, v%d, v%d" % (X, X, H + 1), "s_nop 1",
                       "v_subbrev_co_u32 v%d, vcc, 0, v%d, vcc" % (X + 1, X + 1), "s_branch %s" % back]
        return ch

    def sbox(self, k, base, masks, zero=True):
        """x^7 on state word k in place; temporaries v[base : base+12), masks: three SGPR pair numbers.  zero=False: v[base+11]
        already holds 0 (ZERO_HOIST)."""
        T, H, X1, X2, X3, Z = base, base + 2, base + 4, base + 6, base + 8, base + 10
        cm, c, bq = masks
        x = (S(k), S(k) + 1)
        x2, x3, x4 = (X1, X1 + 1), (X2, X2 + 1), (X3, X3 + 1)
        ch = [Ins("v_mov_b32 v%d, 0" % (Z + 1))] if zero else []
        ch += self.mulw(x, x, T, X1, H, Z, X1, cm, c, bq)          # x^2
        ch += self.mulw(x2, x, T, X2, H, Z, X2, cm, c, bq)         # x^3
        ch += self.mulw(x2, x2, T, X3, H, Z, X3, cm, c, bq)        # x^4
        ch += self.mulw(x4, x3, T, X1, H, Z, S(k), cm, c, bq)      # x^7 -> the state word
        return ch

    # ------------------------------------------------------------------------------------------------ folding a (low, high) pair of sums
    def combine(self, A, B, out, k01, mask):
        """out <- weak(A + B * 2^32) for 64-bit sums A, B < 2^48 held in pairs v[A:A+1], v[B:B+1]; k01: a 32-bit temporary.
        B * 2^32 = B.hi * 2^64 + B.lo * 2^32 == B.hi * EPS + B.lo * 2^32: T = A + B.hi * EPS (no carry, T.hi < 2^17), then one
        add on the high word.  That add wraps only when B.lo >= 2^32 - T.hi: once in ~2^18 (partial rounds) to ~2^25 (external
        layers) words, so the "+EPS" of a wrapped word (then < 2^49: no second carry) lives out of line behind a wave-uniform
        branch — 2 VALU instructions per word instead of 4 (BJ_P2_COMBINE=inline restores the branch-free form)."""
        if COMBINE_INLINE:
            return [
                Ins("v_mad_u64_u32 %s, vcc, v%d, -1, %s" % (vp(A), B + 1, vp(A))),
                Ins("v_add_co_u32 v%d, %s, v%d, v%d" % (A + 1, sp(mask), A + 1, B), defs=[mask]),
                Ins("v_cndmask_b32 v%d, 0, 1, %s" % (k01, sp(mask)), uses=[mask]),
                Ins("v_mad_u64_u32 %s, vcc, v%d, -1, %s" % (vp(out), k01, vp(A))),
            ]
        back, stub = self.label("cb"), self.label("cf")
        self.stubs += ["%s:" % stub, "s_nop 1", "v_cndmask_b32 v%d, 0, -1, %s" % (k01, sp(mask)),
                       "v_add_co_u32 v%d, vcc, v%d, v%d" % (out, out, k01), "s_nop 1",
                       "v_addc_co_u32 v%d, vcc, 0, v%d, vcc" % (out + 1, out + 1), "s_branch %s" % back]
        return [
            Ins("v_mad_u64_u32 %s, vcc, v%d, -1, %s" % (vp(out), B + 1, vp(A))),
            Ins("v_add_co_u32 v%d, %s, v%d, v%d" % (out + 1, sp(mask), out + 1, B), defs=[mask]),
            Ins("s_cmp_lg_u64 %s, 0" % sp(mask)),
            Ins("s_cbranch_scc1 %s" % stub, glue=True),
            Ins("%s:" % back, glue=True),
        ]

    # ------------------------------------------------------------------------------------------------ external layer
    def ext_layer(self):
        """state <- circ(2*M4, M4, M4) * state + 12 constants  (suggested_mds.rs:21-103).  The constants are folded THROUGH the
        matrix (fold_ext_constants): out_b = M4 (x_b + sum_b x_b + D_b), D_0 rides on the first multiply-add of the column sums,
        blocks 1 and 2 add D_b - D_0 to their z — 16 multiply-adds per layer instead of 24 after the matrix.
        Registers: U_j (low / high plane) in v24..v39, block temporaries in v40..v63, flags v64..v66."""
        U = lambda plane, j: 24 + 8 * plane + 2 * j
        seq = []
        for j in range(4):                       # U_j = x_{0,j} + x_{1,j} + x_{2,j}, per plane (multiply-adds zero-extend for free)
            for plane in range(2):
                for b in range(3):
                    src = S(4 * b + j) + plane
                    # the sum starts from block 0's folded constant D_0j (a {half, 0} scalar pair): free
                    seq.append(Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(U(plane, j)), src, sp(S_RC + 2 * (2 * j + plane)) if b == 0 else vp(U(plane, j)))))
        self.emit(seq)
        for b in range(3):
            Z_ = lambda plane, j: 40 + 12 * plane + 2 * j        # z_j, later y_j (4 pairs per plane) ...
            T0 = lambda plane: 40 + 12 * plane + 8               # ... + t0 / t1 (2 pairs per plane)
            T1 = lambda plane: 40 + 12 * plane + 10
            seq = []
            for j in range(4):
                for plane in range(2):
                    seq.append(Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(Z_(plane, j)), S(4 * b + j) + plane, vp(U(plane, j)))))
                    if b:                        # blocks 1, 2: + (D_bj - D_0j), word halves apart
                        seq.append(Ins("v_mad_u64_u32 %s, vcc, s%d, 1, %s" % (vp(Z_(plane, j)), S_RC + 16 + 8 * (b - 1) + 2 * j + plane, vp(Z_(plane, j)))))
            m4 = [[], []]
            for plane in range(2):
                z = [Z_(plane, j) for j in range(4)]
                t0, t1 = T0(plane), T1(plane)
                la = lambda d, a, sh, c: Ins("v_lshl_add_u64 %s, %s, %d, %s" % (vp(d), vp(a), sh, vp(c)))
                m4[plane] = [
                    la(t0, z[0], 0, z[1]),       # t0 = z0 + z1
                    la(t1, z[2], 0, z[3]),       # t1 = z2 + z3
                    la(z[1], z[1], 1, t1),       # t2 = 2 z1 + t1      (in z1)
                    la(z[3], z[3], 1, t0),       # t3 = 2 z3 + t0      (in z3)
                    la(t1, t1, 2, z[3]),         # t4 = 4 t1 + t3      (in t1)  = y3
                    la(t0, t0, 2, z[1]),         # t5 = 4 t0 + t2      (in t0)  = y1
                    la(z[0], z[3], 0, t0),       # y0 = t3 + t5        (in z0)
                    la(z[2], z[1], 0, t1),       # y2 = t2 + t4        (in z2)
                ]
            seq += interleave(m4)
            y = lambda plane, j: [Z_(plane, 0), T0(plane), Z_(plane, 2), T1(plane)][j]
            comb = [self.combine(y(0, j), y(1, j), S(4 * b + j), 64 + j, S_MASK + 2 * j) for j in range(4)]
            seq += interleave(comb)
            self.emit(seq)

    def load_rc(self, n_words):
        """s[S_RC ...] <- the next n_words constants; pointer advanced."""
        off = 0
        left = 2 * n_words
        reg = S_RC
        while left:
            w = 16 if left >= 16 else 8 if left >= 8 else 4 if left >= 4 else 2
            self.raw("s_load_dwordx%d s[%d:%d], %s, 0x%x" % (w, reg, reg + w - 1, sp(S_PTR), off))
            off += 4 * w
            reg += w
            left -= w
        self.raw("s_add_u32 s%d, s%d, %d" % (S_PTR, S_PTR, 8 * n_words))
        self.raw("s_addc_u32 s%d, s%d, 0" % (S_PTR + 1, S_PTR + 1))

    # ------------------------------------------------------------------------------------------------ rounds
    def full_round(self):
        self.load_rc(16)                          # folded constants of the layer at the end of this round (arrive during the S-boxes)
        ways = WAYS                               # S-box chains issued round-robin: 12 temporaries and 3 mask pairs each
        if ZERO_HOIST:                            # the linear layer before this round used these registers: zero them once per round
            self.emit([Ins("v_mov_b32 v%d, 0" % (24 + 12 * i + 11)) for i in range(ways)])
        for k in range(0, 12, ways):
            chains = [self.sbox(k + i, 24 + 12 * i, (S_MASK + 6 * i, S_MASK + 6 * i + 2, S_MASK + 6 * i + 4), zero=not ZERO_HOIST) for i in range(ways)]
            self.emit(interleave(chains))
        self.raw("s_waitcnt lgkmcnt(0)")
        self.ext_layer()

    def partial_round(self, last=False):
        """word 0 <- (word 0)^7, then state <- (1 + diag(2^SH)) * state, + the next round's constant on word 0 (last=True, LATE_CONST:
        + the twelve constants of the full round that follows, one per word)."""
        self.load_rc(12 if last else 1)
        SL, SH_ = 48, 50                          # sums of the low / high words
        sb = self.sbox(0, 24, (S_MASK, S_MASK + 2, S_MASK + 4), zero=not ZERO_HOIST)
        sums = []
        for k in range(1, 12):
            sums.append(Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(SL), S(k), "0" if k == 1 else vp(SL))))
            sums.append(Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(SH_), S(k) + 1, "0" if k == 1 else vp(SH_))))
        self.emit(interleave([sb, sums]))
        self.raw("s_waitcnt lgkmcnt(0)")
        seq = [Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(SL), S(0), vp(SL))),
               Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(SH_), S(0) + 1, vp(SH_)))]
        self.emit(seq)
        for g in range(0, 12, 3):                 # three words at a time: A = lo * 2^e + SL, B = hi * 2^e + SH, fold
            chains = []
            for i, k in enumerate(range(g, g + 3)):
                A, B, k01, mask = 52 + 4 * i, 54 + 4 * i, 64 + i, S_MASK + 6 + 2 * i
                e = SH[k]
                mult = str(1 << e) if (1 << e) <= 64 else "s%d" % SHIFT_REG[e]
                ch = [Ins("v_mad_u64_u32 %s, vcc, v%d, %s, %s" % (vp(A), S(k), mult, vp(SL))),
                      Ins("v_mad_u64_u32 %s, vcc, v%d, %s, %s" % (vp(B), S(k) + 1, mult, vp(SH_)))]
                if k == 0 or last:                # the next round's constant for word 0 (for every word in the last partial round)
                    ch += [Ins("v_mad_u64_u32 %s, vcc, s%d, 1, %s" % (vp(A), S_RC + 2 * k, vp(A))),
                           Ins("v_mad_u64_u32 %s, vcc, s%d, 1, %s" % (vp(B), S_RC + 2 * k + 1, vp(B)))]
                ch += self.combine(A, B, S(k), k01, mask)
                chains.append(ch)
            self.emit(interleave(chains))

    def add_constants_weak(self):
        """state += 12 constants (classic weak addition; once, between the partial and the last full rounds)."""
        self.load_rc(12)
        self.raw("s_waitcnt lgkmcnt(0)")
        for g in range(0, 12, 3):
            chains = []
            for i, k in enumerate(range(g, g + 3)):
                T, k01, mask = 52 + 4 * i, 64 + i, S_MASK + 2 * i
                chains.append([
                    Ins("v_mad_u64_u32 %s, vcc, v%d, 1, %s" % (vp(T), S(k), sp(S_RC + 2 * k))),      # rc + x.lo: no carry (rc <= p - 1)
                    Ins("v_add_co_u32 v%d, %s, v%d, v%d" % (T + 1, sp(mask), T + 1, S(k) + 1), defs=[mask]),
                    Ins("v_cndmask_b32 v%d, 0, 1, %s" % (k01, sp(mask)), uses=[mask]),
                    Ins("v_mad_u64_u32 %s, vcc, v%d, -1, %s" % (vp(S(k)), k01, vp(T))),
                ])
            self.emit(interleave(chains))

    def permutation(self):
        self.raw("s_mov_b64 %s, %%[rc]" % sp(S_PTR))
        for e, r in SHIFT_REG.items():
            self.raw("s_mov_b32 s%d, 0x%x" % (r, 1 << e))
        self.load_rc(16)                          # round 0's constants, folded into the initial layer
        self.raw("s_waitcnt lgkmcnt(0)")
        self.ext_layer()
        self.raw("s_mov_b32 s%d, 0" % S_PHASE)
        top, loop = self.label("t"), self.label("l")
        self.raw("%s:" % top)
        self.raw("s_mov_b32 s%d, 4" % S_CNT)
        self.raw("%s:" % loop)
        self.full_round()
        self.raw("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
        self.raw("s_cmp_lg_u32 s%d, 0" % S_CNT)
        self.raw("s_cbranch_scc1 %s" % loop)
        end = self.label("e")
        self.raw("s_cmp_lg_u32 s%d, 0" % S_PHASE)
        self.raw("s_cbranch_scc1 %s" % end)
        self.raw("s_mov_b32 s%d, 1" % S_PHASE)
        self.raw("s_mov_b32 s%d, %d" % (S_CNT, 21 if LATE_CONST else 22))
        if ZERO_HOIST:                            # v35 = 0 for every partial-round S-box (nothing in these rounds writes it)
            self.raw("v_mov_b32 v35, 0")
        ploop = self.label("p")
        self.raw("%s:" % ploop)
        self.partial_round()
        self.raw("s_sub_u32 s%d, s%d, 1" % (S_CNT, S_CNT))
        self.raw("s_cmp_lg_u32 s%d, 0" % S_CNT)
        self.raw("s_cbranch_scc1 %s" % ploop)
        if LATE_CONST:
            self.partial_round(last=True)
        else:
            self.add_constants_weak()
        self.raw("s_branch %s" % top)
        self.lines += self.stubs                  # rare paths of the products, out of line
        self.raw("%s:" % end)


M4 = [[5, 7, 1, 3], [4, 6, 1, 1], [1, 3, 5, 7], [1, 1, 4, 6]]


def _solve(M, rhs):
    n = len(M)
    A = [[x % P for x in row] + [rhs[i] % P] for i, row in enumerate(M)]
    for c in range(n):
        piv = next(r for r in range(c, n) if A[r][c])
        A[c], A[piv] = A[piv], A[c]
        inv = pow(A[c][c], P - 2, P)
        A[c] = [x * inv % P for x in A[c]]
        for r in range(n):
            if r != c and A[r][c]:
                f = A[r][c]
                A[r] = [(x - f * y) % P for x, y in zip(A[r], A[c])]
    return [A[i][n] for i in range(n)]


def fold_ext_constants(rc12):
    """M_E x + rc = M_E (x + C) with C = M_E^-1 rc; in the block form of ext_layer, out_b = M4 (x_b + sum_b' x_b' + D_b) with
    D_b = C_b + sum_b' C_b'.  Sixteen table entries: the eight 32-bit halves of D_0 (each as a {half, 0} pair: they are the
    addend of the first multiply-add of the column sums, order (word j, plane)), then D_1 - D_0 and D_2 - D_0 (four words each)."""
    ME = [[M4[i % 4][j % 4] * (2 if i // 4 == j // 4 else 1) for j in range(12)] for i in range(12)]
    C = _solve(ME, rc12)
    Ssum = [(C[j] + C[4 + j] + C[8 + j]) % P for j in range(4)]
    D = [[(C[4 * b + j] + Ssum[j]) % P for j in range(4)] for b in range(3)]
    chk = [(sum(M4[i][j] * (1000 + 4 * b + j + sum(1000 + 4 * bb + j for bb in range(3)) + D[b][j]) for j in range(4))) % P for b in range(3) for i in range(4)]
    ref = [(sum(ME[r][k] * (1000 + k) for k in range(12)) + rc12[r]) % P for r in range(12)]
    assert chk == ref
    out = []
    for j in range(4):
        out += [D[0][j] & 0xFFFFFFFF, D[0][j] >> 32]
    for b in (1, 2):
        out += [(D[b][j] - D[0][j]) % P for j in range(4)]
    return out


def rc_table():
    """The constants in the order the stream consumes them: the folded 16 entries of the initial layer (round 0's constants); after
    full round r the layer folds round r+1's (word 0 only when a partial round follows, none after the last round); after partial
    round r word 0's constant of round r+1 (none when a full round follows); the 12 of the first closing full round in between
    (added weakly, not through a layer)."""
    txt = open(RC_INC).read()
    rc = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{16}", txt)]
    assert len(rc) == 360
    R = lambda r: [x % P for x in rc[12 * r:12 * r + 12]]
    F = fold_ext_constants
    t = F(R(0)) + F(R(1)) + F(R(2)) + F(R(3)) + F([rc[12 * 4] % P] + [0] * 11)
    t += [rc[12 * r] % P for r in range(5, 26)] + ([] if LATE_CONST else [0])
    t += R(26)
    t += F(R(27)) + F(R(28)) + F(R(29)) + F([0] * 12)
    assert len(t) == 16 * 5 + (21 if LATE_CONST else 22) + 12 + 16 * 4
    return t


def generate():
    g = Gen()
    g.permutation()
    body = "\n".join('        "%s\\n\\t"' % l for l in g.lines[:-1]) + '\n        "%s"' % g.lines[-1]
    tab = rc_table()
    rows = ",\n    ".join(", ".join("0x%016xULL" % v for v in tab[i:i + 4]) for i in range(0, len(tab), 4))
    outs = ", ".join('"={v[%d:%d]}"(s[%d])' % (2 * k, 2 * k + 1, k) for k in range(12))
    ins = ", ".join('"{v[%d:%d]}"(s[%d])' % (2 * k, 2 * k + 1, k) for k in range(12))
    clob = ", ".join(['"v%d"' % r for r in range(24, 72)] + ['"s%d"' % r for r in range(36, S_TOP + 1)] + ['"vcc"', '"scc"'])
    n_ins = sum(1 for l in g.lines if not l.endswith(":"))
    src = '''// GENERATED by tools/gen_p2_asm.py — do not edit.  The Poseidon2 permutation as one scheduled gfx950 instruction stream
// (%d instructions, %d of them s_nop; state in v[0:23], temporaries v[24:71], scalars s[36:101]).
#pragma once

namespace bj {
// round constants in the order the stream consumes them (see rc_table in the generator)
__constant__ gl::u64 P2_ASM_RC[%d] = {
    %s};

// state in: any u64 words; state out: weak words (canonicalise what leaves the sponge with gl::canon)
__device__ __forceinline__ void poseidon2_permutation_asm(gl::u64 (&s)[12]) {
    const gl::u64 *rc = P2_ASM_RC;
    asm volatile(
%s
        : %s
        : %s, [rc] "s"(rc)
        : %s);
}
}  // namespace bj
''' % (n_ins, g.nops, len(tab), rows, body, outs, ins, clob)
    return src


def main(out=None):
    out = out or OUT
    src = generate()
    if not os.path.exists(out) or open(out).read() != src:
        with open(out, "w") as f:
            f.write(src)
    return out


if __name__ == "__main__":
    import sys
    print(main(sys.argv[1] if len(sys.argv) > 1 else None))
