"""The hand-scheduled sequences of csrc/gl_asm.inc (tools/gen_gl_asm.py: two lazy NTT butterflies side by side, the same
without the product) executed on the CPU by the single-lane emulator (tools/p2_emulate.py) and compared with integer
arithmetic: whenever a sequence reports "finished" (its rare mask is zero) both results must be congruent to u + v*w and
u - v*w; on the vectors built to hit the borrow-without-carry product and the second wrap of a sum / difference the mask must be
raised (the C++ wrappers then recompute through the canonical operators, gl.h)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_gl_asm as GG   # noqa: E402
import p2_emulate as EM   # noqa: E402

P = (1 << 64) - (1 << 32) + 1
M64 = (1 << 64) - 1


def _words():
    w = [0, 1, 2, 7, 0xFFFFFFFF, 1 << 32, 0xFFFFFFFF00000000, P - 1, P, P + 1, M64, M64 - 1, 1 << 48, 1 << 63, (1 << 63) - 1,
         0xFFFFFFFE00000002, 0x00000000FFFFFFFE]
    rng = np.random.default_rng(17)
    w += [int(x) for x in rng.integers(0, 1 << 64, size=60, dtype=np.uint64)]
    w += [M64 - int(x) for x in rng.integers(0, 1 << 33, size=12, dtype=np.uint64)]
    w += [int(x) for x in rng.integers(0, 1 << 33, size=12, dtype=np.uint64)]
    return w


def _set64(e, regs, name, val):
    lo, hi = regs[name + "0"], regs[name + "1"]
    e.v[int(lo[1:])], e.v[int(hi[1:])] = val & 0xFFFFFFFF, val >> 32


def _get64(e, reg):
    lo = int(reg[2:reg.index(":")])
    return e.v[lo] | (e.v[lo + 1] << 32)


def _mask(e, reg):
    lo = int(reg[2:reg.index(":")])
    return e.s[lo] | (e.s[lo + 1] << 32)


def test_two_butterflies_side_by_side():
    lines, _ = GG.gen_butterfly2()
    ins = ["ua0", "ua1", "va0", "va1", "wa0", "wa1", "ub0", "ub1", "vb0", "vb1", "wb0", "wb1"]
    e, regs = EM.bind_sequence(lines, ins, ["sa", "da", "sb", "db"], ["pa", "qa", "ra", "xa", "ya", "pb", "qb", "rb", "xb", "yb"])
    with open(os.path.join(ROOT, "tests", "golden", "gl_mul_rare.json")) as f:
        rare = [(v["a"], v["b"], v["class"]) for v in json.load(f)["vectors"]]
    w = _words()
    rng = np.random.default_rng(23)

    def run(ua, va, wa, ub, vb, wb):
        for n, val in (("ua", ua), ("va", va), ("wa", wa), ("ub", ub), ("vb", vb), ("wb", wb)):
            _set64(e, regs, n, val)
        e.execute()
        if _mask(e, regs["xa"]) & 1:
            return False                                        # "not finished": the wrapper recomputes canonically
        assert _get64(e, regs["sa"]) % P == (ua + va * wa) % P and _get64(e, regs["da"]) % P == (ua - va * wa) % P
        assert _get64(e, regs["sb"]) % P == (ub + vb * wb) % P and _get64(e, regs["db"]) % P == (ub - vb * wb) % P
        return True

    finished = sum(run(*(w[int(i)] for i in rng.integers(0, len(w), size=6))) for _ in range(3000))
    assert finished > 2900                                      # a second wrap needs both operands within 2^33 of 2^64 (or of 0)
    # the fixture of tools/find_rare_mul_vectors.c: class 1 = the product's last subtraction borrows without a carry (the one
    # case the sequence leaves to the slow path: the mask must say so, in either chain), class 3 = borrow and carry (in line)
    u, v2, w2 = 5, 0x0123456789ABCDEF, 0x0FEDCBA987654321      # a quiet partner chain and a u that wraps nothing
    for a, b, cls in rare:
        for first in (True, False):
            ok = run(u, a, b, u, v2, w2) if first else run(u, v2, w2, u, a, b)
            assert ok == (cls != 1), (a, b, cls, first)


def test_two_add_sub_pairs_side_by_side():
    lines, _ = GG.gen_addsub2()
    ins = ["ua0", "ua1", "va0", "va1", "ub0", "ub1", "vb0", "vb1"]
    e, regs = EM.bind_sequence(lines, ins, ["sa", "da", "sb", "db"], ["pa", "qa", "ra", "ya", "pb", "qb", "rb", "yb"])
    w = _words()
    finished = raised = 0
    for i, ua in enumerate(w):
        for j in range(0, len(w), 3):
            va, ub, vb = w[j], w[(i + j) % len(w)], w[(7 * i + j + 1) % len(w)]
            for n, val in (("ua", ua), ("va", va), ("ub", ub), ("vb", vb)):
                _set64(e, regs, n, val)
            e.execute()
            if _mask(e, regs["ra"]) & 1:
                raised += 1
                continue
            finished += 1
            assert _get64(e, regs["sa"]) % P == (ua + va) % P and _get64(e, regs["da"]) % P == (ua - va) % P
            assert _get64(e, regs["sb"]) % P == (ub + vb) % P and _get64(e, regs["db"]) % P == (ub - vb) % P
    assert finished > 1000
    assert raised > 0                                           # words within 2^33 of 2^64 / of 0 wrap twice: the mask says so


def test_the_inline_product_of_gl_h():
    """gl::mul_weak (csrc/gl.h, hand-written inline assembly, the product every kernel but the generated streams uses): complete
    by itself — the borrow-without-carry case is finished behind its wave-uniform branch — so the result must be congruent to
    a * b for ANY operands, the rare-product fixture included."""
    src = open(os.path.join(ROOT, "era_boojum_amd", "csrc", "gl.h")).read()
    lines = EM.asm_lines_of(src, "__device__ __forceinline__ u64 mul_weak(u64 a, u64 b)")
    assert len(lines) == 23 and lines[0].startswith("v_mad_u64_u32") and lines[-1].startswith("v_mad_u64_u32 %[out]")
    e, regs = EM.bind_sequence(lines, ["a0", "a1", "b0", "b1"], ["out"], ["cm", "c"])
    with open(os.path.join(ROOT, "tests", "golden", "gl_mul_rare.json")) as f:
        rare = [(v["a"], v["b"]) for v in json.load(f)["vectors"]]
    w = _words()
    pairs = [(a, b) for a in w for b in w[::2]] + rare + [(b, a) for a, b in rare]
    slow = 0
    for a, b in pairs:
        for n, val in (("a", a), ("b", b)):
            lo, hi = regs[n + "0"], regs[n + "1"]
            e.v[int(lo[1:])], e.v[int(hi[1:])] = val & 0xFFFFFFFF, val >> 32
        e.execute()
        slow += e.counts["VALU"] > 15                           # the three extra instructions of the rare path
        assert _get64(e, regs["out"]) % P == a * b % P, (a, b)
    assert slow >= 6                                            # the class-1 vectors, both operand orders


def test_no_scheduled_sequence_reads_a_mask_too_early():
    """tools/hazard_lint.py: two wait states between a VALU write of a scalar pair / vcc and a VALU read of it, in the Poseidon2
    stream under every generator option, the butterfly sequences and gl::mul_weak; and the check does flag a sequence that has
    lost one of its s_nop."""
    import hazard_lint as HL
    seqs = HL.all_sequences()
    assert len(seqs) >= 9
    for name, (lines, scal) in seqs.items():
        assert HL.lint(lines, scal) == [], name
    lines, scal = seqs["gl::mul_weak"]
    broken = [l for l in lines if l != "s_nop 0"]
    assert len(broken) == len(lines) - 1
    assert HL.lint(broken, scal)
    lines, scal = seqs["poseidon2 default"]
    i = next(k for k, l in enumerate(lines) if l.startswith("s_nop"))
    assert HL.lint(lines[:i] + lines[i + 1:], scal)
