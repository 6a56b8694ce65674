#!/usr/bin/env python3
"""Static check of the hand-scheduled gfx950 sequences for the one hazard the hardware does not interlock here: a VALU
instruction that READS an SGPR pair or vcc (carry-in of v_addc / v_subb / v_subbrev, the mask of v_cndmask, a scalar source
operand) needs two wait states after the VALU instruction that WROTE it (carry-out of v_mad_u64_u32 / v_add_co / v_sub_co ...).
Every instruction issued in between counts one wait state, `s_nop k` counts k + 1.  SALU reads of such a register are
interlocked (the sequences rely on that: s_andn2 / s_cmp right behind the write).  Control flow: a label forgets the history
(every out-of-line path is entered through s_cmp + s_cbranch — two issued instructions — and left through s_branch to a label
whose next VALU instruction reads no mask), which keeps the check linear.
    python tools/hazard_lint.py        # the Poseidon2 stream (every generator option), the butterfly sequences, gl::mul_weak"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CARRY_OUT = {"v_mad_u64_u32": 1, "v_add_co_u32": 1, "v_sub_co_u32": 1, "v_addc_co_u32": 1, "v_subb_co_u32": 1, "v_subbrev_co_u32": 1}
MASK_IN = {"v_addc_co_u32": 4, "v_subb_co_u32": 4, "v_subbrev_co_u32": 4, "v_cndmask_b32": 3}


def _split(text):
    out, depth, cur = [], 0, ""
    for ch in text:
        depth += ch == "["
        depth -= ch == "]"
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _scalar_regs(op):
    """The scalar registers an operand names: 'vcc', 's12', 's[12:13]', '%[name]' (an asm operand bound to a scalar pair)."""
    if op == "vcc":
        return {"vcc"}
    m = re.fullmatch(r"s(\d+)", op)
    if m:
        return {"s%d" % int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", op)
    if m:
        return {"s%d" % r for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    m = re.fullmatch(r"%\[(\w+)\]", op)
    if m:
        return {"%" + m.group(1)}
    return set()


def lint(lines, scalar_operands=()):
    """lines: instruction text; scalar_operands: names of %[...] operands that are scalar pairs.  Returns a list of violations."""
    scalars = {"%" + n for n in scalar_operands}
    last_write, pos, bad = {}, 0, []
    for raw in lines:
        t = raw.strip()
        if not t:
            continue
        if t.endswith(":"):
            last_write = {}
            continue
        m = re.match(r"(\S+)\s*(.*)", t)
        op, a = m.group(1), _split(m.group(2))
        if op == "s_nop":
            pos += int(a[0], 0) + 1
            continue
        if op.startswith("v_"):
            reads = set()
            srcs = a[2:] if op in CARRY_OUT else a[1:]
            for o in srcs:
                r = _scalar_regs(o)
                reads |= {x for x in r if not x.startswith("%") or x in scalars}
            for r in reads:
                if r in last_write and pos - last_write[r] - 1 < 2:
                    bad.append("%s reads %s %d wait state(s) after its VALU write" % (t, r, pos - last_write[r] - 1))
            if op in CARRY_OUT:
                for r in _scalar_regs(a[CARRY_OUT[op]]):
                    if not r.startswith("%") or r in scalars:
                        last_write[r] = pos
        pos += 1
    return bad


def all_sequences():
    import gen_gl_asm as GG
    import p2_emulate as EM
    out = {}
    for name, env in (("poseidon2 default", None), ("poseidon2 ways2", {"BJ_P2_WAYS": "2"}), ("poseidon2 ways4", {"BJ_P2_WAYS": "4"}),
                      ("poseidon2 combine_inline", {"BJ_P2_COMBINE": "inline"}), ("poseidon2 zero_hoist only", {"BJ_P2_ZERO_HOIST": "1", "BJ_P2_LATE_CONST": "0"}),
                      ("poseidon2 round-3 stream", {"BJ_P2_ZERO_HOIST": "0", "BJ_P2_LATE_CONST": "0"})):
        e = EM.build(env)
        lines = []
        labels = {v: k for k, v in e.labels.items()}
        for i, (op, a) in enumerate(e.prog):
            if i in labels:
                lines.append(labels[i] + ":")
            lines.append(op + " " + ", ".join(a))
        out[name] = (lines, ())
    out["butterfly2"] = (GG.gen_butterfly2()[0], ("pa", "qa", "ra", "xa", "ya", "pb", "qb", "rb", "xb", "yb"))
    out["addsub2"] = (GG.gen_addsub2()[0], ("pa", "qa", "ra", "ya", "pb", "qb", "rb", "yb"))
    src = open(os.path.join(ROOT, "era_boojum_amd", "csrc", "gl.h")).read()
    out["gl::mul_weak"] = (EM.asm_lines_of(src, "__device__ __forceinline__ u64 mul_weak(u64 a, u64 b)"), ("cm", "c"))
    return out


if __name__ == "__main__":
    rc = 0
    for name, (lines, scal) in all_sequences().items():
        bad = lint(lines, scal)
        print("%-34s %5d lines, %d violation(s)" % (name, len(lines), len(bad)))
        for b in bad[:8]:
            print("    " + b)
        rc |= bool(bad)
    sys.exit(rc)
