# brute-force LDS bank-conflict check for the front10 tile (1024 mid x 8 lo, 512 threads, 16 regs)
import itertools
def pad(x): return x + (x >> 4)
def wr_conf(addrs):   # ds_write_b64: 4 groups of 16 contiguous lanes, 16 8-byte slots
    worst = 1
    for g in range(0, 64, 16):
        slots = {}
        for a in addrs[g:g+16]:
            slots.setdefault(a % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst
def rd_conf(addrs):   # ds_read_b64: 2 groups of 32 lanes, 32 8-byte slots
    worst = 1
    for g in range(0, 64, 32):
        slots = {}
        for a in addrs[g:g+32]:
            slots.setdefault(a % 32, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst
def check(name, E, layA, layB, layC):
    # layX(t, i) -> (m, l)
    res = {}
    for nm, lay, fn in (("A.wr", layA, wr_conf), ("B.rd", layB, rd_conf), ("B.wr", layB, wr_conf), ("C.rd", layC, rd_conf)):
        worst = 0
        for w in range(8):
            for i in range(16):
                addrs = [E(*lay(w * 64 + lane, i)) for lane in range(64)]
                worst = max(worst, fn(addrs))
        res[nm] = worst
    # bijection
    s = set(E(m, l) for m in range(1024) for l in range(8))
    assert len(s) == 8192
    print(name, res, "max idx", max(s))
def layA(t, i):
    l, ml = t & 7, t >> 3
    return (i * 64 + ml, l)
def mk_layB(order):
    # order: function t -> (mh, mll, l)
    def f(t, i):
        mh, mll, l = order(t)
        return (mh * 64 + i * 4 + mll, l)
    return f
def layC(t, i):
    l2, m92 = t & 1, t >> 1
    mm, ll = i >> 2, i & 3
    return (m92 * 4 + mm, l2 * 4 + ll)
ordB1 = lambda t: (t >> 5, (t >> 3) & 3, t & 7)
ordB2 = lambda t: (t & 15, (t >> 4) & 3, t >> 6)
ordB3 = lambda t: (t >> 5, t & 3, (t >> 2) & 7)
for nameE, E in (("pad(m*8+l)", lambda m, l: pad(m * 8 + l)),
                 ("pad(l*1024+m)", lambda m, l: pad(l * 1024 + m)),
                 ("l*1024 + m^(m>>5)", lambda m, l: l * 1024 + ((m & ~31) | ((m ^ (m >> 5)) & 31))),
                 ):
    for nb, ob in (("B1", ordB1), ("B2", ordB2), ("B3", ordB3)):
        check(nameE + " " + nb, E, layA, mk_layB(ob), layC)
print("---- search")
def worst(E, lay, fn):
    w_ = 0
    for w in range(8):
        for i in range(16):
            addrs = [E(*lay(w * 64 + lane, i)) for lane in range(64)]
            w_ = max(w_, fn(addrs))
    return w_
import random
def mkE(a, b, c, base):
    def E(m, l):
        x = (m * 8 + l) if base == 0 else (l * 1024 + m)
        return x ^ (((x >> a) ^ (x >> b if b else 0) ^ (x >> c if c else 0)) & 31)
    return E
best = {}
for base in (0, 1):
    for a in range(5, 13):
        for b in [0] + list(range(a + 1, 13)):
            for c in [0] + (list(range(b + 1, 13)) if b else []):
                E = mkE(a, b, c, base)
                if len(set(E(m, l) for m in range(1024) for l in range(8))) != 8192: continue
                for nb, ob in (("B1", ordB1), ("B2", ordB2), ("B3", ordB3)):
                    lb = mk_layB(ob)
                    ab = (worst(E, layA, wr_conf), worst(E, lb, rd_conf))
                    bc = (worst(E, lb, wr_conf), worst(E, layC, rd_conf))
                    k1 = ("AB", nb)
                    if k1 not in best or sum(ab) < sum(best[k1][0]): best[k1] = (ab, (base, a, b, c))
                    k2 = ("BC", nb)
                    if k2 not in best or sum(bc) < sum(best[k2][0]): best[k2] = (bc, (base, a, b, c))
for k, v in sorted(best.items()): print(k, v)
print("---- linear + bc")
def Elin(m, l): return m * 8 + l
def Ebc(m, l):
    x = m * 8 + l
    return x ^ (((x >> 5) ^ (x >> 10)) & 31)
lb = mk_layB(ordB1)
print("A.wr lin", worst(Elin, layA, wr_conf), "A.rd lin", worst(Elin, layA, rd_conf), "B.rd lin", worst(Elin, lb, rd_conf), "B.wr bc", worst(Ebc, lb, wr_conf), "C.rd bc", worst(Ebc, layC, rd_conf),
      len(set(Ebc(m, l) for m in range(1024) for l in range(8))))
for a_ in range(5, 13):
    for b_ in [0] + list(range(a_ + 1, 13)):
        for c_ in [0] + (list(range(b_ + 1, 13)) if b_ else []):
            def E(m, l, a_=a_, b_=b_, c_=c_):
                x = m * 8 + l
                return x ^ (((x >> a_) ^ (x >> b_ if b_ else 0) ^ (x >> c_ if c_ else 0)) & 31)
            if len(set(E(m, l) for m in range(1024) for l in range(8))) != 8192: continue
            r = (worst(E, lb, wr_conf), worst(E, layC, rd_conf))
            if r == (1, 1): print("ok", a_, b_, c_)
# tile 1024 x 16, 1024 threads; wave-local region = mid bits 9..6 = wave: local index = mloc(6 bits: m[5:0]) * 16 + l(4 bits)
def wr_conf(addrs):
    worst = 1
    for g in range(0, 64, 16):
        slots = {}
        for a in addrs[g:g+16]: slots.setdefault(a % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst
def rd_conf(addrs):
    worst = 1
    for g in range(0, 64, 32):
        slots = {}
        for a in addrs[g:g+32]: slots.setdefault(a % 32, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst
def layB(lane, i):   # lane = (mll : bits 5..4, l : bits 3..0); reg i = m[5:2]
    l, mll = lane & 15, lane >> 4
    return (i * 4 + mll, l)
def layC(lane, i):   # lane = (m[5:2] : bits 5..2, l32 : bits 1..0); reg = (m[1:0], l[1:0])
    l32, m52 = lane & 3, lane >> 2
    return (m52 * 4 + (i >> 2), l32 * 4 + (i & 3))
def worst(E, lay, fn): return max(fn([E(*lay(lane, i)) for lane in range(64)]) for i in range(16))
found = []
for a in range(1, 10):
    for b in [0] + list(range(a + 1, 10)):
        for c in [0] + (list(range(b + 1, 10)) if b else []):
            for mask in (31, 15):
                def E(m, l, a=a, b=b, c=c, mask=mask):
                    x = m * 16 + l
                    return x ^ (((x >> a) ^ (x >> b if b else 0) ^ (x >> c if c else 0)) & mask)
                if len(set(E(m, l) for m in range(64) for l in range(16))) != 1024: continue
                r = (worst(E, layB, wr_conf), worst(E, layC, rd_conf))
                if r == (1, 1): found.append((a, b, c, mask))
print(found[:12])
print("B.rd linear", worst(lambda m, l: m * 16 + l, layB, rd_conf), "C.rd linear", worst(lambda m, l: m * 16 + l, layC, rd_conf), "B.wr lin", worst(lambda m, l: m * 16 + l, layB, wr_conf))
