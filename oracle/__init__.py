"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy binding of ``oracle/liboracle.so`` (the C restatement of the reference's CPU algorithms, see
``oracle/oracle.h``).  May be imported only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` — never by ``era_boojum_amd`` (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
P = (1 << 64) - (1 << 32) + 1
GEN = 7


def build(force=False):
    """Compile liboracle.so with gcc (idempotent)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_merkle_tree_digests.restype = C.c_size_t
        _lib.orc_merkle_proof.restype = C.c_size_t
        _lib.orc_transcript_new.restype = C.c_void_p
        _lib.orc_transcript_challenge.restype = C.c_uint64
        _lib.orc_bools_new.restype = C.c_void_p
        _lib.orc_query_index.restype = C.c_uint64
        _lib.orc_fri_schedule.restype = C.c_size_t
        _lib.orc_do_fri.restype = C.c_void_p
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


# ---------------- field helpers (python ints) ----------------
def inv(a):
    return pow(a % P, P - 2, P)


def omega(log_n):
    w = 0x185629DCDA58878C
    for _ in range(32 - log_n):
        w = w * w % P
    return w


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


# ---------------- NTT ----------------
def twiddles(log_n, inverse=False):
    out = np.zeros(max(1, (1 << log_n) // 2), dtype=np.uint64)
    lib().orc_twiddles(_p(out), C.c_uint(log_n), C.c_int(int(inverse)))
    return out


def fft_natural_to_bitreversed(a, coset=1, tw=None):
    a = _arr(a).copy()
    log_n = int(a.shape[-1]).bit_length() - 1
    tw = twiddles(log_n) if tw is None else tw
    lib().orc_fft_natural_to_bitreversed(_p(a), C.c_uint(log_n), C.c_uint64(coset), _p(tw))
    return a


def ifft_natural_to_natural(a, coset=1, tw=None):
    a = _arr(a).copy()
    log_n = int(a.shape[-1]).bit_length() - 1
    tw = twiddles(log_n, True) if tw is None else tw
    lib().orc_ifft_natural_to_natural(_p(a), C.c_uint(log_n), C.c_uint64(coset), _p(tw))
    return a


def naive_dft(a, coset=1):
    a = _arr(a)
    out = np.zeros_like(a)
    lib().orc_naive_dft(_p(a), _p(out), C.c_uint(int(a.size).bit_length() - 1), C.c_uint64(coset))
    return out


def bitreverse(a):
    a = _arr(a).copy()
    lib().orc_bitreverse(_p(a), C.c_uint(int(a.size).bit_length() - 1))
    return a


def canonical(a):
    a = _arr(a).copy()
    lib().orc_canonicalize(_p(a.reshape(-1)), C.c_size_t(a.size))
    return a


def lde_coset_shifts(log_n, log_lde):
    out = np.zeros(1 << log_lde, dtype=np.uint64)
    lib().orc_lde_coset_shifts(_p(out), C.c_uint(log_n), C.c_uint(log_lde))
    return out


def fft_batch(cols, coset=1, threads=1, out=None):
    """cols: [n_cols, n] natural -> bit-reversed evaluations on coset*<w>; returns a new array, or `out` (same shape, reused by a
    caller that transforms the same columns to many cosets: no fresh pages per call)."""
    if out is not None:
        src = _arr(cols)
        assert out.shape == src.shape and out.dtype == np.uint64 and out.flags["C_CONTIGUOUS"] and out is not src
        n_cols, n = src.shape
        lib().orc_fft_batch_to(_p(src), _p(out), C.c_uint(n.bit_length() - 1), C.c_size_t(n_cols), C.c_uint64(coset), C.c_int(threads))
        return out
    a = _arr(cols).copy()
    n_cols, n = a.shape
    lib().orc_fft_batch(_p(a), C.c_uint(n.bit_length() - 1), C.c_size_t(n_cols), C.c_uint64(coset), C.c_int(threads))
    return a


def ifft_batch(cols, coset=1, threads=1):
    a = _arr(cols).copy()
    n_cols, n = a.shape
    lib().orc_ifft_batch(_p(a), C.c_uint(n.bit_length() - 1), C.c_size_t(n_cols), C.c_uint64(coset), C.c_int(threads))
    return a


def lde_batch(mono, log_lde, threads=1):
    """mono: [n_cols, n] monomials -> [n_cols, L, n] (each coset bit-reversed)."""
    m = _arr(mono)
    n_cols, n = m.shape
    out = np.zeros((n_cols, 1 << log_lde, n), dtype=np.uint64)
    lib().orc_lde_batch(_p(m), _p(out), C.c_uint(n.bit_length() - 1), C.c_uint(log_lde), C.c_size_t(n_cols),
                        C.c_int(threads))
    return out


# ---------------- Poseidon2 / Merkle ----------------
def poseidon2_permutation(state):
    s = _arr(state).copy()
    assert s.size == 12
    lib().orc_poseidon2_permutation(_p(s))
    return s


def poseidon2_avx512_available():
    """True when trees are hashed by oracle/poseidon2_avx512.c (the CPU has AVX-512 F + DQ and ORC_NO_AVX512 is unset; the choice
    is made once per process)."""
    return bool(lib().orc_poseidon2_avx512_available())


def poseidon2_isa():
    return "avx512 (8 permutations per call, lane = leaf)" if poseidon2_avx512_available() else "scalar (u128 accumulators)"


def poseidon2_permutation_x8(states):
    """Eight states (8 x 12) through the AVX-512 permutation; raises where the CPU has no AVX-512."""
    if not poseidon2_avx512_available():
        raise RuntimeError("no AVX-512 on this CPU (or ORC_NO_AVX512 is set)")
    s = np.ascontiguousarray(_arr(states).reshape(8, 12)).copy()
    lib().orc_poseidon2_permutation_x8(_p(s))
    return s


def poseidon_permutation(state):
    """Poseidon (v1, naive) permutation — bench-script transcript only."""
    s = _arr(state).copy()
    lib().orc_poseidon_permutation(_p(s))
    return s


def hash_leaf(els):
    e = _arr(els)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_hash_leaf(_p(e), C.c_size_t(e.size), _p(out))
    return out


def hash_node(l, r):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_hash_node(_p(_arr(l)), _p(_arr(r)), _p(out))
    return out


def merkle_construct(cols, cap_size, threads=1):
    """cols: [n_cols, num_leaves]; returns all layers flat [(2*num_leaves-cap), 4]."""
    a = _arr(cols)
    n_cols, num_leaves = a.shape
    tree = np.zeros((2 * num_leaves - cap_size, 4), dtype=np.uint64)
    lib().orc_merkle_construct_strided(_p(a), C.c_size_t(num_leaves), C.c_size_t(n_cols), C.c_size_t(num_leaves),
                                       C.c_size_t(cap_size), _p(tree), C.c_int(threads))
    return tree


def merkle_construct_chunked(srcs, elems_per_leaf, cap_size, threads=1):
    """srcs: [n_srcs, len]; leaf j = srcs[0][jE:(j+1)E] || srcs[1][jE:(j+1)E] ..."""
    a = _arr(srcs)
    n_srcs, ln = a.shape
    num_leaves = ln // elems_per_leaf
    tree = np.zeros((2 * num_leaves - cap_size, 4), dtype=np.uint64)
    ptrs = (u64p * n_srcs)(*[_p(a[i]) for i in range(n_srcs)])
    lib().orc_merkle_construct_chunked(ptrs, C.c_size_t(n_srcs), C.c_size_t(ln), C.c_size_t(elems_per_leaf),
                                       C.c_size_t(cap_size), _p(tree), C.c_int(threads))
    return tree


def merkle_nodes_from_leaf_hashes(leaf_hashes, cap_size, threads=1):
    lh = _arr(leaf_hashes)
    num_leaves = lh.shape[0]
    tree = np.zeros((2 * num_leaves - cap_size, 4), dtype=np.uint64)
    tree[:num_leaves] = lh
    lib().orc_merkle_nodes(_p(tree), C.c_size_t(num_leaves), C.c_size_t(cap_size), C.c_int(threads))
    return tree


def merkle_cap(tree, num_leaves, cap_size):
    return tree[2 * num_leaves - 2 * cap_size:2 * num_leaves - cap_size].copy()


def merkle_proof(tree, num_leaves, cap_size, idx):
    depth = (num_leaves // cap_size).bit_length() - 1
    leaf = np.zeros(4, dtype=np.uint64)
    path = np.zeros((max(depth, 1), 4), dtype=np.uint64)
    d = lib().orc_merkle_proof(_p(tree), C.c_size_t(num_leaves), C.c_size_t(cap_size), C.c_size_t(idx), _p(leaf), _p(path))
    assert d == depth
    return leaf, path[:depth]


def merkle_verify(path, cap, leaf_hash, idx):
    path = _arr(path).reshape(-1, 4) if len(path) else np.zeros((0, 4), dtype=np.uint64)
    pbuf = path if path.size else np.zeros((1, 4), dtype=np.uint64)
    return bool(lib().orc_merkle_verify(_p(pbuf), C.c_size_t(path.shape[0]), _p(_arr(cap)), _p(_arr(leaf_hash)),
                                        C.c_size_t(idx)))


# ---------------- transcript ----------------
TRANSCRIPT_POSEIDON2, TRANSCRIPT_POSEIDON = 1, 2


class Transcript:
    """Algebraic sponge transcript (transcript.rs:48-131) over Poseidon2 (:144-151, the golden proof's) or, kind=2, over
    the Poseidon (v1) permutation of the SHA-256 bench script (:133-141; no KAT exists for that permutation)."""

    def __init__(self, kind=TRANSCRIPT_POSEIDON2):
        lib().orc_transcript_new_kind.restype = C.c_void_p
        self.kind = kind
        self._h = C.c_void_p(lib().orc_transcript_new_kind(int(kind)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_transcript_free(self._h)
            self._h = None

    def absorb(self, els):
        e = _arr(els).reshape(-1)
        if e.size:
            lib().orc_transcript_absorb(self._h, _p(e), C.c_size_t(e.size))

    def absorb_cap(self, cap):
        self.absorb(_arr(cap).reshape(-1))

    def challenge(self):
        return int(lib().orc_transcript_challenge(self._h))

    def challenge_ext(self):
        return (self.challenge(), self.challenge())


class QueryIndexer:
    def __init__(self, log_n, log_lde):
        self.log_n, self.log_lde = log_n, log_lde
        self._h = C.c_void_p(lib().orc_bools_new(C.c_uint(log_n + log_lde)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_bools_free(self._h)
            self._h = None

    def next(self, transcript):
        return int(lib().orc_query_index(self._h, transcript._h, C.c_uint(self.log_n), C.c_uint(self.log_lde)))


# ---------------- FRI ----------------
def fri_schedule(security_bits, cap_size, pow_bits, rate_log2, initial_degree_log2):
    sched = (C.c_uint32 * 32)()
    new_pow = C.c_uint32()
    nq = C.c_size_t()
    fd = C.c_size_t()
    ln = lib().orc_fri_schedule(C.c_uint32(security_bits), C.c_size_t(cap_size), C.c_uint32(pow_bits),
                                C.c_uint32(rate_log2), C.c_uint32(initial_degree_log2), C.byref(new_pow),
                                C.byref(nq), sched, C.byref(fd))
    return new_pow.value, nq.value, [int(sched[i]) for i in range(ln)], fd.value


def fri_fold(c0, c1, roots, coset_inv, ch):
    c0, c1, roots = _arr(c0), _arr(c1), _arr(roots)
    o0 = np.zeros(c0.size // 2, dtype=np.uint64)
    o1 = np.zeros_like(o0)
    lib().orc_fri_fold(_p(c0), _p(c1), C.c_size_t(c0.size), _p(o0), _p(o1), _p(roots), C.c_uint64(coset_inv),
                       C.c_uint64(ch[0]), C.c_uint64(ch[1]))
    return o0, o1


class _FriResult(C.Structure):
    _fields_ = [("num_oracles", C.c_size_t), ("trees", u64p * 32), ("tree_leaves", C.c_size_t * 32),
                ("elems_per_leaf", C.c_size_t * 32), ("src_c0", u64p * 32), ("src_c1", u64p * 32),
                ("src_len", C.c_size_t * 32), ("final_c0", u64p), ("final_c1", u64p), ("final_degree", C.c_size_t),
                ("challenges", (C.c_uint64 * 2) * 32)]


def do_fri(c0, c1, log_lde, schedule, cap_size, transcript, threads=1):
    """Runs the reference's do_fri; returns dict with per-oracle trees/caps/sources and the final monomials."""
    c0, c1 = _arr(c0), _arr(c1)
    log_full = int(c0.size).bit_length() - 1
    sched = (C.c_uint32 * len(schedule))(*schedule)
    h = lib().orc_do_fri(_p(c0), _p(c1), C.c_uint(log_full), C.c_uint(log_lde), sched, C.c_size_t(len(schedule)),
                         C.c_size_t(cap_size), transcript._h, C.c_int(threads))
    r = C.cast(h, C.POINTER(_FriResult)).contents
    out = {"trees": [], "caps": [], "sources": [], "challenges": []}
    for i in range(r.num_oracles):
        leaves = r.tree_leaves[i]
        nd = 2 * leaves - cap_size
        tree = np.ctypeslib.as_array(r.trees[i], shape=(nd, 4)).copy()
        out["trees"].append(tree)
        out["caps"].append(merkle_cap(tree, leaves, cap_size))
        ln = r.src_len[i]
        out["sources"].append((np.ctypeslib.as_array(r.src_c0[i], shape=(ln,)).copy(),
                               np.ctypeslib.as_array(r.src_c1[i], shape=(ln,)).copy()))
        out["challenges"].append((int(r.challenges[i][0]), int(r.challenges[i][1])))
    ln = r.src_len[r.num_oracles]
    out["last_folded"] = (np.ctypeslib.as_array(r.src_c0[r.num_oracles], shape=(ln,)).copy(),
                          np.ctypeslib.as_array(r.src_c1[r.num_oracles], shape=(ln,)).copy())
    fd = r.final_degree
    out["final_monomials"] = (np.ctypeslib.as_array(r.final_c0, shape=(ln,)).copy(),
                              np.ctypeslib.as_array(r.final_c1, shape=(ln,)).copy())
    out["final_degree"] = fd
    lib().orc_fri_result_free(C.c_void_p(h))
    return out


def batch_inverse(a):
    a = _arr(a)
    out = np.zeros_like(a)
    lib().orc_batch_inverse(_p(a), _p(out), C.c_size_t(a.size))
    return out


# ---------------- openings: barycentric evaluation + DEEP ----------------
def barycentric_weights(log_n, coset, at):
    n = 1 << log_n
    w0, w1 = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    lib().orc_barycentric_weights(C.c_uint(log_n), C.c_uint64(coset), _p(_arr(at)), _p(w0), _p(w1))
    return w0, w1


def barycentric_eval_base(values, w0, w1):
    v = _arr(values)
    out = np.zeros(2, dtype=np.uint64)
    lib().orc_barycentric_eval_base(_p(v), _p(w0), _p(w1), C.c_size_t(v.size), _p(out))
    return (int(out[0]), int(out[1]))


def barycentric_eval_ext(v0, v1, w0, w1):
    v0, v1 = _arr(v0), _arr(v1)
    out = np.zeros(2, dtype=np.uint64)
    lib().orc_barycentric_eval_ext(_p(v0), _p(v1), _p(w0), _p(w1), C.c_size_t(v0.size), _p(out))
    return (int(out[0]), int(out[1]))


def deep_quotient_accumulate(sources, values, challenges, at, log_n, log_lde, dst0, dst1, threads=1):
    """sources: list of (c0_array, c1_array_or_None) each of length 2^(log_n+log_lde); dst updated in place."""
    k = len(sources)
    keep = []
    p0 = (u64p * k)()
    p1 = (u64p * k)()
    for i, (a, b) in enumerate(sources):
        a = _arr(a); keep.append(a); p0[i] = _p(a)
        if b is None:
            p1[i] = None
        else:
            b = _arr(b); keep.append(b); p1[i] = _p(b)
    vals = _arr(values).reshape(-1)
    chs = _arr(challenges).reshape(-1)
    lib().orc_deep_quotient_accumulate(p0, p1, C.c_size_t(k), _p(vals), _p(chs), _p(_arr(at)), C.c_uint(log_n),
                                       C.c_uint(log_lde), _p(dst0), _p(dst1), C.c_int(threads))


def deep_quotient_accumulate_range(sources, values, challenges, at, log_n, log_lde, first, dst0, dst1, threads=1):
    """The same on the flat LDE indices [first, first + len(dst0)): sources and dst hold only those points."""
    k = len(sources)
    keep = []
    p0 = (u64p * k)()
    p1 = (u64p * k)()
    count = dst0.size
    for i, (a, b) in enumerate(sources):
        a = _arr(a); keep.append(a); p0[i] = _p(a)
        assert a.size == count
        if b is None:
            p1[i] = None
        else:
            b = _arr(b); keep.append(b); p1[i] = _p(b)
    vals = _arr(values).reshape(-1)
    chs = _arr(challenges).reshape(-1)
    assert dst0.flags["C_CONTIGUOUS"] and dst1.flags["C_CONTIGUOUS"] and dst1.size == count
    lib().orc_deep_quotient_accumulate_range(p0, p1, C.c_size_t(k), _p(vals), _p(chs), _p(_arr(at)), C.c_uint(log_n),
                                             C.c_uint(log_lde), C.c_size_t(first), C.c_size_t(count), _p(dst0), _p(dst1),
                                             C.c_int(threads))


def deep_quotient_point(f, values, challenges, at, x):
    """f: list of (c0, c1_or_None) python ints at ONE LDE point x; returns the (c0, c1) contribution."""
    k = len(f)
    f0 = _arr([a for a, _ in f])
    f1 = _arr([0 if b is None else b for _, b in f])
    ie = np.array([0 if b is None else 1 for _, b in f], dtype=np.uint8)
    out = np.zeros(2, dtype=np.uint64)
    lib().orc_deep_quotient_point(_p(f0), _p(f1), ie.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_size_t(k),
                                  _p(_arr(values).reshape(-1)), _p(_arr(challenges).reshape(-1)), _p(_arr(at)),
                                  C.c_uint64(x), _p(out))
    return (int(out[0]), int(out[1]))


def poseidon_round_constants():
    """The 30 x 12 round-constant table of oracle/poseidon_rc.h (ALL_ROUND_CONSTANTS, shared by Poseidon and Poseidon2)."""
    import re
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "poseidon_rc.h")).read()
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{16})ULL", txt)]
    assert len(vals) == 360
    return [vals[12 * r:12 * r + 12] for r in range(30)]
