"""Vectorised Goldilocks arithmetic on numpy uint64 arrays (host-side input synthesis only — never on the proving path).

Used by era_boojum_amd.synthetic to build satisfiable circuits for benches and tests; kept independent of oracle/."""
import numpy as np

P = (1 << 64) - (1 << 32) + 1
_P = np.uint64(P)
_EPS = np.uint64(0xFFFFFFFF)
_M32 = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def _load_native():
    """libsynth_host.so (era_boojum_amd/host/synth_host.c): same arithmetic in C + OpenMP; optional."""
    import ctypes
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsynth_host.so")
    if os.environ.get("BJ_SYNTH_NUMPY") or not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
    except OSError:
        return None
    vp, sz, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64
    lib.synth_mul.argtypes = [vp, vp, vp, sz]
    lib.synth_mul_scalar.argtypes = [vp, u64, vp, sz]
    lib.synth_fma2.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.synth_powers.argtypes = [u64, vp, sz]
    lib.synth_sigma_from_placement.argtypes = [vp, sz, sz, sz, vp]
    lib.synth_sha256_states.argtypes = [vp, sz, vp, vp]
    lib.synth_gather_values.argtypes = [vp, vp, vp, sz]
    for f in (lib.synth_mul, lib.synth_mul_scalar, lib.synth_fma2, lib.synth_powers, lib.synth_sigma_from_placement,
              lib.synth_sha256_states, lib.synth_gather_values):
        f.restype = None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    lib.synth_set_threads(max(1, min(16, cores)))
    return lib


_NATIVE = _load_native()


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def canon(a):
    a = np.asarray(a, dtype=np.uint64)
    return np.where(a >= _P, a - _P, a)


def add(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s = a + b
        t = s + _EPS
    return np.where((s < a) | (t < s), t, s)


def sub(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    with np.errstate(over="ignore"):
        d = a - b
        return np.where(a < b, d - _EPS, d)


def mul(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    if _NATIVE is not None and a.ndim == 1 and a.size >= 1024:
        if b.shape == a.shape:
            a, b = _c(a), _c(b)
            out = np.empty_like(a)
            _NATIVE.synth_mul(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
            return out
        if b.ndim == 0:
            a = _c(a)
            out = np.empty_like(a)
            _NATIVE.synth_mul_scalar(a.ctypes.data, int(b), out.ctypes.data, a.size)
            return out
    with np.errstate(over="ignore"):
        a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
        p00 = a0 * b0
        mid = a0 * b1 + (p00 >> _S32)
        mid2 = a1 * b0 + (mid & _M32)
        lo = (mid2 << _S32) | (p00 & _M32)
        hi = a1 * b1 + (mid >> _S32) + (mid2 >> _S32)
        hi_hi, hi_lo = hi >> _S32, hi & _M32
        t0 = lo - hi_hi
        t0 = np.where(lo < hi_hi, t0 - _EPS, t0)
        t1 = hi_lo * _EPS
        r = t0 + t1
        r = np.where(r < t1, r + _EPS, r)
    return np.where(r >= _P, r - _P, r)


def powers(base, count):
    """[1, base, base^2, ...] (count entries) by repeated doubling of the prefix."""
    out = np.ones(count, dtype=np.uint64)
    if count <= 1:
        return out
    base = int(base) % P
    if _NATIVE is not None:
        _NATIVE.synth_powers(base, out.ctypes.data, count)
        return out
    filled = 1
    step = base
    while filled < count:
        take = min(filled, count - filled)
        out[filled:filled + take] = mul(out[:take], np.uint64(step))
        filled += take
        step = step * step % P
    return out


def omega(log_n):
    w = 0x185629DCDA58878C
    for _ in range(32 - log_n):
        w = w * w % P
    return w


def fma2(a, b, c, d):
    """a*b + c*d"""
    if _NATIVE is not None and all(np.ndim(x) == 1 for x in (a, b, c, d)) and len(a) >= 1024:
        a, b, c, d = _c(a), _c(b), _c(c), _c(d)
        out = np.empty_like(a)
        _NATIVE.synth_fma2(a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data, out.ctypes.data, a.size)
        return out
    return add(mul(a, b), mul(c, d))
