"""ctypes binding of include/boojum_hip.h.  Device buffers are plain integer addresses (e.g. ``tensor.data_ptr()``)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
P = (1 << 64) - (1 << 32) + 1
u64p = C.POINTER(C.c_uint64)

# every symbol include/boojum_hip.h declares (checked by tests/test_abi_symbols.py against the header text)
_SIGNATURES = {
    "bj_abi_version": (C.c_int, []),
    "bj_device_count": (C.c_int, []),
    "bj_status_string": (C.c_char_p, [C.c_int]),
    "bj_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bj_ctx_destroy": (None, [C.c_void_p]),
    "bj_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bj_last_error": (C.c_char_p, [C.c_void_p]),
    "bj_sync": (C.c_int, [C.c_void_p]),
    "bj_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "bj_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bj_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_timer_start": (C.c_int, [C.c_void_p]),
    "bj_timer_stop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "bj_ntt_forward_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t, C.c_uint64]),
    "bj_intt_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t, C.c_uint64]),
    "bj_lde_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]),
    "bj_trace_to_lde_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]),
    "bj_bitreverse_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t]),
    "bj_canonicalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_ntt_forward_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint64]),
    "bj_intt_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint64]),
    "bj_merkle_tree_digests": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "bj_merkle_tree_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p]),
    "bj_merkle_tree_build_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint, C.c_size_t, C.c_size_t, C.c_void_p]),
    "bj_merkle_tree_build_chunked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_size_t, C.c_void_p]),
    "bj_merkle_tree_nodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "bj_merkle_tree_cap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "bj_merkle_tree_proof": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "bj_poseidon2_permute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_fri_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64]),
}


class BoojumHipError(RuntimeError):
    pass


def lib_path():
    # BOOJUM_HIP_LIB lets tuning experiments point at an alternative build of the SAME library (never a fallback)
    return os.environ.get("BOOJUM_HIP_LIB") or os.path.join(_HERE, "libboojum_hip.so")


def exported_symbols():
    return sorted(_SIGNATURES)


_lib = None


def load_library():
    """dlopen libboojum_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise BoojumHipError("%s is missing: run `python -m era_boojum_amd.build` (hipcc, gfx950). "
                                 "There is no CPU fallback." % path)
        lib = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _np_ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One per GPU: wraps bj_ctx (device, stream, twiddle cache, scratch)."""

    def __init__(self, device=0, stream=None):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.bj_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise BoojumHipError("bj_ctx_create(device=%d) failed: %s" % (device, self._lib.bj_status_string(rc).decode()))
        self._h = h
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bj_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BoojumHipError("%s: %s" % (self._lib.bj_status_string(rc).decode(),
                                             self._lib.bj_last_error(self._h).decode()))

    # -- plumbing
    def set_stream(self, stream_handle):
        self._check(self._lib.bj_ctx_set_stream(self._h, C.c_void_p(stream_handle)))

    def sync(self):
        self._check(self._lib.bj_sync(self._h))

    def malloc(self, nbytes):
        p = C.c_void_p()
        self._check(self._lib.bj_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, dptr):
        self._check(self._lib.bj_free(self._h, C.c_void_p(dptr)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        self._check(self._lib.bj_memcpy_h2d(self._h, C.c_void_p(dptr), _np_ptr(arr), arr.nbytes))

    def d2h(self, dptr, shape):
        out = np.empty(shape, dtype=np.uint64)
        self._check(self._lib.bj_memcpy_d2h(self._h, _np_ptr(out), C.c_void_p(dptr), out.nbytes))
        return out

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        d = self.malloc(arr.nbytes)
        self.h2d(d, arr)
        return d

    def timer_start(self):
        self._check(self._lib.bj_timer_start(self._h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._check(self._lib.bj_timer_stop_ms(self._h, C.byref(ms)))
        return ms.value

    # -- NTT family (device pointers)
    def ntt_forward_batch(self, d_in, d_out, log_n, n_cols, col_stride=None, coset=1):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_ntt_forward_batch(self._h, d_in, d_out, log_n, n_cols, col_stride, coset))

    def intt_batch(self, d_in, d_out, log_n, n_cols, col_stride=None, coset=1):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_intt_batch(self._h, d_in, d_out, log_n, n_cols, col_stride, coset))

    def lde_batch(self, d_mono, d_out, log_n, n_cols, log_lde, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_lde_batch(self._h, d_mono, col_stride, d_out, log_n, n_cols, log_lde))

    def trace_to_lde_batch(self, d_cols, d_out, log_n, n_cols, log_lde, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_trace_to_lde_batch(self._h, d_cols, col_stride, d_out, log_n, n_cols, log_lde))

    def bitreverse_batch(self, d_in, d_out, log_n, n_cols, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_bitreverse_batch(self._h, d_in, d_out, log_n, n_cols, col_stride))

    def canonicalize(self, d, n):
        self._check(self._lib.bj_canonicalize(self._h, d, n))

    # -- host convenience
    def ntt_forward_host(self, arr, coset=1):
        a = np.ascontiguousarray(arr, dtype=np.uint64).copy()
        a2 = a.reshape(-1, a.shape[-1])
        self._check(self._lib.bj_ntt_forward_host(self._h, _np_ptr(a2), a2.shape[1].bit_length() - 1, a2.shape[0], coset))
        return a

    def intt_host(self, arr, coset=1):
        a = np.ascontiguousarray(arr, dtype=np.uint64).copy()
        a2 = a.reshape(-1, a.shape[-1])
        self._check(self._lib.bj_intt_host(self._h, _np_ptr(a2), a2.shape[1].bit_length() - 1, a2.shape[0], coset))
        return a

    # -- Merkle
    def merkle_tree_digests(self, num_leaves, cap_size):
        return self._lib.bj_merkle_tree_digests(num_leaves, cap_size)

    def merkle_tree_build(self, d_cols, col_stride, n_cols, num_leaves, cap_size, d_tree):
        self._check(self._lib.bj_merkle_tree_build(self._h, d_cols, col_stride, n_cols, num_leaves, cap_size, d_tree))

    def merkle_tree_build_ptrs(self, col_ptrs, num_leaves, cap_size, d_tree):
        arr = (C.c_void_p * len(col_ptrs))(*col_ptrs)
        self._check(self._lib.bj_merkle_tree_build_ptrs(self._h, arr, len(col_ptrs), num_leaves, cap_size, d_tree))

    def merkle_tree_build_chunked(self, d_c0, d_c1, length, log_elems_per_leaf, cap_size, d_tree):
        self._check(self._lib.bj_merkle_tree_build_chunked(self._h, d_c0, d_c1, length, log_elems_per_leaf, cap_size, d_tree))

    def merkle_tree_nodes(self, d_tree, num_leaves, cap_size):
        self._check(self._lib.bj_merkle_tree_nodes(self._h, d_tree, num_leaves, cap_size))

    def merkle_tree_cap(self, d_tree, num_leaves, cap_size):
        cap = np.empty((cap_size, 4), dtype=np.uint64)
        self._check(self._lib.bj_merkle_tree_cap(self._h, d_tree, num_leaves, cap_size, _np_ptr(cap)))
        return cap

    def merkle_tree_proof(self, d_tree, num_leaves, cap_size, idx):
        depth = (num_leaves // cap_size).bit_length() - 1
        leaf = np.empty(4, dtype=np.uint64)
        path = np.empty((max(depth, 1), 4), dtype=np.uint64)
        self._check(self._lib.bj_merkle_tree_proof(self._h, d_tree, num_leaves, cap_size, idx, _np_ptr(leaf), _np_ptr(path)))
        return leaf, path[:depth]

    def poseidon2_permute(self, d_states, n_states):
        self._check(self._lib.bj_poseidon2_permute(self._h, d_states, n_states))

    # -- FRI
    def fri_fold(self, d_c0, d_c1, length, d_o0, d_o1, log_full, coset_inv, ch):
        self._check(self._lib.bj_fri_fold(self._h, d_c0, d_c1, length, d_o0, d_o1, log_full, coset_inv, ch[0], ch[1]))
