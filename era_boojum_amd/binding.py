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
    "bj_env_reload": (None, []),
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
    "bj_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_timer_start": (C.c_int, [C.c_void_p]),
    "bj_timer_stop_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "bj_ntt_forward_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t, C.c_uint64]),
    "bj_intt_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t, C.c_uint64]),
    "bj_lde_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]),
    "bj_lde_cosets_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint]),
    "bj_trace_to_lde_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]),
    "bj_bitreverse_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t]),
    "bj_monomials_tiled": (C.c_int, [C.c_uint]),
    "bj_intt_batch_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t]),
    "bj_lde_cosets_batch_tiled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint]),
    "bj_tiled_permute_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_size_t, C.c_int]),
    "bj_canonicalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_field_op_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
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
    "bj_barycentric_weights": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_barycentric_eval_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_deep_quotient_accumulate": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_int]),
    "bj_copy_perm_stage2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint,
                                      C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_lookup_polys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint,
                                  C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_quotient_gates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p,
                                    C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "bj_quotient_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p]),
    "bj_quotient_copy_perm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
    "bj_combine_residues": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p]),
    "bj_rccl_available": (C.c_int, []),
    "bj_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "bj_comm_rccl_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]),
    "bj_comm_rccl_destroy": (None, [C.c_void_p]),
    "bj_comm_rccl_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bj_setup_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bj_comm_replay_create": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]),
    "bj_comm_replay_destroy": (None, [C.c_void_p]),
    "bj_comm_replay_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bj_comm_replay_capture": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_comm_replay_captured": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "bj_gate_program_generated": (C.c_int, [C.c_void_p]),
    "bj_gate_program_canonical_info": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_gate_program_emit_body": (C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "bj_gate_program_jit_source": (C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "bj_gate_program_jit_compile_check": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "bj_gate_jit_status": (C.c_int, [C.c_char_p, C.c_size_t]),
    "bj_gate_program_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint,
                                       C.c_uint, C.c_size_t, C.c_void_p]),
    "bj_setup_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_setup_create_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_void_p)]),
    "bj_setup_destroy": (None, [C.c_void_p]),
    "bj_setup_cap": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bj_setup_device_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "bj_prove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_comm_peer_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bj_comm_peer_destroy": (None, [C.c_void_p]),
    "bj_comm_peer_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bj_prove_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_proof_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_proof_poll": (C.c_int, [C.c_void_p]),
    "bj_prove_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_proof_destroy": (None, [C.c_void_p]),
    "bj_proof_size_u64": (C.c_size_t, [C.c_void_p]),
    "bj_proof_serialize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bj_setup_shape": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_setup_dump_info": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p]),
    "bj_setup_create_from_dump": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_prove_from_dumps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_void_p)]),
    "bj_proof_stage_ms": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bj_proof_workspace_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bj_proof_kernel_stats": (C.c_int, [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_proof_comm_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_fri_fold_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint64, C.c_uint64, C.c_uint64]),
    "bj_transcript_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bj_transcript_destroy": (None, [C.c_void_p]),
    "bj_transcript_absorb_cap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_ctx_set_tree_hasher": (C.c_int, [C.c_void_p, C.c_int]),
    "bj_ctx_release_workspace": (C.c_int, [C.c_void_p]),
    "bj_transcript_absorb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "bj_transcript_challenge": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "bj_transcript_query_index": (C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint64)]),
    "bj_fri_schedule": (C.c_int, [C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bj_fri_prove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint32), C.c_size_t,
                               C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bj_fri_destroy": (None, [C.c_void_p]),
    "bj_fri_num_oracles": (C.c_size_t, [C.c_void_p]),
    "bj_fri_final_degree": (C.c_size_t, [C.c_void_p]),
    "bj_fri_cap": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "bj_fri_challenge": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "bj_fri_final_monomials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bj_fri_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]),
}


class BoojumHipError(RuntimeError):
    pass


def lib_path():
    # BOOJUM_HIP_LIB lets tuning experiments point at an alternative build of the SAME library (never a fallback)
    return os.environ.get("BOOJUM_HIP_LIB") or os.path.join(_HERE, "libboojum_hip.so")


def exported_symbols():
    return sorted(_SIGNATURES)


_lib = None


ABI_VERSION = 6


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
        if lib.bj_abi_version() != ABI_VERSION:     # struct layouts below are those of BJ_ABI_VERSION in include/boojum_hip.h
            raise BoojumHipError("%s has ABI version %d, this binding was written for %d: rebuild (python -m era_boojum_amd.build)"
                                 % (path, lib.bj_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def _np_ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One per GPU: wraps bj_ctx (device, stream, twiddle cache, scratch)."""

    def __init__(self, device=0, stream=None):
        self._lib = load_library()
        self.device = int(device)
        h = C.c_void_p()
        rc = self._lib.bj_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise BoojumHipError("bj_ctx_create(device=%d) failed: %s" % (device, self._lib.bj_status_string(rc).decode()))
        self._h = h
        if stream is not None:
            self.set_stream(stream)

    def gate_program_eval(self, program, d_vars, var_stride, d_consts, const_stride, reps, rep_var_stride, rep_const_stride,
                          n_points, d_terms):
        """bj_gate_program_eval: raw terms of a seam-S3 gate program at n_points points."""
        self._check(self._lib.bj_gate_program_eval(self._h, C.byref(program.struct), C.c_void_p(d_vars), var_stride,
                                                   C.c_void_p(d_consts) if d_consts else None, const_stride, reps, rep_var_stride,
                                                   rep_const_stride, n_points, C.c_void_p(d_terms)))

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
    def set_tree_hasher(self, hasher):
        """1 = Poseidon2 (default), 2 = Blake2s-256: hasher of the merkle_tree_* / fri calls on this context."""
        self._check(self._lib.bj_ctx_set_tree_hasher(self._h, int(hasher)))

    def set_stream(self, stream_handle):
        self._check(self._lib.bj_ctx_set_stream(self._h, C.c_void_p(stream_handle)))

    def sync(self):
        self._check(self._lib.bj_sync(self._h))

    def release_workspace(self):
        """Give the proof workspace (arena, NTT scratch, host-witness staging) back to the device allocator."""
        self._check(self._lib.bj_ctx_release_workspace(self._h))

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

    def lde_cosets_batch(self, d_mono, d_out, log_n, n_cols, log_lde, coset_begin, coset_count, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_lde_cosets_batch(self._h, d_mono, col_stride, d_out, log_n, n_cols, log_lde,
                                                  coset_begin, coset_count))

    def monomials_tiled(self, log_n):
        """True when bj_prove keeps the monomials of 2^log_n-row traces in the tiled layout (include/boojum_hip.h)."""
        return bool(self._lib.bj_monomials_tiled(log_n))

    def intt_batch_tiled(self, d_in, d_out, log_n, n_cols, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_intt_batch_tiled(self._h, d_in, d_out, log_n, n_cols, col_stride))

    def lde_cosets_batch_tiled(self, d_mono, d_out, log_n, n_cols, log_lde, coset_begin, coset_count, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_lde_cosets_batch_tiled(self._h, d_mono, col_stride, d_out, log_n, n_cols, log_lde,
                                                        coset_begin, coset_count))

    def tiled_permute_batch(self, d_in, d_out, log_n, n_cols, to_tiled, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_tiled_permute_batch(self._h, d_in, d_out, log_n, n_cols, col_stride, 1 if to_tiled else 0))

    def trace_to_lde_batch(self, d_cols, d_out, log_n, n_cols, log_lde, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_trace_to_lde_batch(self._h, d_cols, col_stride, d_out, log_n, n_cols, log_lde))

    def bitreverse_batch(self, d_in, d_out, log_n, n_cols, col_stride=None):
        col_stride = (1 << log_n) if col_stride is None else col_stride
        self._check(self._lib.bj_bitreverse_batch(self._h, d_in, d_out, log_n, n_cols, col_stride))

    def canonicalize(self, d, n):
        self._check(self._lib.bj_canonicalize(self._h, d, n))

    # -- seam S2: second round and quotient terms on device columns (pointers are device addresses, challenges (c0, c1) tuples)
    @staticmethod
    def _e2(x):
        return np.array([int(x[0]), int(x[1])], dtype=np.uint64)

    def copy_perm_stage2(self, d_vars, var_stride, d_sigmas, sig_stride, non_residues, num_vars, chunk, log_n, beta, gamma, d_z,
                         d_partials):
        nr = np.ascontiguousarray(non_residues, dtype=np.uint64)
        self._check(self._lib.bj_copy_perm_stage2(self._h, d_vars, var_stride, d_sigmas, sig_stride, _np_ptr(nr), num_vars, chunk,
                                                  log_n, _np_ptr(self._e2(beta)), _np_ptr(self._e2(gamma)), d_z, d_partials))

    def lookup_polys(self, d_lvars, var_stride, d_table_id, d_tables, table_stride, d_mult, reps, width, log_n, beta, gamma, d_A, d_B):
        self._check(self._lib.bj_lookup_polys(self._h, d_lvars, var_stride, d_table_id, d_tables, table_stride, d_mult, reps, width,
                                              log_n, _np_ptr(self._e2(beta)), _np_ptr(self._e2(gamma)), d_A, d_B))

    def quotient_gates(self, d_vars, var_stride, num_gp_vars, d_consts, const_stride, num_constant_cols, gates, alphas, num_points,
                       d_out0, d_out1):
        descs = gate_desc_array(gates)
        al = np.ascontiguousarray(np.array(alphas, dtype=np.uint64).reshape(-1))
        self._check(self._lib.bj_quotient_gates(self._h, d_vars, var_stride, num_gp_vars, d_consts, const_stride, num_constant_cols,
                                                C.cast(descs, C.c_void_p), len(gates), _np_ptr(al) if al.size else None,
                                                num_points, d_out0, d_out1))

    def quotient_lookup(self, d_lvars, var_stride, d_table_id, d_tables, table_stride, d_mult, d_A, d_B, s2_stride, reps, width,
                        beta, gamma, alphas, num_points, d_out0, d_out1):
        al = np.ascontiguousarray(np.array(alphas, dtype=np.uint64).reshape(-1))
        self._check(self._lib.bj_quotient_lookup(self._h, d_lvars, var_stride, d_table_id, d_tables, table_stride, d_mult, d_A, d_B,
                                                 s2_stride, reps, width, _np_ptr(self._e2(beta)), _np_ptr(self._e2(gamma)),
                                                 _np_ptr(al), num_points, d_out0, d_out1))

    def quotient_copy_perm(self, d_vars, var_stride, d_sigmas, sig_stride, d_stage2, s2_stride, non_residues, num_vars, chunk, log_n,
                           log_lde, beta, gamma, alphas, num_points, first_point, d_out0, d_out1):
        nr = np.ascontiguousarray(non_residues, dtype=np.uint64)
        al = np.ascontiguousarray(np.array(alphas, dtype=np.uint64).reshape(-1))
        self._check(self._lib.bj_quotient_copy_perm(self._h, d_vars, var_stride, d_sigmas, sig_stride, d_stage2, s2_stride,
                                                    _np_ptr(nr), num_vars, chunk, log_n, log_lde, _np_ptr(self._e2(beta)),
                                                    _np_ptr(self._e2(gamma)), _np_ptr(al), num_points, first_point, d_out0, d_out1))

    def combine_residues(self, d_residues, world, residue_len, num_cols, moduli, d_out):
        """bj_combine_residues: [world][num_cols][E] residues T mod (x^E - a_i) -> [num_cols][world * E] coefficients of T."""
        a = np.ascontiguousarray(np.array([int(v) for v in moduli], dtype=np.uint64))
        self._check(self._lib.bj_combine_residues(self._h, d_residues, world, residue_len, num_cols, _np_ptr(a), d_out))

    FIELD_OPS = {"add": 0, "sub": 1, "mul": 2, "mul_lazy": 3, "square": 4, "inverse": 5, "ext2_mul": 6, "butterfly": 7, "addsub": 8,
                 "add_lazy": 9, "sub_lazy": 10, "ext2_mul_lazy": 11}

    def field_op(self, op, a, b=None):
        """bj_field_op_batch on host arrays (uploads, runs, downloads): elementwise Goldilocks / F_p^2 operators."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        n = a.size // 2 if op in ("ext2_mul", "ext2_mul_lazy", "butterfly", "addsub") else a.size
        da = self.upload(a)
        db = self.upload(np.ascontiguousarray(b, dtype=np.uint64)) if b is not None else None
        do = self.malloc(a.nbytes)
        try:
            self._check(self._lib.bj_field_op_batch(self._h, self.FIELD_OPS[op], C.c_void_p(da), C.c_void_p(db) if db else None,
                                                    C.c_void_p(do), n))
            return self.d2h(do, a.shape)
        finally:
            for p in (da, db, do):
                if p:
                    self.free(p)

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

    # -- openings
    def barycentric_weights(self, log_n, coset, at, d_w0, d_w1):
        a = np.array(at, dtype=np.uint64)
        self._check(self._lib.bj_barycentric_weights(self._h, log_n, coset, _np_ptr(a), d_w0, d_w1))

    def barycentric_eval_batch(self, col_ptrs, log_n, d_w0, d_w1):
        arr = (C.c_void_p * len(col_ptrs))(*col_ptrs)
        out = np.empty((len(col_ptrs), 2), dtype=np.uint64)
        self._check(self._lib.bj_barycentric_eval_batch(self._h, arr, len(col_ptrs), log_n, d_w0, d_w1, _np_ptr(out)))
        return out

    def deep_quotient_accumulate(self, sources, values, challenges, at, log_n, log_lde, d_dst0, d_dst1, accumulate=True):
        """sources: list of (d_c0, d_c1_or_None) device pointers of LDE columns."""
        k = len(sources)
        p0 = (C.c_void_p * k)(*[a for a, _ in sources])
        p1 = (C.c_void_p * k)(*[b for _, b in sources])
        v = np.ascontiguousarray(np.array(values, dtype=np.uint64).reshape(-1))
        ch = np.ascontiguousarray(np.array(challenges, dtype=np.uint64).reshape(-1))
        a = np.array(at, dtype=np.uint64)
        self._check(self._lib.bj_deep_quotient_accumulate(self._h, p0, p1, k, _np_ptr(v), _np_ptr(ch), _np_ptr(a), log_n,
                                                          log_lde, d_dst0, d_dst1, int(bool(accumulate))))

    def fri_fold_step(self, d_c0, d_c1, length, k, d_o0, d_o1, log_full, coset_inv, ch):
        self._check(self._lib.bj_fri_fold_step(self._h, d_c0, d_c1, length, k, d_o0, d_o1, log_full, coset_inv, ch[0], ch[1]))

    def fri_prove(self, d_c0, d_c1, log_n, log_lde, schedule, cap_size, transcript):
        """do_fri on the device; returns a FriProof handle (oracles stay in HBM)."""
        sched = (C.c_uint32 * len(schedule))(*schedule)
        h = C.c_void_p()
        self._check(self._lib.bj_fri_prove(self._h, d_c0, d_c1, log_n, log_lde, sched, len(schedule), cap_size,
                                           transcript._h, C.byref(h)))
        return FriProof(self, h, cap_size, list(schedule))


class Transcript:
    """Host-side Poseidon2 Fiat–Shamir transcript of the product (bj_transcript_*); needs no GPU."""

    def __init__(self, kind=1):
        """kind 1 = Poseidon2 (BJ_TRANSCRIPT_POSEIDON2), 2 = Poseidon v1 (BJ_TRANSCRIPT_POSEIDON)."""
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.bj_transcript_create(int(kind), C.byref(h))
        if rc != 0:
            raise BoojumHipError("bj_transcript_create failed: %d" % rc)
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.bj_transcript_destroy(self._h)
            self._h = None

    def absorb(self, els):
        a = np.ascontiguousarray(np.asarray(els, dtype=np.uint64)).reshape(-1)
        if a.size:
            rc = self._lib.bj_transcript_absorb(self._h, _np_ptr(a), a.size)
            if rc != 0:
                raise BoojumHipError("bj_transcript_absorb failed: %d" % rc)

    def absorb_cap(self, cap):
        a = np.ascontiguousarray(np.asarray(cap, dtype=np.uint64)).reshape(-1)
        if a.size:
            rc = self._lib.bj_transcript_absorb_cap(self._h, _np_ptr(a), a.size)
            if rc != 0:
                raise BoojumHipError("bj_transcript_absorb_cap failed: %d" % rc)

    def challenge(self):
        out = C.c_uint64()
        rc = self._lib.bj_transcript_challenge(self._h, C.byref(out))
        if rc != 0:
            raise BoojumHipError("bj_transcript_challenge failed: %d" % rc)
        return out.value

    def challenge_ext(self):
        return (self.challenge(), self.challenge())

    def query_index(self, log_n, log_lde):
        out = C.c_uint64()
        rc = self._lib.bj_transcript_query_index(self._h, log_n, log_lde, C.byref(out))
        if rc != 0:
            raise BoojumHipError("bj_transcript_query_index failed: %d" % rc)
        return out.value


def fri_schedule(security_bits, cap_size, pow_bits, rate_log2, initial_degree_log2):
    """compute_fri_schedule (prover.rs:2281-2372) through the C ABI: (new_pow_bits, num_queries, schedule, final_degree)."""
    lib = load_library()
    sched = (C.c_uint32 * 32)()
    new_pow, nq, ln, fd = C.c_uint32(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    rc = lib.bj_fri_schedule(security_bits, cap_size, pow_bits, rate_log2, initial_degree_log2, C.byref(new_pow),
                             C.byref(nq), sched, C.byref(ln), C.byref(fd))
    if rc != 0:
        raise BoojumHipError("bj_fri_schedule failed: %d" % rc)
    return new_pow.value, nq.value, [int(sched[i]) for i in range(ln.value)], fd.value


class FriProof:
    def __init__(self, ctx, handle, cap_size, schedule):
        self._ctx, self._h, self.cap_size, self.schedule = ctx, handle, cap_size, schedule
        self._lib = ctx._lib

    def close(self):
        if self._h:
            self._lib.bj_fri_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_oracles(self):
        return self._lib.bj_fri_num_oracles(self._h)

    @property
    def final_degree(self):
        return self._lib.bj_fri_final_degree(self._h)

    def cap(self, i):
        out = np.empty((self.cap_size, 4), dtype=np.uint64)
        self._ctx._check(self._lib.bj_fri_cap(self._h, i, _np_ptr(out)))
        return out

    def challenge(self, i):
        out = np.empty(2, dtype=np.uint64)
        self._ctx._check(self._lib.bj_fri_challenge(self._h, i, _np_ptr(out)))
        return (int(out[0]), int(out[1]))

    def final_monomials(self):
        fd = self.final_degree
        c0, c1 = np.empty(fd, dtype=np.uint64), np.empty(fd, dtype=np.uint64)
        self._ctx._check(self._lib.bj_fri_final_monomials(self._h, _np_ptr(c0), _np_ptr(c1)))
        return c0, c1

    def query(self, oracle, index, oracle_len):
        e = 1 << self.schedule[oracle]
        leaf = np.empty(2 * e, dtype=np.uint64)
        depth = ((oracle_len >> self.schedule[oracle]) // self.cap_size).bit_length() - 1
        path = np.empty((max(depth, 1), 4), dtype=np.uint64)
        self._ctx._check(self._lib.bj_fri_query(self._ctx._h, self._h, oracle, index, _np_ptr(leaf), _np_ptr(path)))
        return leaf, path[:depth]


class _GateDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("path_len", C.c_uint), ("path", C.c_ubyte * 8), ("num_repetitions", C.c_uint),
                ("var_stride", C.c_uint), ("wit_stride", C.c_uint), ("const_stride", C.c_uint), ("num_terms", C.c_uint),
                ("program", C.c_void_p)]


def gate_desc_array(gates):
    """bj_gate_desc[] for gates with the attributes of synthetic.Gate (kind, path, reps, var_stride, const_stride, num_terms)."""
    arr = (_GateDesc * len(gates))()
    for i, g in enumerate(gates):
        arr[i].kind, arr[i].path_len = int(g.kind), len(g.path)
        for b, bit in enumerate(g.path):
            arr[i].path[b] = 1 if bit else 0
        arr[i].num_repetitions, arr[i].var_stride, arr[i].const_stride = int(g.reps), int(g.var_stride), int(g.const_stride)
        arr[i].num_terms, arr[i].program = int(g.num_terms), None
    return arr


class _Circuit(C.Structure):
    _fields_ = [("log_n", C.c_uint), ("num_vars", C.c_uint), ("num_gp_vars", C.c_uint), ("num_witness_cols", C.c_uint),
                ("num_constant_cols", C.c_uint), ("lookup_width", C.c_uint), ("lookup_reps", C.c_uint),
                ("table_id_col", C.c_uint), ("quotient_degree", C.c_uint), ("num_gates", C.c_uint),
                ("gates", C.POINTER(_GateDesc)), ("non_residues", C.POINTER(C.c_uint64)), ("num_public_inputs", C.c_uint),
                ("public_input_cols", C.POINTER(C.c_uint)), ("public_input_rows", C.POINTER(C.c_uint)),
                ("num_specialized_gates", C.c_uint), ("specialized_gates", C.POINTER(_GateDesc))]


class _ProofConfig(C.Structure):
    _fields_ = [("fri_lde_factor", C.c_uint), ("cap_size", C.c_uint), ("security_level", C.c_uint), ("pow_bits", C.c_uint),
                ("transcript", C.c_uint), ("tree_hasher", C.c_uint), ("pow_runner", C.c_uint)]


STAGE_NAMES = ["witness_lde_and_tree", "second_stage", "quotient_work_and_lde", "openings_at_z",
               "batched_fri_opening_computation", "fri", "queries"]


_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_ALL_GATHER_STREAM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class _Comm(C.Structure):  # bj_comm
    _fields_ = [("rank", C.c_uint), ("world", C.c_uint), ("all_gather", _ALL_GATHER_FN), ("user", C.c_void_p),
                ("all_gather_stream", _ALL_GATHER_STREAM_FN)]


RCCL_UNIQUE_ID_BYTES = 128


def rccl_unique_id():
    """bj_rccl_unique_id: 128 opaque bytes made by rank 0 and handed to every rank of the proof."""
    buf = C.create_string_buffer(RCCL_UNIQUE_ID_BYTES)
    rc = load_library().bj_rccl_unique_id(buf)
    if rc != 0:
        raise BoojumHipError("bj_rccl_unique_id failed: %s" % load_library().bj_status_string(rc).decode())
    return buf.raw


class RcclComm:
    """The in-library transport (bj_comm_rccl_create): ncclAllGather on the context's stream, directly on the prover's buffers.
    `unique_id` = the 128 bytes of rccl_unique_id() from rank 0 (ship them with any channel, e.g. torch.distributed's
    broadcast_object_list).  Collective: every rank of the proof constructs it."""

    def __init__(self, ctx, unique_id, rank, world):
        self._ctx, self._lib = ctx, ctx._lib
        self.rank, self.world = rank, world
        self.struct = _Comm()
        ctx._check(self._lib.bj_comm_rccl_create(ctx._h, C.c_char_p(unique_id), rank, world, C.byref(self.struct)))

    @property
    def calls(self):
        return self._stats()[0]

    @property
    def bytes(self):
        return self._stats()[1]

    def _stats(self):
        a, b = C.c_size_t(), C.c_size_t()
        self._lib.bj_comm_rccl_stats(C.byref(self.struct), C.byref(a), C.byref(b))
        return a.value, b.value

    def all_gather(self, d_send, d_recv, nbytes, stream=None):
        """One all-gather through the transport's own entry (bj_comm.all_gather_stream when `stream` is given, the blocking
        bj_comm.all_gather otherwise): what the sharded prover calls, exposed for self-tests."""
        if stream is not None:
            rc = self.struct.all_gather_stream(self.struct.user, C.c_void_p(d_send), C.c_void_p(d_recv), nbytes, C.c_void_p(stream))
        else:
            rc = self.struct.all_gather(self.struct.user, C.c_void_p(d_send), C.c_void_p(d_recv), nbytes)
        if rc != 0:
            raise BoojumHipError("RcclComm.all_gather failed (%d)" % rc)

    def close(self):
        if self.struct.user:
            self._lib.bj_comm_rccl_destroy(C.byref(self.struct))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReplayComm:
    """The recorded-peer transport (bj_comm_replay_create): one rank of a `world`-rank proof alone on its GPU, every all-gather
    answered by a device copy of what that collective gathered in a real run.  `recorded` = [(device pointer, bytes)] in call
    order: n_setup buffers for the setup's collectives, then the buffers of one proof (used cyclically)."""

    def __init__(self, ctx, rank, world, recorded, n_setup=0, verify=False, keepalive=None, capture=None):
        """keepalive: whatever owns the recorded device buffers (e.g. the list of torch tensors) — held for the life of the comm.
        capture = (device pointer, capacity in bytes): the first collective past the recorded ones is not served; this rank's
        contribution is copied there and the call fails, ending the proof (bj_comm_replay_capture); `captured` then says how many
        bytes were taken."""
        self._ctx, self._lib = ctx, ctx._lib
        self.rank, self.world = rank, world
        self._keep = keepalive
        self._ptrs = (C.c_void_p * max(1, len(recorded)))(*[p for p, _ in recorded])
        self._sizes = (C.c_size_t * max(1, len(recorded)))(*[b for _, b in recorded])
        self.struct = _Comm()
        ctx._check(self._lib.bj_comm_replay_create(ctx._h, rank, world, self._ptrs, self._sizes, n_setup, len(recorded) - n_setup,
                                                   1 if verify else 0, C.byref(self.struct)))
        if capture is not None:
            ctx._check(self._lib.bj_comm_replay_capture(C.byref(self.struct), C.c_void_p(capture[0]), capture[1]))

    @property
    def captured(self):
        b = C.c_size_t()
        self._lib.bj_comm_replay_captured(C.byref(self.struct), C.byref(b))
        return int(b.value)

    def stats(self):
        a, b, m = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._lib.bj_comm_replay_stats(C.byref(self.struct), C.byref(a), C.byref(b), C.byref(m))
        return a.value, b.value, m.value

    calls = property(lambda self: self.stats()[0])
    bytes = property(lambda self: self.stats()[1])
    mismatches = property(lambda self: self.stats()[2])

    def close(self):
        if self.struct.user:
            self._lib.bj_comm_replay_destroy(C.byref(self.struct))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ThreadGroup:
    """`world` ranks of ONE sharded proof as threads of this process, all on one GPU (each with its own Context = its own stream
    and arena): the all-gather is a rendezvous of the threads and device-to-device copies.  Functionally the sharded prover as
    it runs over RCCL (the ranks time-slice the GPU, so its timing means nothing); with record=True rank 0 keeps a copy of
    every gathered buffer — what ReplayComm then serves to one rank running alone (bench.py --replay-world)."""

    def __init__(self, world, record=False):
        import threading
        self.world, self.record = world, record
        self._barrier = threading.Barrier(world)
        self._posted = [None] * world
        self.recorded = []            # [(torch uint8 tensor)] in call order (rank 0's view)
        self.marks = []               # len(recorded) at the moments mark() was called
        self.error = None

    def mark(self):
        self.marks.append(len(self.recorded))

    def comm(self, ctx, rank):
        return _ThreadComm(self, ctx, rank)


class _ThreadComm:
    def __init__(self, group, ctx, rank):
        self._g, self._ctx, self.rank, self.world = group, ctx, rank, group.world
        self.calls, self.bytes = 0, 0
        self._fn = _ALL_GATHER_FN(self._all_gather)
        self.struct = _Comm(rank, group.world, self._fn, None, _ALL_GATHER_STREAM_FN())

    def _all_gather(self, _user, d_send, d_recv, nbytes):
        g = self._g
        try:
            g._posted[self.rank] = (d_send, nbytes)
            g._barrier.wait()                   # every rank's stream is idle here (the library's contract for this entry)
            ctx, lib = self._ctx, self._ctx._lib
            for r in range(self.world):
                src, nb = g._posted[r]
                if nb != nbytes:
                    raise BoojumHipError("ranks disagree on the size of a collective (%d vs %d bytes)" % (nb, nbytes))
                ctx._check(lib.bj_memcpy_d2d(ctx._h, C.c_void_p(d_recv + r * nbytes), C.c_void_p(src), nbytes))
            if g.record and self.rank == 0:
                import torch
                t = torch.empty(nbytes * self.world, dtype=torch.uint8, device=torch.device("cuda", ctx.device))
                ctx._check(lib.bj_memcpy_d2d(ctx._h, C.c_void_p(t.data_ptr()), C.c_void_p(d_recv), nbytes * self.world))
                g.recorded.append(t)
            g._barrier.wait()                   # nobody overwrites its send buffer before every peer has read it
            self.calls += 1
            self.bytes += nbytes * self.world
            return 0
        except Exception as e:                  # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            g.error = e
            try:
                g._barrier.abort()
            except Exception:
                pass
            return 1


class TorchComm:
    """bj_comm over torch.distributed: the all-gather the sharded prover asks the host for.

    The library hands over raw device pointers; they are copied (device to device) through torch-owned staging tensors
    so that the collective itself is torch.distributed's: backend nccl (= RCCL over xGMI) runs on the staging tensors
    directly, backend gloo (CPU tests, or several ranks sharing one GPU) bounces through host memory."""

    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist
        self._ctx, self._torch, self._dist, self._group = ctx, torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._device = torch.device("cuda", ctx.device)
        self._on_device = dist.get_backend(group) == "nccl"
        self._send = self._recv = None
        self.calls, self.bytes = 0, 0
        self._fn = _ALL_GATHER_FN(self._all_gather)        # keep the trampoline alive
        self.struct = _Comm(self.rank, self.world, self._fn, None, _ALL_GATHER_STREAM_FN())

    def _staging(self, nbytes):
        t = self._torch
        if self._send is None or self._send.numel() < nbytes:
            self._send = t.empty(nbytes, dtype=t.uint8, device=self._device)
            self._recv = t.empty(nbytes * self.world, dtype=t.uint8, device=self._device)
        return self._send[:nbytes], self._recv[:nbytes * self.world]

    def _all_gather(self, _user, d_send, d_recv, nbytes):
        try:
            ctx, lib = self._ctx, self._ctx._lib
            send, recv = self._staging(nbytes)
            ctx._check(lib.bj_memcpy_d2d(ctx._h, C.c_void_p(send.data_ptr()), C.c_void_p(d_send), nbytes))
            if self._on_device:
                self._dist.all_gather_into_tensor(recv, send, group=self._group)
                self._torch.cuda.synchronize(self._device)
            else:
                h_send = send.cpu()
                h_recv = self._torch.empty(nbytes * self.world, dtype=self._torch.uint8)
                self._dist.all_gather_into_tensor(h_recv, h_send, group=self._group)
                recv.copy_(h_recv)
                self._torch.cuda.synchronize(self._device)
            ctx._check(lib.bj_memcpy_d2d(ctx._h, C.c_void_p(d_recv), C.c_void_p(recv.data_ptr()), nbytes * self.world))
            self.calls += 1
            self.bytes += nbytes * self.world
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            self.error = e
            return 1


_HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class PeerComm:
    """bj_comm_peer_create: full-mesh peer copies for the bulk exchanges of a sharded proof (every rank writes its contribution
    straight into every peer's receive buffer, mapped through HIP IPC), everything smaller through `base` (a TorchComm or RcclComm).
    The control channel — a blocking all-gather of a few host bytes — is torch.distributed on `ctrl_group` (default: the default
    group): host tensors over gloo, tiny device tensors over nccl.  Every rank of the proof constructs it."""

    def __init__(self, ctx, base, ctrl_group=None, bulk_threshold_bytes=1 << 20):
        import torch
        import torch.distributed as dist
        self._ctx, self._lib, self._base = ctx, ctx._lib, base
        self._torch, self._dist = torch, dist
        self.rank, self.world = base.rank, base.world
        # the control channel rides on the process group the host already has (no second rendezvous that could fail on its own):
        # host tensors over gloo, tiny device tensors over nccl (= RCCL)
        self._ctrl = ctrl_group
        self._on_device = dist.get_backend(ctrl_group) == "nccl"
        self._device = torch.device("cuda", ctx.device)
        self.error = None
        self._fn = _HOST_EXCHANGE_FN(self._exchange)             # keep the trampoline alive
        self.struct = _Comm()
        ctx._check(self._lib.bj_comm_peer_create(ctx._h, C.byref(base.struct), self._fn, None, bulk_threshold_bytes, C.byref(self.struct)))

    def _exchange(self, _user, h_send, h_recv, nbytes):
        try:
            t = self._torch
            send = t.frombuffer((C.c_ubyte * nbytes).from_address(h_send), dtype=t.uint8).clone()
            if self._on_device:
                d_send = send.to(self._device)
                d_recv = t.empty(nbytes * self.world, dtype=t.uint8, device=self._device)
                self._dist.all_gather_into_tensor(d_recv, d_send, group=self._ctrl)
                recv = d_recv.cpu()
            else:
                recv = t.empty(nbytes * self.world, dtype=t.uint8)
                self._dist.all_gather_into_tensor(recv, send, group=self._ctrl)
            C.memmove(h_recv, recv.data_ptr(), nbytes * self.world)
            return 0
        except Exception as e:  # an exception must not unwind through the C frames
            import traceback
            traceback.print_exc()
            self.error = e
            return 1

    def stats(self):
        a, b, c, d = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._lib.bj_comm_peer_stats(C.byref(self.struct), C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return {"bulk_calls": a.value, "bulk_bytes_received": b.value, "small_calls": c.value, "fallbacks": d.value}

    @property
    def calls(self):
        s = self.stats()
        return s["bulk_calls"] + s["small_calls"]

    @property
    def bytes(self):
        return self.stats()["bulk_bytes_received"] + getattr(self._base, "bytes", 0)

    def close(self):
        if self.struct.user:
            self._lib.bj_comm_peer_destroy(C.byref(self.struct))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Ticket:
    """A proof in flight (bj_ticket) + the host arrays it reads."""

    def __init__(self, handle, keep):
        self.handle, self.keep = handle, keep


class ProverSetup:
    """Device-resident setup for a circuit of era_boojum_amd.synthetic.Circuit shape (bj_setup_create).

    With `comm` (a TorchComm) the LDE cosets of every column are split across the ranks of the process group and
    `prove` / `prove_dev` become collective: every rank passes the same witness and gets the same proof
    (bj_setup_create_sharded)."""

    def __init__(self, ctx, circuit, fri_lde_factor=8, cap_size=16, security_level=100, pow_bits=0, comm=None,
                 transcript="poseidon2", setup_base_dump=None, pow_runner="blake2s"):
        """pow_runner: "blake2s" | "keccak256" — the PoWRunner (pow.rs), independent of the transcript as in the reference.
        setup_base_dump: the bytes of the reference's `SetupBaseStorage::write_into_buffer` (bj_setup_create_from_dump): the
        columns, constant-column count, table-id column, selector paths, quotient degree and non-residues then come from the
        dump / are computed by the library, and `circuit` only has to carry the geometry and the gate list."""
        self._ctx, self._lib, self.circuit = ctx, ctx._lib, circuit
        self._comm = comm
        self.fri_lde_factor, self.cap_size, self.security_level, self.pow_bits = fri_lde_factor, cap_size, security_level, pow_bits
        c = circuit
        gates = (_GateDesc * len(c.gates))()
        from_dump = setup_base_dump is not None
        for i, g in enumerate(c.gates):
            gates[i].kind = g.kind
            gates[i].path_len = 0 if from_dump else len(g.path)
            for b, bit in enumerate([] if from_dump else g.path):
                gates[i].path[b] = 1 if bit else 0
            gates[i].num_repetitions, gates[i].var_stride, gates[i].const_stride, gates[i].num_terms = \
                g.reps, g.var_stride, g.const_stride, g.num_terms
            gates[i].wit_stride = getattr(g, "wit_stride", 0)
            prog = getattr(g, "program", None)       # seam S3: evaluate this gate from its op list (gate_program.py)
            if prog is not None:
                gates[i].kind = 5
                gates[i].program = C.cast(C.pointer(prog.struct), C.c_void_p)
        spec_list = list(getattr(c, "specialized_gates", []) or [])
        spec = (_GateDesc * max(1, len(spec_list)))()
        for i, g in enumerate(spec_list):                # gates over specialized columns: op lists, no selector
            spec[i].kind, spec[i].path_len = 5, 0
            spec[i].num_repetitions, spec[i].var_stride, spec[i].const_stride, spec[i].num_terms = g.reps, g.var_stride, g.const_stride, g.num_terms
            spec[i].program = C.cast(C.pointer(g.program.struct), C.c_void_p)
        nr = np.array(c.non_residues, dtype=np.uint64)
        cols = (C.c_uint * max(1, len(c.public_inputs)))(*[p[0] for p in c.public_inputs])
        rows = (C.c_uint * max(1, len(c.public_inputs)))(*[p[1] for p in c.public_inputs])
        self.num_witness_cols = int(getattr(c, "num_witness_cols", 0))
        cc = _Circuit(c.log_n, c.num_vars, c.num_gp_vars, self.num_witness_cols, 0 if from_dump else c.num_constant_cols, c.lookup_width,
                      c.lookup_reps, 0 if from_dump else c.table_id_col, 0 if from_dump else c.quotient_degree, len(c.gates), gates,
                      None if from_dump else nr.ctypes.data_as(C.POINTER(C.c_uint64)), len(c.public_inputs),
                      cols, rows, len(spec_list), spec if spec_list else None)
        self.transcript_kind = {"poseidon2": 1, "poseidon": 2, "blake2s": 3, "keccak256": 4}[transcript]
        self.hasher_kind = {"blake2s": 2, "keccak256": 3}.get(transcript, 1)      # Transcript::CompatibleCap = TreeHasher::Output
        cfg = _ProofConfig(fri_lde_factor, cap_size, security_level, pow_bits, self.transcript_kind, self.hasher_kind,
                           {"blake2s": 1, "keccak256": 2}[pow_runner])
        sig = np.ascontiguousarray(c.sigmas, dtype=np.uint64)
        con = np.ascontiguousarray(c.constants, dtype=np.uint64)
        tab = np.ascontiguousarray(c.tables, dtype=np.uint64)
        h = C.c_void_p()
        if from_dump:
            if comm is not None:
                raise BoojumHipError("a setup from a dump is single-GPU")
            blob = bytes(setup_base_dump)
            ctx._check(self._lib.bj_setup_create_from_dump(ctx._h, C.byref(cc), blob, len(blob), C.byref(cfg), C.byref(h)))
            self._h = h
            return
        ctx._check(self._lib.bj_setup_create_sharded(ctx._h, C.byref(cc), _np_ptr(sig), _np_ptr(con),
                                                     _np_ptr(tab) if c.lookup_reps else None, C.byref(cfg),
                                                     C.byref(comm.struct) if comm is not None else None, C.byref(h)))
        self._h = h

    def device_bytes(self):
        """HBM held by this setup (its shard when sharded): bj_setup_device_bytes."""
        b = C.c_size_t()
        self._ctx._check(self._lib.bj_setup_device_bytes(self._h, C.byref(b)))
        return int(b.value)

    def set_comm(self, comm):
        """bj_setup_set_comm: another transport for the same shard (same rank / world)."""
        self._ctx._check(self._lib.bj_setup_set_comm(self._h, C.byref(comm.struct)))
        self._comm = comm

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bj_setup_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cap(self):
        out = np.empty((self.cap_size, 4), dtype=np.uint64)
        self._ctx._check(self._lib.bj_setup_cap(self._h, _np_ptr(out)))
        return out

    def _finish(self, h):
        n = self._lib.bj_proof_size_u64(h)
        buf = np.empty(n, dtype=np.uint64)
        self._ctx._check(self._lib.bj_proof_serialize(h, _np_ptr(buf)))
        ms = (C.c_float * 8)()
        self._lib.bj_proof_stage_ms(h, ms)
        cms, calls, recv = C.c_float(), C.c_size_t(), C.c_size_t()
        self._lib.bj_proof_comm_stats(h, C.byref(cms), C.byref(calls), C.byref(recv))
        self.last_comm = {"ms_in_collectives": float(cms.value), "calls": int(calls.value), "bytes_received": int(recv.value)}
        self.last_kernels = {}        # name -> (ms, algorithmic bytes) of the first launch of each probed kernel in this proof
        name, kms, kb, k = C.c_char_p(), C.c_float(), C.c_double(), 0
        while self._lib.bj_proof_kernel_stats(h, k, C.byref(name), C.byref(kms), C.byref(kb)) == 0:
            self.last_kernels[name.value.decode()] = (float(kms.value), float(kb.value))
            k += 1
        res, hw, slabs = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._lib.bj_proof_workspace_bytes(h, C.byref(res), C.byref(hw), C.byref(slabs))
        self.last_workspace = {"reserved_bytes": int(res.value), "high_water_bytes": int(hw.value), "overflow_slabs": int(slabs.value)}
        self._lib.bj_proof_destroy(h)
        if slabs.value and not os.environ.get("BJ_ALLOW_WORKSPACE_OVERFLOW"):
            # the proof is right, but the reservation of prove_impl was not an upper bound for this geometry: every proof the test
            # suite makes passes through here, so the list of buffers in csrc/prover.hip cannot drift away from the allocations
            raise BoojumHipError("the proof needed %d overflow slab(s): reserved %d bytes, high-water mark %d bytes"
                                 % (slabs.value, res.value, hw.value))
        stages = dict(zip(STAGE_NAMES, [float(x) for x in ms][:7]))
        stages["witness_tree_leaf_kernel"] = float(ms[7])
        return buf, stages

    def prove(self, variables=None, multiplicities=None, public_values=None):
        """Host-memory entry point (bj_prove): returns (serialised proof u64 array, per-stage ms)."""
        c = self.circuit
        v = np.ascontiguousarray(c.variables if variables is None else variables, dtype=np.uint64)
        if self.num_witness_cols and v.shape[0] == c.num_vars:     # the non-copiable witness columns travel behind the variables
            v = np.ascontiguousarray(np.concatenate([v, c.witness], axis=0))
        m = np.ascontiguousarray(c.multiplicities if multiplicities is None else multiplicities, dtype=np.uint64)
        pv = np.array([p[2] for p in c.public_inputs] if public_values is None else public_values, dtype=np.uint64)
        if pv.size == 0:
            pv = np.zeros(1, dtype=np.uint64)
        h = C.c_void_p()
        self._ctx._check(self._lib.bj_prove(self._ctx._h, self._h, _np_ptr(v), _np_ptr(m) if c.lookup_reps else None,
                                            _np_ptr(pv), C.byref(h)))
        return self._finish(h)

    def prove_async(self, variables=None, multiplicities=None, public_values=None):
        """bj_prove_async: queues the proof on one of the context's two lanes and returns a ticket for `wait`.  Arrays passed in
        are used in place when they are contiguous uint64 (e.g. views of pinned tensors) and kept alive by the ticket."""
        c = self.circuit
        v = np.ascontiguousarray(c.variables if variables is None else variables, dtype=np.uint64)
        if self.num_witness_cols and v.shape[0] == c.num_vars:
            v = np.ascontiguousarray(np.concatenate([v, c.witness], axis=0))
        m = np.ascontiguousarray(c.multiplicities if multiplicities is None else multiplicities, dtype=np.uint64)
        pv = np.array([p[2] for p in c.public_inputs] if public_values is None else public_values, dtype=np.uint64)
        if pv.size == 0:
            pv = np.zeros(1, dtype=np.uint64)
        t = C.c_void_p()
        self._ctx._check(self._lib.bj_prove_async(self._ctx._h, self._h, _np_ptr(v), _np_ptr(m) if c.lookup_reps else None,
                                                  _np_ptr(pv), C.byref(t)))
        return _Ticket(t, (v, m))

    def done(self, ticket):
        """bj_proof_poll: True when `wait` would not block."""
        return bool(self._lib.bj_proof_poll(ticket.handle))

    def wait(self, ticket):
        """bj_proof_wait: blocks until the ticket's proof is complete; returns (serialised proof, per-stage ms) like `prove`."""
        h = C.c_void_p()
        t, ticket.handle = ticket.handle, None
        rc = self._lib.bj_proof_wait(t, C.byref(h))
        ticket.keep = None
        self._ctx._check(rc)
        return self._finish(h)

    def prove_from_dumps(self, witness_vec_dump, variables_hint_dump, witness_hint_dump=None):
        """bj_prove_from_dumps: the reference's `WitnessVec` and `DenseVariablesCopyHint` (+ `DenseWitnessCopyHint`) bytes in, the
        proof out."""
        w, v = bytes(witness_vec_dump), bytes(variables_hint_dump)
        x = bytes(witness_hint_dump) if witness_hint_dump is not None else None
        h = C.c_void_p()
        self._ctx._check(self._lib.bj_prove_from_dumps(self._ctx._h, self._h, w, len(w), v, len(v), x, len(x) if x else 0, C.byref(h)))
        return self._finish(h)

    def prove_dev(self, d_variables, d_multiplicities, public_values=None):
        """Witness already in HBM (bj_prove_dev)."""
        c = self.circuit
        pv = np.array([p[2] for p in c.public_inputs] if public_values is None else public_values, dtype=np.uint64)
        if pv.size == 0:
            pv = np.zeros(1, dtype=np.uint64)
        h = C.c_void_p()
        self._ctx._check(self._lib.bj_prove_dev(self._ctx._h, self._h, d_variables, d_multiplicities, _np_ptr(pv), C.byref(h)))
        return self._finish(h)
