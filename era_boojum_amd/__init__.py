"""era_boojum_amd — MI355X-native Boojum proving hot path.

The product is ``libboojum_hip.so`` (hand-written HIP kernels for gfx950 behind the C ABI of ``include/boojum_hip.h``).
This Python package is only a thin ctypes binding of that ABI for tests and ``bench.py`` (the reference's host
language, Rust, is not available in this image; INTEGRATION.md shows the Rust-side shim).  There is no CPU fallback:
importing works anywhere, but creating a :class:`Context` without a HIP device raises.
"""
from .binding import (BoojumHipError, Context, FriProof, P, PeerComm, ProverSetup, RcclComm, ReplayComm, ThreadGroup, TorchComm, Transcript,  # noqa: F401
                      exported_symbols,
                      fri_schedule, lib_path, load_library)
