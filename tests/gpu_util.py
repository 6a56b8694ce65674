"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np

import era_boojum_amd as E

P = E.P
_ctx = None


def ctx():
    """One context for the whole test session; raises (no skip, no fallback) when there is no GPU."""
    global _ctx
    if _ctx is None:
        # PyTorch-ROCm ships its own libamdhip64; whichever copy is loaded first serves the whole process, and torch cannot
        # initialise on top of the system copy.  bench.py and the sharded workers import torch first — do the same here so
        # that tests which use torch.distributed in this process (TorchComm over RCCL) see the GPU.
        import os
        import sys
        dbg = os.environ.get("BJ_TEST_DEBUG")
        if dbg:
            print("hip libraries mapped before torch:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "hip" in l or "hsa" in l}), file=sys.stderr)
        try:
            import torch
            torch.cuda.init()
        except Exception as e:
            if dbg:
                print("torch.cuda.init failed:", repr(e), file=sys.stderr)
        if dbg:
            print("hip libraries mapped after torch:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "hip" in l or "hsa" in l}), file=sys.stderr)
        _ctx = E.Context(0)
    return _ctx


def oracle_threads(default=64):
    """OpenMP threads for the oracle's bulk operations: the container's CPU quota when there is one (the GPU boxes show 256 CPUs in the
    affinity mask and deliver 16 cores: 64 threads on them cost the coset-streaming restatement 8 %, tools/oracle_threads_probe.py)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, default))


def rand_gl(rng, shape, noncanonical=False):
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    if noncanonical:
        mask = rng.random(size=shape) < 0.05
        a = np.where(mask, np.uint64(P) + rng.integers(0, (1 << 32) - 1, size=shape, dtype=np.uint64), a)
        flat = a.reshape(-1)
        if flat.size:
            flat[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        if flat.size > 1:
            flat[-1] = np.uint64(P)
    return a


class DevBuf:
    def __init__(self, arr=None, nelems=None):
        c = ctx()
        if arr is not None:
            arr = np.ascontiguousarray(arr, dtype=np.uint64)
            self.n = arr.size
            self.ptr = c.upload(arr)
        else:
            self.n = nelems
            self.ptr = c.malloc(max(8, 8 * nelems))

    def get(self, shape=None):
        return ctx().d2h(self.ptr, shape if shape is not None else (self.n,))

    def free(self):
        if self.ptr:
            ctx().free(self.ptr)
            self.ptr = None
