#!/usr/bin/env python3
"""Times bj_copy_perm_stage2 (copy_perm_rational + prefix / scan kernels) at the bench shape: 92 columns x 2^22 rows, chunks of 8.
    [BOOJUM_HIP_LIB=variant.so] python tools/copy_perm_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import era_boojum_amd as E

log_n, V, q = 22, 92, 8
n = 1 << log_n
ctx = E.Context(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
vars_ = torch.randint(0, 2**63 - 1, (V, n), dtype=torch.int64, device=dev, generator=g)
sig = torch.randint(0, 2**63 - 1, (V, n), dtype=torch.int64, device=dev, generator=g)
n_chunks = (V + q - 1) // q
z = torch.empty((2, n), dtype=torch.int64, device=dev)
part = torch.empty((n_chunks - 1, 2, n), dtype=torch.int64, device=dev)
nr = np.arange(1, V + 1, dtype=np.uint64) * np.uint64(7)
run = lambda: ctx.copy_perm_stage2(vars_.data_ptr(), n, sig.data_ptr(), n, nr, V, q, log_n, (123456789, 987654321), (55555, 66666), z.data_ptr(), part.data_ptr())
run(); torch.cuda.synchronize()
ctx.timer_start()
for _ in range(5):
    run()
ms = ctx.timer_stop_ms() / 5
print("copy_perm_stage2 %d x 2^%d: %.3f ms   checksum %x" % (V, log_n, ms, int(z.view(-1)[::4099].sum().item()) & 0xFFFFFFFFFFFF))
