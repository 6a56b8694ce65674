#!/usr/bin/env python3
"""cfg2 timing alone: forward NTT 2^20 x 256 columns, coset 7 (HIP events on the launch stream); also 2^22 x 32 x 8 LDE."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import era_boojum_amd as E
dev = torch.device("cuda", 0)
ctx = E.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
nlog, ncols = 20, 256
src = torch.randint(0, 1 << 62, (ncols, 1 << nlog), dtype=torch.int64, device=dev)
dst = torch.empty_like(src)
for _ in range(2):
    ctx.ntt_forward_batch(src.data_ptr(), dst.data_ptr(), nlog, ncols, coset=7)
ctx.timer_start()
for _ in range(10):
    ctx.ntt_forward_batch(src.data_ptr(), dst.data_ptr(), nlog, ncols, coset=7)
ms = ctx.timer_stop_ms() / 10
out = {"cfg2_ms": round(ms, 4), "cfg2_frac": round(16.0 * (1 << nlog) * ncols / ms / 1e6 / 8000.0, 4)}
del src, dst
if "--cfg2-only" in sys.argv:
    print(json.dumps(out))
    sys.exit(0)
mono = torch.randint(0, 1 << 62, (32, 1 << 22), dtype=torch.int64, device=dev)
lde = torch.empty((32, 8, 1 << 22), dtype=torch.int64, device=dev)
ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), 22, 32, 3)
ctx.timer_start()
for _ in range(3):
    ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), 22, 32, 3)
out["lde_32x2p22x8_ms"] = round(ctx.timer_stop_ms() / 3, 3)
print(json.dumps(out))
