"""A/B of the two-pass plan (ntt_front10 + ntt_local12) against the three-pass plan (first4 + strided8 + local10) at 2^22 on the GPU:
results must be identical, then both are timed.  python tools/ntt_two_pass_ab.py [n_cols]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import era_boojum_amd as E

P = E.P
log_n, L = 22, 8
n = 1 << log_n
n_cols = int(sys.argv[1]) if len(sys.argv) > 1 else 93
lib = E.load_library()
ctx = E.Context(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
mono = torch.randint(0, 2**63 - 1, (n_cols, n), dtype=torch.int64, device=dev, generator=g)
lde = torch.empty((n_cols, L, n), dtype=torch.int64, device=dev)
tmp = torch.empty((n_cols, n), dtype=torch.int64, device=dev)


def plan(two):
    os.environ["BJ_NTT_TWO_PASS"] = "1" if two else "0"
    lib.bj_env_reload()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


res = {}
for two in (False, True):
    plan(two)
    ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, n_cols, 3)
    torch.cuda.synchronize()
    res[("lde", two)] = lde[: min(n_cols, 8)].clone()
    ctx.intt_batch(mono.data_ptr(), tmp.data_ptr(), log_n, n_cols)
    torch.cuda.synchronize()
    res[("intt", two)] = tmp[: min(n_cols, 8)].clone()
    ctx.ntt_forward_batch(mono.data_ptr(), tmp.data_ptr(), log_n, n_cols, coset=7)
    torch.cuda.synchronize()
    res[("fwd7", two)] = tmp[: min(n_cols, 8)].clone()
    ctx.lde_cosets_batch(mono.data_ptr(), lde.data_ptr(), log_n, n_cols, 3, 5, 1)
    torch.cuda.synchronize()
    res[("lde1", two)] = lde.view(-1)[: min(n_cols, 8) * n].clone()
for k in ("lde", "intt", "fwd7", "lde1"):
    same = bool(torch.equal(res[(k, False)], res[(k, True)]))
    print("%s: two-pass == three-pass: %s" % (k, same))
    assert same or os.environ.get("BJ_AB_NOCHECK"), k
for two in (False, True):
    plan(two)
    t_lde = timed(lambda: ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, n_cols, 3))
    t_intt = timed(lambda: ctx.intt_batch(mono.data_ptr(), tmp.data_ptr(), log_n, n_cols))
    t_l1 = timed(lambda: ctx.lde_cosets_batch(mono.data_ptr(), lde.data_ptr(), log_n, n_cols, 3, 5, 1))
    lb = 8.0 * n * (1 + L) * n_cols
    print("%s: LDE %d x 2^22 x 8: %.3f ms (%.1f GB/s, frac %.4f)   iNTT: %.3f ms   one coset of eight: %.3f ms"
          % ("two-pass  " if two else "three-pass", n_cols, t_lde, lb / t_lde / 1e6, lb / t_lde / 1e6 / 8000, t_intt, t_l1))
# round 6: the same two-pass transforms through the TILED monomial layout (no bit-reversal pass behind the inverse transform)
plan(True)
if ctx.monomials_tiled(log_n):
    tl = torch.empty((n_cols, n), dtype=torch.int64, device=dev)
    ctx.intt_batch(mono.data_ptr(), tmp.data_ptr(), log_n, n_cols)
    ctx.intt_batch_tiled(mono.data_ptr(), tl.data_ptr(), log_n, n_cols)
    back = torch.empty_like(tl)
    ctx.tiled_permute_batch(tl.data_ptr(), back.data_ptr(), log_n, n_cols, to_tiled=False)
    torch.cuda.synchronize()
    print("intt tiled == intt natural (after re-layout):", bool(torch.equal(back, tmp)))
    ctx.lde_batch(tmp.data_ptr(), lde.data_ptr(), log_n, n_cols, 3)
    torch.cuda.synchronize()
    want = lde[: min(n_cols, 4)].clone()
    ctx.lde_cosets_batch_tiled(tl.data_ptr(), lde.data_ptr(), log_n, n_cols, 3, 0, 8)
    torch.cuda.synchronize()
    print("lde from tiled == lde from natural:", bool(torch.equal(lde[: min(n_cols, 4)], want)))
    t_i = timed(lambda: ctx.intt_batch_tiled(mono.data_ptr(), tl.data_ptr(), log_n, n_cols))
    t_n = timed(lambda: ctx.intt_batch(mono.data_ptr(), tmp.data_ptr(), log_n, n_cols))
    t_l = timed(lambda: ctx.lde_cosets_batch_tiled(tl.data_ptr(), lde.data_ptr(), log_n, n_cols, 3, 0, 8))
    t_ln = timed(lambda: ctx.lde_batch(tmp.data_ptr(), lde.data_ptr(), log_n, n_cols, 3))
    t_p = timed(lambda: ctx.tiled_permute_batch(tl.data_ptr(), back.data_ptr(), log_n, n_cols, to_tiled=False))
    print("tiled     : iNTT %.3f ms (natural %.3f)   LDE %.3f ms (natural %.3f)   re-layout pass %.3f ms" % (t_i, t_n, t_l, t_ln, t_p))
    del tl, back
sys.stdout.flush()
del mono, lde, tmp, res
torch.cuda.synchronize()
ctx.close()
