#!/usr/bin/env python3
"""Secondary kernel measurements on one GPU (not the headline bench): Poseidon2 Merkle tree, FRI fold, LDE, iNTT.
Prints one JSON object per kernel with algorithmic bytes / HIP-event time.
    python tools/kernel_suite.py [--log-n 20] [--cols 93] [--log-lde 3]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import era_boojum_amd as E


def timed(ctx, fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop_ms() / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cols", type=int, default=93)
    ap.add_argument("--log-lde", type=int, default=3)
    ap.add_argument("--cap", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = E.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    n, L, C = 1 << a.log_n, 1 << a.log_lde, a.cols
    g = torch.Generator(device=dev); g.manual_seed(1)
    trace = (torch.randint(0, 0xFFFFFFFF, (C, n), dtype=torch.int64, device=dev, generator=g) << 32) | \
        torch.randint(0, 1 << 32, (C, n), dtype=torch.int64, device=dev, generator=g)
    lde = torch.empty((C, L, n), dtype=torch.int64, device=dev)
    out = []
    work = trace.clone()
    ms = timed(ctx, lambda: ctx.intt_batch(work.data_ptr(), work.data_ptr(), a.log_n, C))
    out.append({"kernel": "intt_batch", "shape": "%d x 2^%d" % (C, a.log_n), "ms": ms, "alg_GBps": 16.0 * n * C / ms / 1e6})
    ms = timed(ctx, lambda: ctx.lde_batch(work.data_ptr(), lde.data_ptr(), a.log_n, C, a.log_lde))
    out.append({"kernel": "lde_batch", "shape": "%d x 2^%d x %d cosets" % (C, a.log_n, L), "ms": ms,
                "alg_GBps": 8.0 * n * (1 + L) * C / ms / 1e6})
    leaves = L * n
    nd = ctx.merkle_tree_digests(leaves, a.cap)
    tree = torch.empty((nd, 4), dtype=torch.int64, device=dev)
    ms = timed(ctx, lambda: ctx.merkle_tree_build(lde.data_ptr(), leaves, C, leaves, a.cap, tree.data_ptr()), reps=3)
    bytes_tree = leaves * (8 * C + 32) + (leaves - a.cap) * 96
    perms = leaves * ((C + 7) // 8) + (leaves - a.cap)
    out.append({"kernel": "merkle_tree_build(poseidon2)", "shape": "2^%d leaves x %d cols" % (a.log_n + a.log_lde, C),
                "ms": ms, "alg_GBps": bytes_tree / ms / 1e6, "Gperm_per_s": perms / ms / 1e6})
    c = lde.view(-1)[: 2 * leaves].view(2, leaves)
    o = torch.empty((2, leaves // 8), dtype=torch.int64, device=dev)
    ms = timed(ctx, lambda: ctx.fri_fold_step(c[0].data_ptr(), c[1].data_ptr(), leaves, 3, o[0].data_ptr(), o[1].data_ptr(),
                                              a.log_n + a.log_lde, 123456789, (5, 7)))
    out.append({"kernel": "fri_fold_step(k=3)", "shape": "2^%d ext" % (a.log_n + a.log_lde), "ms": ms,
                "alg_GBps": (16.0 * leaves + 16.0 * leaves / 8) / ms / 1e6})
    for r in out:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))


if __name__ == "__main__":
    main()
