#!/usr/bin/env python3
"""Does a one-absorption leaf kernel (8 columns: the quotient oracle's shape) reach the permutation rate of the wide one when
it does not follow memory-heavy NTT kernels?  Builds the 2^25-leaf tree over 8 columns several times back to back, then once
more right after an LDE, and prints the leaf-kernel durations (run under rocprofv3 --kernel-trace for per-launch times, or
read the event times printed here).
    python tools/narrow_leaf_rate.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import era_boojum_amd as E


def main():
    dev = torch.device("cuda", 0)
    ctx = E.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    log_n, L, C = 22, 8, 8
    n = 1 << log_n
    g = torch.Generator(device=dev); g.manual_seed(1)
    mono = (torch.randint(0, 0xFFFFFFFF, (C, n), dtype=torch.int64, device=dev, generator=g) << 32) | \
        torch.randint(0, 1 << 32, (C, n), dtype=torch.int64, device=dev, generator=g)
    lde = torch.empty((C, L, n), dtype=torch.int64, device=dev)
    ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, C, 3)
    leaves = L * n
    tree = torch.empty((ctx.merkle_tree_digests(leaves, 16), 4), dtype=torch.int64, device=dev)
    out = {}

    def tree_ms():
        ctx.timer_start()
        ctx.merkle_tree_build(lde.data_ptr(), leaves, C, leaves, 16, tree.data_ptr())
        return ctx.timer_stop_ms()

    torch.cuda.synchronize()
    out["tree_ms_back_to_back"] = [round(tree_ms(), 3) for _ in range(6)]
    after = []
    for _ in range(3):   # an LDE of 32 columns' worth of work first (the proof's situation), then the tree
        for _ in range(4):
            ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, C, 3)
        after.append(round(tree_ms(), 3))
    out["tree_ms_right_after_ldes"] = after
    perms = leaves + leaves - 16
    out["Gperm_per_s_back_to_back"] = round(perms / min(out["tree_ms_back_to_back"]) / 1e6, 3)
    out["Gperm_per_s_after_ldes"] = round(perms / min(after) / 1e6, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
