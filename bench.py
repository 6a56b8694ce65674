#!/usr/bin/env python3
"""bench.py — headline measurement of the MI355X-native Boojum hot path.

Workload at N = 1 (BASELINE.json configs[1], "cfg2"): forward Goldilocks NTT, n = 2^20, 256 columns, natural ->
bit-reversed, LDE coset shift 7, u64 data resident in HBM when the timed region starts.  One "step" = one pass of the
batched NTT over the 256 columns.  With N > 1 every rank transforms its own 256 columns (independent polynomial
columns shard across GPUs with no data-path collective): weak scaling.

Printed JSON (one line, rank 0): see the driver contract.  `value` = algorithmic bytes (16 B per element: 8 read +
8 written) of ALL ranks / wall time of the timed region.  `roofline` prices the NTT kernels against the
HBM roofline using the per-step GPU time measured with HIP events on the stream the kernels run on.
`cpu_baseline` = the C oracle (restated reference CPU algorithm, one polynomial per thread like the reference's
Worker policy) timed on this box's host cores on the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--cols", type=int, default=256)
    ap.add_argument("--coset", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import era_boojum_amd as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    log_n, n_cols = args.log_n, args.cols
    n = 1 << log_n
    dev = torch.device("cuda", local_rank)
    # synthetic input: uniform u64 below p (seed per rank), resident in HBM
    g = torch.Generator(device=dev)
    g.manual_seed(20240807 + rank)
    hi = torch.randint(0, 0xFFFFFFFF, (n_cols, n), dtype=torch.int64, device=dev, generator=g)
    lo = torch.randint(0, 1 << 32, (n_cols, n), dtype=torch.int64, device=dev, generator=g)
    src = (hi << 32) | lo          # hi < 2^32-1  =>  value < p
    del hi, lo
    dst = torch.empty_like(src)

    ctx = E.Context(local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    def step():
        ctx.ntt_forward_batch(src.data_ptr(), dst.data_ptr(), log_n, n_cols, coset=args.coset)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    gpu_ms = ctx.timer_stop_ms()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed, gpu_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, gpu_ms = float(tt[0]), float(tt[1])

    bytes_per_step_per_gpu = 16.0 * n * n_cols
    total_bytes = bytes_per_step_per_gpu * world * args.steps
    value = total_bytes / elapsed / 1e9
    kern_s = gpu_ms / 1e3 / args.steps
    achieved = bytes_per_step_per_gpu / kern_s / 1e9

    out = {
        "metric": "goldilocks_ntt_algorithmic_throughput",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "cfg2: forward Goldilocks NTT 2^%d x %d columns per GPU, natural->bit-reversed, coset %d"
                               % (log_n, n_cols, args.coset),
                   "log_n": log_n, "columns_per_gpu": n_cols, "sharding": "independent columns per rank, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel": "NTT batch = all pass kernels of one step (HIP events on the launch stream)",
                     "gpu_ms_per_step": round(kern_s * 1e3, 4),
                     "algorithmic_bytes_per_step": bytes_per_step_per_gpu},
    }

    if rank == 0 and not args.no_cpu_baseline:
        import oracle as O
        threads = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        threads = min(threads, 64)
        host = src[: min(n_cols, 4 * threads)].cpu().numpy().view(np.uint64)
        # parity spot check of what was just timed (2 columns), then the timed CPU leg
        got = dst[:2].cpu().numpy().view(np.uint64)
        want = O.fft_batch(host[:2], args.coset, threads=2)
        if not np.array_equal(got, want):
            raise SystemExit("parity failure: HIP NTT differs from the oracle")
        reps, t_cpu = 0, 0.0
        c0 = time.perf_counter()
        while t_cpu < 10.0 and reps < 50:
            O.fft_batch(host, args.coset, threads=threads)
            reps += 1
            t_cpu = time.perf_counter() - c0
        cpu_gbps = 16.0 * n * host.shape[0] * reps / t_cpu / 1e9
        out["cpu_baseline"] = {"value": round(cpu_gbps, 3), "unit": "GB/s", "cores": threads, "kind": "port",
                               "sample": "%d columns of 2^%d x %d repetitions, one polynomial per thread (C oracle, "
                                         "-O3 -march=x86-64-v3, OpenMP)" % (host.shape[0], log_n, reps)}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
