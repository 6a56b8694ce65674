#!/usr/bin/env python3
"""One rank of the W-rank sharded 2^22-row proof alone on this GPU, peers replayed (era_boojum_amd/scale_replay.py), for profilers:
    rocprofv3 --kernel-trace --stats -- python tools/replay_rank.py <world> <rank> <steps> [log_n]
prints the JSON of scale_replay.measure for that rank."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from era_boojum_amd import scale_replay, sha256_circuit as SHA  # noqa: E402

world, rank, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
log_n = int(sys.argv[4]) if len(sys.argv) > 4 else 22
circuit = SHA.sha256_circuit(SHA.bench_message(SHA.message_len_for_log_n(log_n), seed=42))
print(json.dumps(scale_replay.measure(circuit, world, steps=steps, warmup=1, ranks=[rank])))
