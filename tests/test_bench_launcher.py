"""bench.py's launcher logic on a box without GPUs: `--gpus N` never degrades to a single-GPU run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_over):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID", "BJ_BENCH_BACKEND"):
        env.pop(k, None)
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_more_gpus_than_visible_is_refused_before_anything_runs():
    import torch
    want = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 2
    r = _run(["--gpus", str(want), "--log-n", "14", "--steps", "1"])
    assert r.returncode == 2 and "refusing" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_world_size_of_the_launcher_must_match_gpus():
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0


def test_a_rank_that_dies_after_rendezvous_ends_the_run_with_an_error():
    """`bench.py --gpus 2` as the driver launches it for N > 1 (torch.distributed.run, here over gloo on the CPU): when one rank
    dies right after the rendezvous, the launcher tears the other one down and the command exits non-zero well inside the
    watchdog window — it neither hangs in a collective nor prints a JSON line for a run that did not happen."""
    import socket
    import time
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BJ_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log-n", "14", "--steps", "1", "--test-die-rank", "1"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_circuit_cache_round_trip():
    """save_circuit / load_circuit (the one-synthesis-per-node path of `bench.py --gpus N`): same circuit, arrays memory-mapped."""
    import tempfile
    import numpy as np
    from era_boojum_amd import synthetic as S
    c = S.sha_shaped_circuit(8, seed=1, table_bits=2)
    d = os.path.join(tempfile.mkdtemp(), "cache")
    S.save_circuit(c, d, "note")
    back, note = S.load_circuit(d)
    assert note == "note" and isinstance(back.variables, np.memmap) and np.array_equal(back.sigmas, c.sigmas)
    assert back.num_vars == c.num_vars and [g.name for g in back.gates] == [g.name for g in c.gates] and S.check_satisfied(back)
