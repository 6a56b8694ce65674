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
