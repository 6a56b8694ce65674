"""tools/scale_model.py: the predicted multi-GPU times are arithmetic on a committed single-GPU profile — keep it
runnable and its classes sane (every big kernel of the profile is classed, the prediction at W = 1 is the measured wall time)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scale_model_runs_on_the_committed_profile():
    prof = os.path.join(ROOT, "profiles", "r04_prover_2p22_kernel_stats.csv")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_model.py"), prof, "--wall-ms", "261.2"], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert abs(d["T"]["1"]["ms"] - 261.2) < 0.05
    assert 15.0 < d["replicated_ms"] < 25.0 and d["sharded_ms"] > 10 * d["replicated_ms"]
    t = [d["T"][w]["ms"] for w in ("1", "2", "4", "8")]
    assert t[0] > t[1] > t[2] > t[3] > d["replicated_ms"]
    assert d["T"]["8"]["efficiency"] < d["T"]["2"]["efficiency"] <= 1.0


def test_scale_model_against_the_replayed_ranks_of_round_5():
    """A REGRESSION PIN, not a validation: the model's per-proof constant was fitted to this very file (profiles/
    r05_bench_scale_replay.json, what one rank alone measures); the test only says the fit still reproduces it within 3 % at
    W = 2, 4, 8 after a change to tools/scale_model.py.  Nothing on more than one GPU has been measured."""
    prof = os.path.join(ROOT, "profiles", "r04_prover_2p22_kernel_stats.csv")
    line = os.path.join(ROOT, "profiles", "r05_bench_scale_replay.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_model.py"), prof, "--wall-ms", "261.7", "--replay", line],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    for w in ("2", "4", "8"):
        assert 0.97 < d["T"][w]["model_over_measured"] < 1.03, (w, d["T"][w])
