"""The boundary is a C ABI: examples/host_example.c is plain C99 built with gcc against include/boojum_hip.h and the shared
library only (no HIP headers, no C++), the way a cgo / Rust-FFI binding consumes it.  Without a GPU it must refuse to run
(exit code 2: no CPU fallback); on a GPU it commits to a batch of columns and checks NTT round trip, a Merkle path and the
transcript through the ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="host_example"):
    from era_boojum_amd import build
    build.build()
    exe = os.path.join(str(tmp_path), name)
    libdir = os.path.join(ROOT, "era_boojum_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L", libdir, "-lboojum_hip", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_host_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    import era_boojum_amd as E
    exe = _build(tmp_path)
    if E.load_library().bj_device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_c_host_commits_a_batch_of_columns(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "round trip" in r.stdout and "reaches cap" in r.stdout and r.stdout.strip().endswith("ok")


def test_c_sharded_host_builds_and_refuses_to_run_without_a_gpu(tmp_path):
    import era_boojum_amd as E
    exe = _build(tmp_path, "host_sharded")
    if E.load_library().bj_device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_c_host_proves_through_the_sharded_entry_point_with_the_rccl_transport(tmp_path):
    """examples/host_sharded.c: a whole proof from C — circuit arrays, bj_rccl_unique_id / bj_comm_rccl_create,
    bj_setup_create_sharded, bj_prove, the pipelined bj_prove_async / bj_proof_wait loop — as rank 0 of a world of 1 (one GPU here; `host_sharded <rank> <world> <id file>` is the
    same program on every GPU of a node).  It checks the proof against the unsharded entry point and that a broken witness is
    refused."""
    exe = _build(tmp_path, "host_sharded")
    r = subprocess.run([exe, "0", "1", os.path.join(str(tmp_path), "rccl.id")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fingerprint" in r.stdout and "identical proof" in r.stdout and r.stdout.strip().endswith("ok")
    assert "pipelined: 5 proofs" in r.stdout
