"""-m gpu: seam S3 with what the reference emits (src/gpu_synthesizer/mod.rs:210-444).  Every evaluator's capture in the
reference's own call order and numbering (tests/reference_capture.py: sparse temporaries from a process-wide counter) runs on
the device — through its build-time kernel, the kernel compiled at run time from the op list, or the interpreter — and gives
the independent formulas' terms; a proof whose circuit carries a gate the library was not built with (a host's own matrix)
equals the oracle prover's proof, with and without the run-time compiler."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import era_boojum_amd as E
from era_boojum_amd import proof_format, synthetic as S
from gpu_util import DevBuf, ctx, rand_gl, P
from oracle import prover as OP
from oracle import verifier as OV
import reference_capture as RC
import test_gate_canon as TC

pytestmark = pytest.mark.gpu


def jit_status():
    buf = C.create_string_buffer(512)
    n = E.load_library().bj_gate_jit_status(buf, 512)
    return n, buf.value.decode()


@pytest.mark.parametrize("name", sorted(TC.CAPTURES))
def test_capture_order_lists_on_the_device(name):
    thunk, nv, nc, nw = TC.CAPTURES[name]
    prog = RC.to_program_raw(thunk())                       # the reference's own sparse numbers, not even renumbered
    reps = 1 if nv > 100 else 3
    n_points = 256 if nv > 100 else 600
    rng = np.random.default_rng(nv * 31 + nc)
    var = rand_gl(rng, (max(1, nv) * reps, n_points), noncanonical=True)
    con = rand_gl(rng, (max(1, nc), n_points), noncanonical=True)
    d_var, d_con, d_out = DevBuf(var), DevBuf(con), DevBuf(nelems=reps * prog.num_terms * n_points)
    if nw:      # the stand-alone evaluator takes no witness columns: covered through a whole proof in test_gpu_gate_program.py
        with pytest.raises(E.BoojumHipError, match="witness"):
            ctx().gate_program_eval(prog, d_var.ptr, n_points, d_con.ptr, n_points, reps, nv, 0, n_points, d_out.ptr)
        return
    before = jit_status()[0]
    ctx().gate_program_eval(prog, d_var.ptr, n_points, d_con.ptr, n_points, reps, nv, 0, n_points, d_out.ptr)
    got = d_out.get((reps, prog.num_terms, n_points))
    for i in range(0, n_points, 41 if nv > 100 else 37):
        for r in range(reps):
            v = [int(x) % P for x in var[r * nv:(r + 1) * nv, i]]
            c = [int(x) % P for x in con[:nc, i]]
            assert [int(x) for x in got[r, :, i]] == TC.want_terms(name, v, c, []), (name, i, r)
    if name == "matrix_multiplication_host_matrix" and not os.environ.get("BJ_GATE_NO_JIT"):
        n, status = jit_status()
        assert n >= max(1, before), status                  # compiled (or cached) at run time, not interpreted
    for d in (d_var, d_con, d_out):
        d.free()


def _prove_host_gate_circuit():
    c = S.sha_shaped_circuit(10, seed=5, table_bits=2, gates=S.host_gates(), mix=(0.05, 0.3, 0.3, 0.2))
    gsetup = E.ProverSetup(ctx(), c, 8, 16, 30)
    buf, _ = gsetup.prove()
    cap = gsetup.cap()
    gsetup.close()
    return c, buf, cap


def test_proof_with_a_hosts_own_gate_equals_the_oracle_proof(tmp_path):
    """MatrixMultiplicationGate with a matrix only the host knows: no build-time kernel can exist.  The HIP proof (gate kernel
    compiled at bj_setup_create) equals the oracle prover's proof, the verifier restatement accepts it, and a process with
    BJ_GATE_NO_JIT=1 (the interpreter on the canonical schedule) produces the same bytes."""
    from test_gpu_prover import _compare
    c, buf, cap = _prove_host_gate_circuit()
    n, status = jit_status()
    if not os.environ.get("BJ_GATE_NO_JIT"):
        assert n >= 1, status
    osetup = OP.Setup(c, 8, 16, threads=8)
    assert np.array_equal(cap, osetup.cap)
    po = OP.prove(c, osetup, 8, 16, security_level=30, threads=8)
    pg = proof_format.parse(buf, security_level=30)
    _compare(pg, po)
    assert OV.verify(OV.VerificationKey(c, cap, 8, 16), pg, verbose=True)
    out = os.path.join(str(tmp_path), "proof.npy")
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; torch.cuda.init();"
            "import test_gpu_gate_capture as T; np.save(%r, T._prove_host_gate_circuit()[1])") % (os.path.dirname(here), here, out)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BJ_GATE_NO_JIT="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(out), buf)


def test_recursion_class_circuit_from_capture_order_lists_only():
    """The circuit of the golden proof's class with EVERY evaluator handed over as the reference would capture it (sparse
    numbering, evaluate_once order; the Poseidon2 flattened gate as its 9.6 k-relation list): each list finds its build-time
    kernel / the hand-written Poseidon2 evaluator, and the proof is the one the hand-wired kinds give, byte for byte."""
    a = S.recursion_like_circuit(10, seed=9)
    b = S.recursion_like_circuit(10, seed=9, poseidon2_as_op_list=True)
    lib = E.load_library()
    names = {"U8x4FMAGate": "u8x4_fma", "Poseidon2FlattenedGate": "poseidon2_flattened", "DotProductGate<4>": "dot_product4",
             "ZeroCheckGate": "zero_check", "UIntXAddGate": "uintx_add", "SelectionGate": "selection",
             "ParallelSelectionGate<4>": "parallel_selection4", "FmaGateInBaseFieldWithoutConstant": "fma",
             "ReductionGate<4>": "reduction4", "ConstantsAllocatorGate": "constants_allocator"}
    for g in b.gates:
        if g.name in names:
            g.program = RC.to_program_raw(TC.CAPTURES[names[g.name]][0]())
            g.kind = S.GATE_PROGRAM
            assert lib.bj_gate_program_generated(C.byref(g.program.struct)) == 1, g.name
    sa, sb = E.ProverSetup(ctx(), a, 2, 32, 30), E.ProverSetup(ctx(), b, 2, 32, 30)
    pa, _ = sa.prove()
    pb, _ = sb.prove()
    assert np.array_equal(pa, pb)
    sa.close(); sb.close()


@pytest.mark.parametrize("log_n", [14, 18])
def test_recursion_class_proof_from_capture_lists_equals_the_oracle_proof(log_n):
    """Byte identity, not only "same as the hand-wired kinds": the recursion-class circuit (155 columns, eleven gate types, the
    golden proof's FRI parameters: LDE 2, cap 32) at 2^14 and 2^18 rows with every evaluator handed over in the reference's capture order
    and sparse numbering — fused sweep of the build-time kernels, the hand-written Poseidon2 evaluator reached through the
    fingerprint of the 9.6 k-relation capture — against the oracle prover (numpy semantics of the same op lists, the compact
    Poseidon2 restatement).  ~1 minute of oracle time at 2^18."""
    from test_gpu_prover import _compare
    c = S.recursion_like_circuit(log_n, seed=21)      # the oracle evaluates the Poseidon2 gate from its compact restatement
    names = {"U8x4FMAGate": "u8x4_fma", "Poseidon2FlattenedGate": "poseidon2_flattened", "DotProductGate<4>": "dot_product4",
             "ZeroCheckGate": "zero_check", "UIntXAddGate": "uintx_add", "SelectionGate": "selection",
             "ParallelSelectionGate<4>": "parallel_selection4", "FmaGateInBaseFieldWithoutConstant": "fma",
             "ReductionGate<4>": "reduction4", "ConstantsAllocatorGate": "constants_allocator"}
    osetup = OP.Setup(c, 2, 32, threads=16)
    po = OP.prove(c, osetup, 2, 32, security_level=60, threads=16)      # the oracle takes kinds 1-3 natively, op lists by numpy
    for g in c.gates:
        if g.name in names:
            g.program = RC.to_program_raw(TC.CAPTURES[names[g.name]][0]())
            g.kind = S.GATE_PROGRAM
    gsetup = E.ProverSetup(ctx(), c, 2, 32, 60)
    assert np.array_equal(gsetup.cap(), osetup.cap)
    buf, _ = gsetup.prove()
    pg = proof_format.parse(buf, security_level=60)
    _compare(pg, po)
    gsetup.close()


def test_run_time_compiled_kernels_are_cached_on_disk(tmp_path):
    """BJ_GATE_JIT_CACHE=<dir>: the first process compiles the host's gate and leaves its code object there, the second one
    loads it instead of compiling (bj_gate_jit_status tells) — same terms both times."""
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, ctypes as C, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import torch; torch.cuda.init();"
            "import era_boojum_amd as E; from gpu_util import DevBuf, ctx; import reference_capture as RC, test_gate_canon as TC;"
            "prog = RC.to_program(TC.CAPTURES['matrix_multiplication_host_matrix'][0]());"
            "var = np.arange(24 * 64, dtype=np.uint64).reshape(24, 64) * np.uint64(0x9E3779B97F4A7C15);"
            "d_var, d_out = DevBuf(var), DevBuf(nelems=12 * 64);"
            "ctx().gate_program_eval(prog, d_var.ptr, 64, d_var.ptr, 64, 1, 24, 0, 64, d_out.ptr);"
            "buf = C.create_string_buffer(256); n = E.load_library().bj_gate_jit_status(buf, 256);"
            "print('STATUS', n, buf.value.decode()); np.save(sys.argv[1], d_out.get((12, 64)))") % (os.path.dirname(here), here)
    outs = []
    for k in range(2):
        out = os.path.join(str(tmp_path), "terms%d.npy" % k)
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, BJ_GATE_JIT_CACHE=str(tmp_path)), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((np.load(out), [l for l in r.stdout.splitlines() if l.startswith("STATUS")][0]))
    assert any(f.endswith(".hsaco") for f in os.listdir(str(tmp_path)))
    assert "1 gate kernels compiled" in outs[0][1] and "0 loaded" in outs[0][1]
    assert "0 gate kernels compiled" in outs[1][1] and "1 loaded" in outs[1][1]
    assert np.array_equal(outs[0][0], outs[1][0])
    # a cache file that does not belong to THIS build is refused and recompiled, never loaded: (a) truncated,
    # (b) a foreign / pre-header file of the right name, (c) one byte of the code object flipped, (d) header with another build key
    path = [os.path.join(str(tmp_path), f) for f in os.listdir(str(tmp_path)) if f.endswith(".hsaco")][0]
    good = open(path, "rb").read()
    assert good[:8] == b"BJJITv2\0" and int.from_bytes(good[16:24], "little") == len(good) - 32
    flipped = bytearray(good)
    flipped[len(good) // 2] ^= 0x40
    wrong_key = bytearray(good)
    wrong_key[8] ^= 1
    for k, bad in enumerate((good[:len(good) // 2], good[32:], bytes(flipped), bytes(wrong_key))):
        open(path, "wb").write(bad)
        out = os.path.join(str(tmp_path), "terms_bad%d.npy" % k)
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, BJ_GATE_JIT_CACHE=str(tmp_path)), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        status = [l for l in r.stdout.splitlines() if l.startswith("STATUS")][0]
        assert "1 gate kernels compiled" in status and "0 loaded" in status and "1 cache files refused" in status, (k, status)
        assert np.array_equal(np.load(out), outs[0][0])
        assert open(path, "rb").read() == good          # and the file was rewritten with a valid one
