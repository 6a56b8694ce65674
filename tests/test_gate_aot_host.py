"""The fused gate sweep (csrc/gate_aot.hip, generated) on the host: its per-point logic is __host__ __device__, so the
scheduling of repetitions over column windows and the folded selector can be checked without a GPU against the per-gate
logic, for random gate sets (tools/gate_fused_host_check.cpp).  The -m gpu tests then check the kernels themselves through
whole proofs (tests/test_gpu_gate_program.py)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_fused_gate_sweep_equals_per_gate_evaluation_on_the_host(tmp_path):
    from era_boojum_amd import gate_codegen
    gate_codegen.write()
    exe = str(tmp_path / "gate_fused_host_check")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-x", "hip", "-std=c++17", "-O1", "-DBJ_GATE_AOT_HOST_ONLY",
                           "-I" + os.path.join(ROOT, "era_boojum_amd", "csrc"), os.path.join(ROOT, "tools", "gate_fused_host_check.cpp"),
                           "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fused == per-gate" in out.stdout


def test_fused_gate_sweep_keeps_its_occupancy():
    """The fused sweep's register file is that of its heaviest body: a body that does not belong there (a 12 x 12 matrix gate
    needs > 100 VGPRs) once halved the recursion-class quotient's speed.  gate_codegen.py admits light bodies only; the
    compiler's own resource report for the generated file must stay at 6+ waves per SIMD (<= 80 VGPRs)."""
    import os
    import re
    import subprocess
    from era_boojum_amd import build as B
    r = subprocess.run([B.HIPCC] + B.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, "gate_aot.hip"), "-o", os.devnull],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"Function Name: \S*gate_aot_fused_kernel\S*.*?\n.*?\n.*?VGPRs: (\d+)", r.stderr, re.S)
    assert m, "no resource report for the fused kernel"
    assert int(m.group(1)) <= 80, "gate_aot_fused_kernel uses %s VGPRs" % m.group(1)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_weak_residue_helpers_equal_128_bit_arithmetic_on_the_host(tmp_path):
    """gl::add_weak / sub_weak / mul7_weak / e2_mul_weak / mul_pow2 and the canonical operators (csrc/gl.h) are __host__
    __device__: their host halves are checked here on extreme and random words against unsigned __int128 arithmetic
    (tools/gl_weak_host_check.cpp); the device halves, incl. the inline-assembly product, by tests/test_gpu_field_ops.py."""
    exe = str(tmp_path / "gl_weak_host_check")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-x", "hip", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "era_boojum_amd", "csrc"),
                           os.path.join(ROOT, "tools", "gl_weak_host_check.cpp"), "-o", exe], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "weak helpers == 128-bit arithmetic" in out.stdout
