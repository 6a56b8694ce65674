"""Build libboojum_hip.so (all HIP kernels + the C-ABI layer) in-tree with hipcc for gfx950.

    python -m era_boojum_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libboojum_hip.so")
SOURCES = ["abi.hip", "ntt.hip", "ntt_r16.hip", "poseidon2.hip", "blake2s.hip", "keccak.hip", "fri.hip", "fri_prover.hip", "comm_rccl.hip", "openings.hip", "openings_abi.hip", "stage_ops_abi.hip", "stage2.hip", "quotient.hip", "gate_program.hip", "gate_aot.hip", "gate_poseidon2.hip", "prover.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


SYNTH_SRC = os.path.join(HERE, "host", "synth_host.c")
SYNTH_LIB = os.path.join(HERE, "libsynth_host.so")


def build_synth(force=False):
    """gcc build of the host-side input-synthesis helper (not on the proving path; field_np.py falls back to numpy)."""
    if not force and os.path.exists(SYNTH_LIB) and os.path.getmtime(SYNTH_LIB) >= os.path.getmtime(SYNTH_SRC):
        return SYNTH_LIB
    subprocess.check_call([os.environ.get("CC", "gcc"), "-O3", "-march=x86-64-v2", "-fopenmp", "-shared", "-fPIC", "-o", SYNTH_LIB,
                           SYNTH_SRC])
    return SYNTH_LIB


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    out.append(os.path.join(os.path.dirname(HERE), "include", "boojum_hip.h"))
    out.append(os.path.abspath(__file__))
    return out


def build_variant(out_path, extra_flags):
    """Tuning experiments: the same library built with extra -D flags into another file (BOOJUM_HIP_LIB selects it at load)."""
    bdir = os.path.join(HERE, "build", "variant_" + os.path.basename(out_path))
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_path] + objs + ["-ldl"])
    return out_path


def build(force=False, verbose=False):
    build_synth(force)
    from . import gate_codegen          # csrc/gate_aot.hip: straight-line kernels generated from the known gate op lists
    gate_codegen.write()
    import importlib.util               # csrc/gl_asm.inc: hand-scheduled lazy-arithmetic sequences (tools/gen_gl_asm.py)
    spec = importlib.util.spec_from_file_location("gen_gl_asm", os.path.join(os.path.dirname(HERE), "tools", "gen_gl_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    spec = importlib.util.spec_from_file_location("gen_p2_asm", os.path.join(os.path.dirname(HERE), "tools", "gen_p2_asm.py"))
    mod = importlib.util.module_from_spec(spec)      # csrc/p2_asm.inc: the Poseidon2 permutation as one scheduled instruction stream
    spec.loader.exec_module(mod)
    mod.main()
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(p) for p in _deps()):
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:      # python -m era_boojum_amd.build --variant out.so -DFOO -DBAR
        i = sys.argv.index("--variant")
        print(build_variant(os.path.abspath(sys.argv[i + 1]), sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
