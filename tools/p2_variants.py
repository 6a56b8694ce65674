#!/usr/bin/env python3
"""A/B builds of the Poseidon2 instruction stream: the library relinked around poseidon2.hip compiled with another schedule
(tools/gen_p2_asm.py under BJ_P2_WAYS / BJ_P2_COMBINE), into exp/libbj_p2_<name>.so; BOOJUM_HIP_LIB selects one at load.
    python tools/p2_variants.py build            # on the build host (hipcc cross-compiles)
    python tools/p2_variants.py bench            # on the GPU: tree 2^23 x 93 with every variant, JSON lines
A list of variant names after the command restricts both to those."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXP = os.path.join(ROOT, "exp")
VARIANTS = {"w2": {"BJ_P2_WAYS": "2"}, "w3": {"BJ_P2_WAYS": "3"}, "r3": {"BJ_P2_WAYS": "3", "BJ_P2_ZERO_HOIST": "0", "BJ_P2_LATE_CONST": "0"}, "w4": {"BJ_P2_WAYS": "4"},
            "w2_inline": {"BJ_P2_WAYS": "2", "BJ_P2_COMBINE": "inline"}, "w3_inline": {"BJ_P2_WAYS": "3", "BJ_P2_COMBINE": "inline"},
            "w3_hoist": {"BJ_P2_WAYS": "3", "BJ_P2_ZERO_HOIST": "1", "BJ_P2_LATE_CONST": "0"},
            "w3_hoist_lc": {"BJ_P2_WAYS": "3", "BJ_P2_ZERO_HOIST": "1", "BJ_P2_LATE_CONST": "1"}}
if len(sys.argv) > 2:   # python tools/p2_variants.py build|bench name [name ...]
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[2:]}


def build():
    from era_boojum_amd import build as B
    B.build()
    os.makedirs(EXP, exist_ok=True)
    objs = [os.path.join(B.HERE, "build", s.replace(".hip", ".o").replace(".cpp", ".o")) for s in B.SOURCES if s != "poseidon2.hip"]
    for name, env in VARIANTS.items():
        inc = os.path.join(EXP, "p2_asm_%s.inc" % name)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_p2_asm.py"), inc], env=dict(os.environ, **env))
        obj = os.path.join(EXP, "poseidon2_%s.o" % name)
        subprocess.check_call([B.HIPCC] + B.FLAGS + ['-DBJ_P2_ASM_INC="%s"' % inc, "-c", os.path.join(B.CSRC, "poseidon2.hip"), "-o", obj])
        lib = os.path.join(EXP, "libbj_p2_%s.so" % name)
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + objs + ["-ldl"])
        print(lib)


def bench():
    for name in VARIANTS:
        lib = os.path.join(EXP, "libbj_p2_%s.so" % name)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_suite.py"), "--log-n", "20", "--cols", "93"],
                           env=dict(os.environ, BOOJUM_HIP_LIB=lib), capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if "merkle" in line:
                d = json.loads(line)
                print(json.dumps({"variant": name, "tree_ms": d["ms"], "Gperm_per_s": d["Gperm_per_s"]}), flush=True)
        if r.returncode:
            print(json.dumps({"variant": name, "error": r.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    {"build": build, "bench": bench}[sys.argv[1]]()
