#!/usr/bin/env python3
"""Static instruction classes of bj::poseidon2_leaves_kernel in the gfx950 code object (no GPU needed): compiles
csrc/poseidon2.hip with -save-temps in a scratch directory and counts the kernel's instructions by class.
    python tools/leaf_isa_classes.py > profiles/r<NN>_leaf_kernel_isa_classes.json"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from era_boojum_amd import build as B

with tempfile.TemporaryDirectory() as d:
    subprocess.check_call([B.HIPCC] + B.FLAGS + ["-save-temps", "-c", os.path.join(B.CSRC, "poseidon2.hip"), "-o", os.path.join(d, "p2.o")],
                          cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(d, asm)).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2bj23poseidon2_leaves_kernelE"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
ins = []
for l in lines[start + 1:end + 1]:
    t = l.split(";")[0].strip()
    if t and not t.startswith((".", "//", "#")) and not t.endswith(":"):
        ins.append(t.split()[0])
c = collections.Counter(ins)
valu = {k: v for k, v in c.items() if k.startswith("v_")}
classes = {"v_mad_u64_u32": c["v_mad_u64_u32"], "v_lshl_add_u64": c["v_lshl_add_u64"],
           "carry-writing 32-bit (v_add_co / v_addc_co / v_sub_co / v_subb_co / v_subbrev_co)":
               sum(v for k, v in c.items() if re.match(r"v_(add|addc|sub|subb|subbrev)_co_u32", k)),
           "v_cndmask_b32": c["v_cndmask_b32"], "v_mov_b32": c["v_mov_b32"]}
classes["other VALU"] = sum(valu.values()) - sum(classes.values())
print(json.dumps({
    "kernel": "bj::poseidon2_leaves_kernel (static instruction counts of the gfx950 code object, hipcc -O3 -save-temps; tools/leaf_isa_classes.py)",
    "total_instructions": len(ins), "VALU": sum(valu.values()),
    "SALU": sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_endpgm"))),
    "s_cbranch + s_branch": sum(v for k, v in c.items() if k.startswith(("s_cbranch", "s_branch"))), "s_nop": c["s_nop"],
    "SMEM (s_load)": sum(v for k, v in c.items() if k.startswith("s_load")),
    "VMEM (global_load / global_store)": sum(v for k, v in c.items() if k.startswith("global_")),
    "VALU_classes": classes,
    "note": "the permutation is one generated stream (tools/gen_p2_asm.py) inside loops (4 full rounds, 22 partial rounds, 4 full rounds); the out-of-line "
            "stubs of the rare carries are counted too (most s_nop live there); the dynamic count per permutation is in the PMC summary of the same round"}, indent=1))
