import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import era_boojum_amd as E
from era_boojum_amd import synthetic as S, proof_format
from oracle import verifier as OV
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
only_lists = len(sys.argv) > 2 and sys.argv[2] == "oplists"      # EVERY evaluator as the list the reference's gpu_synthesizer would capture
t = time.time(); c = S.recursion_like_circuit(log_n, seed=4, table_bits=2, poseidon2_as_op_list=only_lists); print("gen %.1f s" % (time.time() - t))
if only_lists:
    from era_boojum_amd import gate_program as GP
    for g in c.gates:
        prog = {"ConstantsAllocatorGate": GP.constants_allocator_program, "FmaGateInBaseFieldWithoutConstant": GP.fma_program,
                "ReductionGate<4>": GP.reduction4_program}.get(g.name)
        if prog:
            g.program, g.kind = prog(), S.GATE_PROGRAM
    print("gates:", [(g.name, g.kind) for g in c.gates])
import torch; torch.cuda.init()
ctx = E.Context(0)
for fri, cap in ((2, 32), (8, 16)):
    setup = E.ProverSetup(ctx, c, fri, cap, 100)
    d_vars, d_mult = ctx.upload(c.variables), ctx.upload(c.multiplicities)
    best = 1e9
    for r in range(4):
        ctx.sync(); t = time.time(); buf, st = setup.prove_dev(d_vars, d_mult); dt = time.time() - t
        if r: best = min(best, dt)
    ok = OV.verify(OV.VerificationKey(c, setup.cap(), fri, cap), proof_format.parse(buf, security_level=100))
    print("recursion-class circuit 2^%d x 155 cols, LDE %d cap %d: %.2f ms, verifier accepts: %s, stages %s" % (log_n, fri, cap, best * 1e3, ok, {k: round(v, 2) for k, v in st.items()}))
    setup.close()
