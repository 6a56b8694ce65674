import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mode = sys.argv[1]
import era_boojum_amd as E
from era_boojum_amd import synthetic as S
c = S.sha_shaped_circuit(20, seed=42, table_bits=4)
if mode != "plain":
    import torch
    torch.cuda.set_device(0)
ctx = E.Context(0)
if mode == "torch_stream":
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    print("stream handle", torch.cuda.current_stream().cuda_stream)
setup = E.ProverSetup(ctx, c, 8, 16, 100)
if mode in ("torch_mem", "torch_stream"):
    dv = torch.from_numpy(c.variables.view(np.int64)).to("cuda:0"); dm = torch.from_numpy(c.multiplicities.view(np.int64)).to("cuda:0")
    pv, pm = dv.data_ptr(), dm.data_ptr()
else:
    pv, pm = ctx.upload(c.variables), ctx.upload(c.multiplicities)
for i in range(3):
    t0 = time.time(); buf, st = setup.prove_dev(pv, pm); print(mode, round((time.time()-t0)*1e3,1), {k: round(v,1) for k,v in st.items()})
