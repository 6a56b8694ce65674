#!/usr/bin/env python3
"""Times the DEEP stage's building blocks at the 2^23 shapes in isolation (each step printed as it finishes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import era_boojum_amd as E
dev = torch.device("cuda", 0)
ctx = E.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n, N = 1 << log_n, 1 << (log_n + 3)
def T(name, fn):
    torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); print("%-50s %10.2f ms" % (name, (time.time() - t) * 1e3), flush=True)
mono = torch.randint(0, 1 << 62, (2, n), dtype=torch.int64, device=dev)
lde = torch.empty((2, N), dtype=torch.int64, device=dev)
T("lde 2 cols (first call: twiddles)", lambda: ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, 2, 3))
T("lde 2 cols", lambda: ctx.lde_batch(mono.data_ptr(), lde.data_ptr(), log_n, 2, 3))
dst = torch.zeros((2, N), dtype=torch.int64, device=dev)
src1 = [(lde[0].data_ptr(), lde[1].data_ptr())]
T("deep 1 ext source (first)", lambda: ctx.deep_quotient_accumulate(src1, [(3, 4)], [(5, 6)], (7, 8), log_n, 3, dst[0].data_ptr(), dst[1].data_ptr(), False))
T("deep 1 ext source", lambda: ctx.deep_quotient_accumulate(src1, [(3, 4)], [(5, 6)], (7, 8), log_n, 3, dst[0].data_ptr(), dst[1].data_ptr(), True))
src9 = [(lde[i % 2].data_ptr(), None) for i in range(9)]
T("deep 9 base sources", lambda: ctx.deep_quotient_accumulate(src9, [(i, 0) for i in range(9)], [(i + 1, i + 2) for i in range(9)], (0, 0), log_n, 3, dst[0].data_ptr(), dst[1].data_ptr(), True))
w = torch.empty((2, n), dtype=torch.int64, device=dev)
T("barycentric weights", lambda: ctx.barycentric_weights(log_n, 7, (11, 13), w[0].data_ptr(), w[1].data_ptr()))
T("barycentric eval 4 cols", lambda: ctx.barycentric_eval_batch([lde[0].data_ptr(), lde[1].data_ptr(), mono[0].data_ptr(), mono[1].data_ptr()], log_n, w[0].data_ptr(), w[1].data_ptr()))
T("intt 2 cols", lambda: ctx.intt_batch(mono.data_ptr(), mono.data_ptr(), log_n, 2))
print("done")
