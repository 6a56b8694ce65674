#!/usr/bin/env python3
"""How many OpenMP threads should the oracle use on this host?  (The GPU boxes show 256 CPUs in the affinity mask and deliver a cgroup
quota of 16 cores.)  Times the coset-streaming restatement at 2^<log_n> rows for a few thread counts.  Host-only.
    python tools/oracle_threads_probe.py 20 16 32 64"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from era_boojum_amd import sha256_circuit as S  # noqa: E402
from oracle import prover_streaming as PS  # noqa: E402

log_n = int(sys.argv[1])
c = S.sha256_circuit(S.bench_message(S.message_len_for_log_n(log_n)))
cap = np.zeros((16, 4), dtype=np.uint64)
for t in [int(x) for x in sys.argv[2:]]:
    t0 = time.time()
    PS.commitments_and_openings(c, cap, 8, 16, threads=t, check_setup_cosets=(0, 5), rest_of_the_proof=True)
    print("threads %d: %.1f s  (OMP_WAIT_POLICY=%s)" % (t, time.time() - t0, os.environ.get("OMP_WAIT_POLICY", "default")), flush=True)
