"""The real SHA-256 bench circuit (era_boojum_amd/sha256_circuit.py): what the reference's synthesis must
satisfy.  The Rust constraint system cannot run here, so the layout is pinned by its invariants: the digest wired out of
the circuit is SHA-256 of the message, every gate / lookup / copy constraint holds, rows follow the placement rules, the
8 KiB bench message lands on 2^16 rows (BASELINE config 1), and the oracle prover's proof of it is accepted by the
verifier restatement."""
import hashlib

import numpy as np
import pytest

from era_boojum_amd import field_np as F
from era_boojum_amd import sha256_circuit as S
from era_boojum_amd.synthetic import check_satisfied


@pytest.mark.parametrize("length", [1, 55, 56, 64, 119, 120, 200])
def test_digest_and_satisfiability(length):
    """Padding edge cases (sha256/mod.rs:39-63): one or two padding blocks, words mixing data and constants."""
    msg = bytes((7 * i + 3) & 255 for i in range(length))
    c, info = S.sha256_circuit(msg, return_info=True)
    assert info["digest"] == hashlib.sha256(msg).digest()
    assert info["num_blocks"] == len(msg) // 64 + (1 if len(msg) % 64 <= 55 else 2)
    assert check_satisfied(c)
    assert c.public_inputs == [] and c.quotient_degree == 4 and c.num_constant_cols == 8


def test_rows_follow_the_placement_rules():
    msg = S.bench_message(300)
    c, info = S.sha256_circuit(msg, return_info=True)
    n, gp_rows, lk_rows = c.n, info["gp_rows"], info["lookup_rows"]
    path_cols = max(len(g.path) for g in c.gates)
    by_name = {g.name: g for g in c.gates}
    sel = {}
    for g in c.gates:
        m = np.ones(n, dtype=bool)
        for i, bit in enumerate(g.path):
            m &= c.constants[i] == (1 if bit else 0)
        sel[g.name] = m
    # rows [0, gp_rows) hold gates, the rest are NopGate rows (setup.rs:336-341)
    assert not sel["NopGate"][:gp_rows].any() and sel["NopGate"][gp_rows:].all()
    # instance counts: rows of a (type, constants) are full except its last one
    fma = by_name["FmaGateInBaseFieldWithoutConstant"]
    d = len(fma.path)
    rows = np.flatnonzero(sel[fma.name])
    keys = {}
    for r in rows:
        keys.setdefault((int(c.constants[d, r]), int(c.constants[d + 1, r])), []).append(r)
    assert set(keys) == {(1 << 4, 1), (1 << 16, 1), (1 << 32, 1), (1, 1)}
    for k, rs in keys.items():
        total = info["gate_instances"]["fma[%d, %d]" % k]
        assert len(rs) == -(-total // 15)
    # lookups: 8 per row, table id in the constant column, padding with row 1 of table 1 = (0, 0, 1, 1)
    assert (c.constants[c.table_id_col] >= 1).all() and (c.constants[c.table_id_col] <= 5).all()
    assert (c.constants[c.table_id_col, lk_rows:] == 1).all()
    pad = c.variables[c.num_gp_vars:, n - 1].reshape(8, 4)
    assert (pad == np.array([0, 0, 1, 1], dtype=np.uint64)).all()
    assert int(c.multiplicities.sum()) == 8 * n
    assert sum(info["lookups"].values()) + sum((8 - v % 8) % 8 for v in info["lookups"].values()) == 8 * lk_rows


def test_bench_message_of_8_kib_takes_2_pow_16_rows():
    """BASELINE config 1: `prove_sha256(8 * (1 << 10))` is a 2^16-row circuit (sha256/mod.rs:294)."""
    c, info = S.sha256_circuit(S.bench_message(8 << 10), return_info=True)
    assert c.log_n == 16
    assert info["lookup_rows"] > info["gp_rows"] > (1 << 15)      # the lookup columns are the fuller ones
    assert S.message_len_for_log_n(16) == 8640                    # 135 data blocks still fit, 136 do not
    assert S.sha256_circuit(S.bench_message(8640 + 64)).log_n == 17


def test_replayed_blocks_equal_blocks_traced_one_by_one(monkeypatch):
    """The vectorised replay of the data blocks changes nothing: same circuit as tracing every block on its own."""
    msg = S.bench_message(64 * 5 + 17)
    fast = S.sha256_circuit(msg)
    monkeypatch.setattr(S, "REPLAY_BLOCKS", False)
    slow = S.sha256_circuit(msg)
    for name in ("variables", "sigmas", "constants", "multiplicities", "tables"):
        assert np.array_equal(getattr(fast, name), getattr(slow, name)), name


def test_sigma_numpy_fallback_equals_native():
    if F._NATIVE is None:
        pytest.skip("libsynth_host.so not built")
    msg = S.bench_message(130)
    a = S.sha256_circuit(msg)
    native, F._NATIVE = F._NATIVE, None
    try:
        b = S.sha256_circuit(msg)
    finally:
        F._NATIVE = native
    assert np.array_equal(a.sigmas, b.sigmas) and np.array_equal(a.variables, b.variables)


def test_oracle_proof_of_real_sha256_is_accepted():
    from oracle import prover as OP, verifier as OV
    c = S.sha256_circuit(b"boojum on MI355X")
    setup = OP.Setup(c, 8, 16, threads=8)
    proof = OP.prove(c, setup, 8, 16, security_level=30, threads=8)
    assert OV.verify(OV.VerificationKey(c, setup.cap, 8, 16), proof)
