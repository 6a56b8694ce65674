"""MemcopySerializable dumps (era_boojum_amd/memcopy_format.py): byte layouts pinned by hand-assembled buffers that follow
the reference's write_into_buffer code (fast_serialization.rs, polynomial_storage.rs:77-126, witness.rs:42-71,
hints/mod.rs), and round trips of a whole prover input (the real SHA-256 circuit)."""
import struct

import numpy as np
import pytest

from era_boojum_amd import memcopy_format as M
from era_boojum_amd import sha256_circuit as SHA
from era_boojum_amd.synthetic import sha_bench_gates, check_satisfied


def q(*v):
    return b"".join(struct.pack("<Q", x) for x in v)


def test_field_vec_and_poly_vec_layout():
    assert M.write_field_vec([1, 2, 0xFFFFFFFF00000000]) == q(3, 1, 2, 0xFFFFFFFF00000000)
    assert M.write_poly_vec([[5, 6], [7, 8]]) == q(2) + q(2, 5, 6) + q(2, 7, 8)
    r = M._Reader(q(2) + q(2, 5, 6) + q(2, 7, 8))
    cols = M.read_poly_vec(r)
    assert [c.tolist() for c in cols] == [[5, 6], [7, 8]] and r.pos == 56
    with pytest.raises(ValueError, match="truncated"):
        M.read_field_vec(M._Reader(q(3, 1, 2)))


def test_witness_vec_layout():
    buf = M.write_witness_vec([(3, 5)], [10, 11, 12], [1, 0, 2])
    # public_inputs_locations: count + (usize, usize); all_values: len + words; multiplicities: count + u32 each
    assert buf == q(1) + q(3, 5) + q(3, 10, 11, 12) + q(3) + struct.pack("<III", 1, 0, 2)
    locs, vals, mult = M.read_witness_vec(buf)
    assert locs == [(3, 5)] and vals.tolist() == [10, 11, 12] and mult.tolist() == [1, 0, 2]
    with pytest.raises(ValueError, match="trailing"):
        M.read_witness_vec(buf + b"\0")


def test_copy_hint_layout_and_placeholders():
    buf = M.write_variables_hint([[0, -1], [2, 1]])
    assert buf == q(2) + q(2, 0, 1 << 63) + q(2, 2, 1)
    hint = M.read_variables_hint(buf)
    cols = M.variables_from_witness_vec(np.array([100, 101, 102], dtype=np.uint64), hint)
    assert cols.tolist() == [[100, 0], [102, 101]]                   # a placeholder cell holds 0 (witness.rs:425-432)
    assert M.multiplicity_column(np.array([4, 5], dtype=np.uint32), 4).tolist() == [[4, 5, 0, 0]]


def test_tree_node_bincode_layout():
    gates = sha_bench_gates()
    fma, nop = gates[1], gates[3]
    tree = ("fork", ("gate", fma), ("gate", nop))
    idx = {id(g): i for i, g in enumerate(gates)}
    buf = M._write_tree(tree, idx)
    # enum tags are u32; GateDescription = gate_idx, num_constants, degree (usize as u64), needs_selector, is_lookup (bool bytes)
    assert buf == struct.pack("<I", 2) + struct.pack("<I", 1) + q(1, 2, 3) + b"\x01\x00" + struct.pack("<I", 1) + q(3, 0, 0) + b"\x01\x00"
    back = M._read_tree(M._Reader(buf), gates)
    assert back[0] == "fork" and back[1][1] is fma and back[2][1] is nop
    assert M._write_tree(None, idx) == struct.pack("<I", 0)
    bad = bytearray(buf); bad[8 + 8] = 9                                # num_constants of the FMA leaf
    with pytest.raises(ValueError, match="constants"):
        M._read_tree(M._Reader(bytes(bad)), sha_bench_gates())


def test_prover_input_round_trip_of_the_sha256_circuit():
    c, info = SHA.sha256_circuit(SHA.bench_message(70), return_info=True)
    tabs = SHA.sha_tables()
    offs = np.cumsum([0] + [t.shape[0] for t in tabs])
    mult = c.multiplicities[0, :offs[-1]].astype(np.uint32)             # per-table counters, concatenated
    setup_base = M.write_setup_base(c)
    witness = M.write_witness_vec([], info["all_values"], mult)
    hint = M.write_variables_hint(info["var_ids"])
    assert len(setup_base) == 8 * 3 + (92 + 8 + 5) * (8 + 8 * c.n) + 8 + 8 + len(M._write_tree(c.selector_tree, {id(g): i for i, g in enumerate(c.gates)}))
    back = M.circuit_from_dumps(setup_base, witness, hint, sha_bench_gates(), num_gp_vars=60, lookup_width=4, lookup_reps=8)
    for name in ("variables", "sigmas", "constants", "multiplicities", "tables"):
        assert np.array_equal(getattr(back, name), getattr(c, name)), name
    assert (back.log_n, back.quotient_degree, back.table_id_col, back.num_constants_for_gates, back.total_tables_len, back.non_residues) == \
        (c.log_n, c.quotient_degree, c.table_id_col, c.num_constants_for_gates, c.total_tables_len, c.non_residues)
    assert [g.path for g in back.gates] == [g.path for g in c.gates]
    assert check_satisfied(back)


def test_the_librarys_reader_agrees_and_survives_mangled_dumps():
    """csrc/dumps.hip (what bj_setup_create_from_dump parses with) through its host-only entry point bj_setup_dump_info: the shape of
    a dump written by memcopy_format.write_setup_base, and thousands of truncated / bit-flipped / spliced variants of it — each is
    either refused or parsed to a shape, never read out of bounds (the reader checks every length against the buffer)."""
    import ctypes as C
    import random
    import era_boojum_amd as E
    from era_boojum_amd import memcopy_format as M, synthetic as S
    lib = E.load_library()
    c = S.sha_shaped_circuit(6, seed=3, table_bits=1)
    dump = M.write_setup_base(c)
    info = (C.c_uint64 * 8)()
    assert lib.bj_setup_dump_info(dump, len(dump), info) == 0
    n, n_sig, n_con, n_tab, n_ids, id0, gates, deg = list(info)
    longest = max(len(g.path) for g in c.gates)
    assert (n, n_sig, n_con, n_tab, n_ids, id0) == (c.n, c.num_vars, c.num_constant_cols, c.lookup_width + 1, 1, c.table_id_col)
    assert gates == len(c.gates) and deg >> 32 == longest and (deg & 0xFFFFFFFF) == max(len(g.path) + g.degree for g in c.gates)
    rnd = random.Random(7)
    refused = 0
    for _ in range(3000):
        b = bytearray(dump)
        kind = rnd.randrange(4)
        if kind == 0:
            b = b[:rnd.randrange(len(b))]
        elif kind == 1:
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            i = rnd.randrange(len(b) - 8)
            b[i:i + 8] = rnd.choice([(1 << 64) - 1, 1 << 40, 0, len(b)]).to_bytes(8, "little")      # a hostile length field
        else:
            i, j = sorted(rnd.randrange(len(b)) for _ in range(2))
            b = b[:i] + b[j:]
        rc = lib.bj_setup_dump_info(bytes(b), len(b), info)
        assert rc in (0, -1)
        refused += rc != 0
    assert refused > 1000


def test_hostile_gate_degrees_are_refused_not_looped_on():
    """GateDescription.degree / num_constants were read unchecked; a degree above 2^31 + 1 sent the doubling
    loop that derives the quotient degree to 0 for ever, and depth + degree could wrap.  Every such dump is refused at once by
    the host part bj_setup_create_from_dump shares with bj_setup_dump_info."""
    import ctypes as C
    import struct
    import time
    import era_boojum_amd as E
    from era_boojum_amd import memcopy_format as M, synthetic as S
    lib = E.load_library()
    c = S.sha_shaped_circuit(6, seed=3, table_bits=1)
    dump = M.write_setup_base(c)
    info = (C.c_uint64 * 8)()
    g = c.gates[1]
    leaf = struct.pack("<IQQQ", 1, 1, g.num_constants, g.degree)
    at = dump.rindex(leaf)
    t0 = time.perf_counter()
    for degree in ((1 << 31) + 5, (1 << 32) + 3, (1 << 63) + 1, (1 << 64) - 1, (1 << 64) - 2, 1 << 16 | 1, 200):
        bad = dump[:at] + struct.pack("<IQQQ", 1, 1, g.num_constants, degree) + dump[at + len(leaf):]
        assert lib.bj_setup_dump_info(bad, len(bad), info) == -1, degree
    for numc in ((1 << 64) - 1, (1 << 16) + 1):
        bad = dump[:at] + struct.pack("<IQQQ", 1, 1, numc, g.degree) + dump[at + len(leaf):]
        assert lib.bj_setup_dump_info(bad, len(bad), info) == -1, numc
    ok = dump[:at] + struct.pack("<IQQQ", 1, 1, g.num_constants, 30) + dump[at + len(leaf):]      # a legal one: quotient degree 32
    assert lib.bj_setup_dump_info(ok, len(ok), info) == 0
    assert time.perf_counter() - t0 < 5.0
