"""-m gpu parity: HIP NTT / iNTT / LDE (through the C ABI) vs the CPU oracle, bit-exact canonical residues.
Mirrors the reference's differential tests (fft/mod.rs:1345-1709) incl. non-canonical inputs and coset 7."""
import numpy as np
import pytest

import oracle as O
import era_boojum_amd as E
from gpu_util import DevBuf, ctx, rand_gl, P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", list(range(0, 20)) + [22])
@pytest.mark.parametrize("coset", [1, 7])
def test_forward_matches_oracle(log_n, coset):
    rng = np.random.default_rng(1000 + log_n)
    n_cols = 3 if log_n < 16 else 2
    a = rand_gl(rng, (n_cols, 1 << log_n), noncanonical=True)
    want = O.fft_batch(a, coset, threads=4)
    d_in, d_out = DevBuf(a), DevBuf(nelems=a.size)
    ctx().ntt_forward_batch(d_in.ptr, d_out.ptr, log_n, n_cols, coset=coset)
    got = d_out.get(a.shape)
    assert np.array_equal(got, want)
    # in place
    ctx().ntt_forward_batch(d_in.ptr, d_in.ptr, log_n, n_cols, coset=coset)
    assert np.array_equal(d_in.get(a.shape), want)
    d_in.free(); d_out.free()


def test_forward_random_coset_and_strided_columns():
    log_n, n_cols, stride = 11, 5, (1 << 11) + 64
    rng = np.random.default_rng(5)
    buf = rand_gl(rng, (n_cols, stride))
    coset = 0x123456789ABCDEF % P
    want = O.fft_batch(np.ascontiguousarray(buf[:, :1 << log_n]), coset)
    d = DevBuf(buf)
    ctx().ntt_forward_batch(d.ptr, d.ptr, log_n, n_cols, col_stride=stride, coset=coset)
    got = d.get(buf.shape)
    assert np.array_equal(got[:, :1 << log_n], want)
    assert np.array_equal(got[:, 1 << log_n:], buf[:, 1 << log_n:])  # gap between columns untouched
    d.free()


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 12, 13, 14, 15, 16, 18])
@pytest.mark.parametrize("coset", [1, 7])
def test_inverse_matches_oracle_and_roundtrips(log_n, coset):
    rng = np.random.default_rng(2000 + log_n)
    a = rand_gl(rng, (2, 1 << log_n), noncanonical=True)
    want = O.ifft_batch(a, coset, threads=2)
    d = DevBuf(a)
    out = DevBuf(nelems=a.size)
    ctx().intt_batch(d.ptr, out.ptr, log_n, 2, coset=coset)
    assert np.array_equal(out.get(a.shape), want)
    # NTT(iNTT(x)) on the same coset, un-bit-reversed, is x
    ctx().ntt_forward_batch(out.ptr, out.ptr, log_n, 2, coset=coset)
    ctx().bitreverse_batch(out.ptr, out.ptr, log_n, 2)
    assert np.array_equal(out.get(a.shape), O.canonical(a))
    d.free(); out.free()


@pytest.mark.parametrize("log_n,log_lde", [(0, 1), (3, 1), (6, 3), (10, 2), (13, 3), (14, 1), (15, 2), (17, 3), (18, 3), (19, 1)])
def test_lde_matches_oracle(log_n, log_lde):
    rng = np.random.default_rng(3000 + log_n)
    n_cols = 3
    mono = rand_gl(rng, (n_cols, 1 << log_n), noncanonical=True)
    want = O.lde_batch(mono, log_lde, threads=4)
    d_m, d_o = DevBuf(mono), DevBuf(nelems=want.size)
    ctx().lde_batch(d_m.ptr, d_o.ptr, log_n, n_cols, log_lde)
    assert np.array_equal(d_o.get(want.shape), want)
    d_m.free(); d_o.free()


def test_trace_to_lde_matches_oracle():
    log_n, log_lde, n_cols = 12, 3, 4
    rng = np.random.default_rng(77)
    trace = rand_gl(rng, (n_cols, 1 << log_n))
    mono = O.ifft_batch(trace, 1, threads=4)
    want = O.lde_batch(mono, log_lde, threads=4)
    d_t, d_o = DevBuf(trace), DevBuf(nelems=want.size)
    ctx().trace_to_lde_batch(d_t.ptr, d_o.ptr, log_n, n_cols, log_lde)
    assert np.array_equal(d_o.get(want.shape), want)
    assert np.array_equal(d_t.get(trace.shape), mono)      # monomials are left in the input buffer
    # first coset of an LDE with shift g equals a plain coset-7 NTT (utils.rs:345-346 with bitrev(0) = 0)
    assert np.array_equal(want[:, 0, :], O.fft_batch(mono, 7))
    d_t.free(); d_o.free()


def test_host_convenience_path():
    rng = np.random.default_rng(9)
    a = rand_gl(rng, (2, 1 << 10), noncanonical=True)
    assert np.array_equal(ctx().ntt_forward_host(a, 7), O.fft_batch(a, 7))
    assert np.array_equal(ctx().intt_host(a, 7), O.ifft_batch(a, 7))


def test_bad_arguments_are_reported_not_crashed():
    import era_boojum_amd as E
    d = DevBuf(nelems=16)
    with pytest.raises(E.BoojumHipError):
        ctx().ntt_forward_batch(d.ptr, d.ptr, 33, 1)
    with pytest.raises(E.BoojumHipError):
        ctx().ntt_forward_batch(d.ptr, d.ptr, 3, 2, col_stride=4)
    with pytest.raises(E.BoojumHipError):
        ctx().lde_batch(d.ptr, d.ptr, 2, 1, 1)
    with pytest.raises(E.BoojumHipError):
        ctx().ntt_forward_batch(d.ptr, d.ptr, 2, 1, coset=0)
    d.free()


def test_full_size_2_20_properties_and_spot_columns():
    """BASELINE cfg2 shape (2^20 rows; 16 of the 256 columns here): spot columns vs the oracle, plus the
    size-independent properties linearity and iNTT∘NTT = id on all columns."""
    log_n, n_cols = 20, 16
    rng = np.random.default_rng(20240807)
    a = rand_gl(rng, (n_cols, 1 << log_n), noncanonical=True)
    d_a, d_o = DevBuf(a), DevBuf(nelems=a.size)
    for coset in (1, 7):
        ctx().ntt_forward_batch(d_a.ptr, d_o.ptr, log_n, n_cols, coset=coset)
        got = d_o.get(a.shape)
        want = O.fft_batch(a[[0, 7, 15]], coset, threads=3)
        assert np.array_equal(got[[0, 7, 15]], want)
        # linearity: NTT(col0 + col1) == NTT(col0) + NTT(col1)
        s = ((a[0].astype(object) + a[1].astype(object)) % P).astype(np.uint64)
        d_s = DevBuf(s)
        ctx().ntt_forward_batch(d_s.ptr, d_s.ptr, log_n, 1, coset=coset)
        lhs = d_s.get()
        rhs = ((got[0].astype(object) + got[1].astype(object)) % P).astype(np.uint64)
        assert np.array_equal(lhs, rhs)
        d_s.free()
        # round trip on all columns
        ctx().bitreverse_batch(d_o.ptr, d_o.ptr, log_n, n_cols)
        ctx().intt_batch(d_o.ptr, d_o.ptr, log_n, n_cols, coset=coset)
        assert np.array_equal(d_o.get(a.shape), O.canonical(a))
    d_a.free(); d_o.free()


def _splitmix64_column(seed, n):
    """n outputs of SplitMix64 started at `seed`, reduced to [0, p) (column c of cfg2 uses seed 20240807 + c)."""
    g = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + g * np.arange(1, n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % np.uint64(P)


def test_splitmix64_known_answer():
    # first outputs of SplitMix64(seed = 0), the generator's published test vector
    g = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over="ignore"):
        z = g * np.arange(1, 4, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]


@pytest.mark.parametrize("noncanonical", [False, True])
def test_cfg2_all_256_columns_match_the_oracle(noncanonical):
    """BASELINE cfg2 exactly as SURVEY §8d defines it: 256 columns x 2^20, column c from SplitMix64(20240807 + c); forward NTT
    natural -> bit-reversed with coset shift 1 and 7; EVERY column compared with the oracle.  Second run: a fraction of the
    values forced into [p, 2^64) (non-canonical inputs are legal in the reference's memory)."""
    log_n, n_cols = 20, 256
    n = 1 << log_n
    a = np.stack([_splitmix64_column(20240807 + c, n) for c in range(n_cols)])
    if noncanonical:
        rng = np.random.default_rng(1)
        idx = rng.integers(0, a.size, size=a.size // 16)
        flat = a.reshape(-1)
        small = flat[idx] < np.uint64((1 << 32) - 1)
        flat[idx[small]] += np.uint64(P)             # same residue, representative in [p, 2^64)
        flat[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
        flat[-1] = np.uint64(P)
    d_a, d_o = DevBuf(a), DevBuf(nelems=a.size)
    for coset in (1, 7):
        ctx().ntt_forward_batch(d_a.ptr, d_o.ptr, log_n, n_cols, coset=coset)
        got = d_o.get(a.shape)
        want = O.fft_batch(a, coset, threads=64)
        assert np.array_equal(got, want), "coset %d: columns %s differ" % (coset, np.nonzero((got != want).any(axis=1))[0][:8])
    d_a.free(); d_o.free()


@pytest.mark.parametrize("log_n,log_lde,n_cols", [(20, 3, 6), (22, 3, 3), (21, 1, 3), (23, 1, 1)])
def test_lde_at_bench_sizes_matches_oracle(log_n, log_lde, n_cols):
    """The LDE of the bench sizes (2^20 and 2^22 rows x 8 cosets: the multi-pass NTT paths incl. the remainder rounds of sizes
    with (log n - 12) mod 4 != 0) on a few columns, every value against the oracle."""
    rng = np.random.default_rng(4000 + log_n)
    mono = rand_gl(rng, (n_cols, 1 << log_n), noncanonical=True)
    want = O.lde_batch(mono, log_lde, threads=32)
    d_m, d_o = DevBuf(mono), DevBuf(nelems=want.size)
    ctx().lde_batch(d_m.ptr, d_o.ptr, log_n, n_cols, log_lde)
    assert np.array_equal(d_o.get(want.shape), want)
    # the inverse transform at the same size brings coset 0 back to the monomials
    if log_lde:
        d_c = DevBuf(want[:, 0, :])
        ctx().bitreverse_batch(d_c.ptr, d_c.ptr, log_n, n_cols)
        ctx().intt_batch(d_c.ptr, d_c.ptr, log_n, n_cols, coset=7)
        assert np.array_equal(d_c.get(mono.shape), O.canonical(mono))
        d_c.free()
    d_m.free(); d_o.free()


def _tiled_index(e):
    """The tiled layout of a 2^22-word column (include/boojum_hip.h, csrc/ntt_r16.hip: tiled_index)."""
    e = np.asarray(e, dtype=np.int64)       # e = m * 4096 + r * 8 + l: tile r (nine bits) = 8192 contiguous words, ordered (l pair, m, l & 1)
    return ((e >> 3) & 511) * 8192 + ((e >> 1) & 3) * 2048 + (e >> 12) * 2 + (e & 1)


@pytest.mark.parametrize("n_cols", [1, 5, 40])
def test_tiled_monomials_inverse_transform_and_extension_match_the_oracle(n_cols):
    """What bj_prove runs at 2^22 rows since round 6 (ifft_natural_to_natural fft/mod.rs:464-491 without a bit-reversal pass;
    transform_monomials_to_lde utils.rs:311-403 reading contiguous tiles): bj_intt_batch_tiled = the oracle's monomials at their
    tiled positions, in place and out of place, for column counts that take one workgroup column loop, several, and more than one
    scratch group (32 columns per GiB); bj_lde_cosets_batch_tiled of them = the oracle's LDE, all eight cosets and a rank's single
    coset; bj_tiled_permute_batch both ways; every other size is refused."""
    log_n, n = 22, 1 << 22
    assert ctx().monomials_tiled(22) and not ctx().monomials_tiled(21) and not ctx().monomials_tiled(23)
    rng = np.random.default_rng(2200 + n_cols)
    vals = rand_gl(rng, (n_cols, n), noncanonical=True)
    mono = O.ifft_batch(vals, 1, threads=32)
    perm = _tiled_index(np.arange(n))
    want_tiled = np.empty_like(mono)
    want_tiled[:, perm] = mono
    d_v, d_m = DevBuf(vals), DevBuf(nelems=vals.size)
    ctx().intt_batch_tiled(d_v.ptr, d_m.ptr, log_n, n_cols)
    assert np.array_equal(d_m.get(vals.shape), want_tiled)
    ctx().intt_batch_tiled(d_v.ptr, d_v.ptr, log_n, n_cols)                  # in place
    assert np.array_equal(d_v.get(vals.shape), want_tiled)
    d_b = DevBuf(nelems=vals.size)
    ctx().tiled_permute_batch(d_m.ptr, d_b.ptr, log_n, n_cols, to_tiled=False)
    assert np.array_equal(d_b.get(vals.shape), mono)
    ctx().tiled_permute_batch(d_b.ptr, d_v.ptr, log_n, n_cols, to_tiled=True)
    assert np.array_equal(d_v.get(vals.shape), want_tiled)
    k = min(n_cols, 3)                                                       # the extension: a few columns against the oracle
    want = O.lde_batch(mono[:k], 3, threads=32)
    d_o = DevBuf(nelems=want.size)
    ctx().lde_cosets_batch_tiled(d_m.ptr, d_o.ptr, log_n, k, 3, 0, 8)
    assert np.array_equal(d_o.get(want.shape), want)
    ctx().lde_cosets_batch_tiled(d_m.ptr, d_o.ptr, log_n, k, 3, 5, 1)         # one coset, as a rank of eight extends it
    assert np.array_equal(d_o.get((k * n,)).reshape(k, n), want[:, 5, :])
    with pytest.raises(E.BoojumHipError):
        ctx().intt_batch_tiled(d_v.ptr, d_m.ptr, 21, 1)
    with pytest.raises(E.BoojumHipError):
        ctx().lde_cosets_batch_tiled(d_m.ptr, d_o.ptr, 20, 1, 3, 0, 8)
    for d in (d_v, d_m, d_b, d_o):
        d.free()


@pytest.mark.parametrize("log_n,n_cols", [(14, 5), (15, 3), (18, 3), (19, 2), (22, 2)])
def test_lde_and_ntt_with_unaligned_buffers_and_odd_strides(log_n, n_cols):
    """A host may hand over columns at any 8-byte address and stride: the 16-byte vector accesses of the coset-expanding front
    pass (ntt_first4) then give way to the one-index-per-lane plan (ntt_first5 + the nine-round local pass for sizes with
    14 + 4k rounds), which no aligned caller ever reaches.  Same values as the oracle, neighbours of the columns untouched."""
    n, L = 1 << log_n, 8
    rng = np.random.default_rng(4000 + log_n)
    stride = n + 1                                           # odd column stride
    buf = rand_gl(rng, (n_cols * stride + 1,), noncanonical=True)
    mono = np.stack([buf[1 + c * stride: 1 + c * stride + n] for c in range(n_cols)])      # columns start 8 bytes off a 16-byte line
    want = O.lde_batch(O.canonical(mono), 3, threads=8)
    d_in = DevBuf(buf)
    d_out = DevBuf(nelems=n_cols * L * n + 1)
    ctx().lde_batch(d_in.ptr + 8, d_out.ptr + 8, log_n, n_cols, 3, col_stride=stride)
    got = d_out.get((n_cols * L * n + 1,))[1:].reshape(n_cols, L, n)
    assert np.array_equal(got, want)
    assert np.array_equal(d_in.get(buf.shape), buf)          # out of place: the input (and the gaps between its columns) is intact
    # the forward transform alone, in place on the odd-strided, offset columns
    want_f = O.fft_batch(O.canonical(mono), 7, threads=8)
    ctx().ntt_forward_batch(d_in.ptr + 8, d_in.ptr + 8, log_n, n_cols, col_stride=stride, coset=7)
    after = d_in.get(buf.shape)
    for c in range(n_cols):
        assert np.array_equal(after[1 + c * stride: 1 + c * stride + n], want_f[c])
        if c + 1 < n_cols:
            assert after[1 + c * stride + n] == buf[1 + c * stride + n]                     # the gap element
    assert after[0] == buf[0]
    d_in.free(); d_out.free()
