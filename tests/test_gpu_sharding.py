"""-m gpu: the coset-range LDE entry point and the sharded commit path with the real HIP backend (world = 1 here;
world > 1 is covered on CPU over gloo in tests/test_sharding_gloo.py)."""
import numpy as np
import pytest

import oracle as O
import sharding_model as sharding
from gpu_util import DevBuf, ctx, rand_gl

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c0,cnt", [(0, 8), (0, 1), (3, 2), (6, 2), (7, 1)])
def test_lde_coset_range_matches_full_lde(c0, cnt):
    log_n, n_cols, log_lde = 12, 3, 3
    rng = np.random.default_rng(c0 * 10 + cnt)
    mono = rand_gl(rng, (n_cols, 1 << log_n), noncanonical=True)
    want = O.lde_batch(mono, log_lde, threads=4)[:, c0:c0 + cnt, :]
    d_m, d_o = DevBuf(mono), DevBuf(nelems=want.size)
    ctx().lde_cosets_batch(d_m.ptr, d_o.ptr, log_n, n_cols, log_lde, c0, cnt)
    assert np.array_equal(d_o.get(want.shape), want)
    d_m.free(); d_o.free()


def test_sharded_commit_on_gpu_emulating_four_ranks():
    """Each 'rank' runs sequentially on the one GPU with the real Context backend; concatenating the per-rank cap
    fragments (what the RCCL all-gather does) must give the cap of the full tree."""
    log_n, n_cols, log_lde, cap, world = 10, 9, 3, 16, 4
    rng = np.random.default_rng(8)
    mono = rand_gl(rng, (n_cols, 1 << log_n))
    d_m = DevBuf(mono)
    frags = []
    for rank in range(world):
        c0, cnt = sharding.coset_shard(1 << log_lde, world, rank)
        frag = sharding.cap_fragment_size(cap, world)
        leaves = cnt << log_n
        d_lde = DevBuf(nelems=n_cols * leaves)
        ctx().lde_cosets_batch(d_m.ptr, d_lde.ptr, log_n, n_cols, log_lde, c0, cnt)
        d_t = DevBuf(nelems=4 * (2 * leaves - frag))
        ctx().merkle_tree_build(d_lde.ptr, leaves, n_cols, leaves, frag, d_t.ptr)
        frags.append(ctx().merkle_tree_cap(d_t.ptr, leaves, frag))
        d_lde.free(); d_t.free()
    N = (1 << log_n) << log_lde
    lde = O.lde_batch(mono, log_lde, threads=4).reshape(n_cols, N)
    want = O.merkle_cap(O.merkle_construct(lde, cap, threads=4), N, cap)
    assert np.array_equal(np.concatenate(frags), want)
    # and through sharding.sharded_commit with world = 1
    d_lde, d_tree, leaves, cap_full = sharding.sharded_commit(ctx(), d_m.ptr, log_n, n_cols, log_lde, cap, 1, 0)
    assert leaves == N and np.array_equal(cap_full, want)
    ctx().free(d_lde); ctx().free(d_tree); d_m.free()
