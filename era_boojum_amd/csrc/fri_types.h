// Internal: the object behind the opaque `bj_fri` (shared by fri_prover.hip and prover.hip).
#pragma once
#include "gl.h"
#include "../../include/boojum_hip.h"
#include <vector>
using gl::u64;

struct bj_ctx;
struct bj_transcript;
struct bj_fri;
namespace bj {
// which contiguous slice of every LDE-domain array this GPU holds, and how to reach the other slices
struct Shard {
    unsigned rank = 0, world = 1;
    bj_comm comm{};
};
// d_recv[r][elems] <- rank r's d_send[elems]   (world == 1: a device copy)
int all_gather(bj_ctx *ctx, const Shard &sh, const u64 *d_send, u64 *d_recv, size_t elems);
// all-gather [world][parts][part_len] and lay it out as [parts][world*part_len] (column-major arrays split by rows)
int all_gather_columns(bj_ctx *ctx, const Shard &sh, const u64 *d_send, u64 *d_dst, unsigned parts, size_t part_len);
// cap of a tree whose leaves are split across the ranks: every rank hashed its subtree down to cap_size/world nodes
int gather_cap(bj_ctx *ctx, const Shard &sh, const u64 *d_tree_local, size_t leaves_local, size_t cap_size,
               u64 *h_cap /* cap_size*4 */);
// do_fri on a codeword of which this rank holds [rank*N/world, (rank+1)*N/world)
int fri_prove_sharded(bj_ctx *ctx, const Shard &sh, const u64 *d_c0, const u64 *d_c1, unsigned log_n, unsigned log_lde,
                      const uint32_t *schedule, size_t schedule_len, size_t cap_size, bj_transcript *tr, bj_fri **out);
}  // namespace bj

struct bj_fri {
    int device = 0;
    size_t cap_size = 0;
    unsigned log_full = 0, log_lde = 0;
    struct Oracle {
        u64 *d_c0 = nullptr, *d_c1 = nullptr;  // leaf sources (oracle 0: the caller's codeword, not owned)
        size_t len = 0;
        unsigned log_e = 0;
        u64 *d_tree = nullptr;
        size_t num_leaves = 0;                 // leaves of the LOCAL tree (all of them unless `world` > 1)
        unsigned world = 1;                    // > 1: oracle 0 of a sharded proof; leaf l lives on rank l / num_leaves
        std::vector<u64> cap;
        u64 ch0 = 0, ch1 = 0;
    };
    std::vector<Oracle> oracles;
    std::vector<void *> owned;                   // hipMalloc'ed blocks to release (blocks from the proof arena are not listed)
    u64 *d_last0 = nullptr, *d_last1 = nullptr;  // last folded layer
    size_t last_len = 0;
    std::vector<u64> final_c0, final_c1;
    size_t final_degree = 0;
};

