// Internal: the context object behind the opaque `bj_ctx` of include/boojum_hip.h, shared by the C-ABI translation
// units (abi.hip, fri_prover.hip, ...).
#pragma once
#include "../../include/boojum_hip.h"
#include "gl.h"
#include "kernels.h"

#include <hip/hip_runtime.h>
#include <string>
#include <utility>
#include <vector>

#define BJ_MAX_KERNEL_PROBES 12

namespace bj {
struct Pipeline;   // prove_async.hip: the two lanes of bj_prove_async
}

struct bj_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // twiddle caches (bit-reversed tables; a table for 2^k serves every smaller size as a prefix)
    gl::u64 *tw_fwd = nullptr, *tw_inv = nullptr;
    unsigned tw_fwd_log = 0, tw_inv_log = 0;
    gl::u64 *tw_inv_scaled22 = nullptr;   // tw_inv[j] / 2^22, j < 2^21: the last round of a 2^22-point inverse transform into the tiled layout
    gl::u64 *d_small = nullptr;  // 64 shifts + 64*32 per-round scales + 4096 for gathered Merkle caps + the front-pass twiddle table
    const gl::u64 **d_ptrs = nullptr;
    size_t d_ptrs_cap = 0;
    gl::u64 *d_scratch = nullptr;  // big scratch for out-of-place steps
    size_t scratch_elems = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // bump-allocated workspace reused across proofs (hipMalloc/hipFree of multi-GB buffers per proof is slow and
    // synchronising); grown on demand, reset at the start of every bj_prove_dev
    gl::u64 *arena = nullptr;
    size_t arena_elems = 0, arena_off = 0;
    // A proof whose buffers outgrow the reservation does not fail: further blocks come out of overflow slabs (hipMalloc in the
    // middle of a proof — slow, once), the proof reports its high-water mark (bj_proof_workspace_bytes) and the next reservation
    // on this context is at least that large.  The suite asserts that no proof needed a slab (the reservation is an upper bound).
    std::vector<std::pair<gl::u64 *, size_t>> arena_slabs;   // (base, elems), each slab bump-allocated like the arena
    size_t slab_off = 0;                 // elements used in the last slab
    size_t arena_high_water = 0;         // elements the proof in flight has taken (arena + slabs)
    size_t arena_learned = 0;            // largest high-water mark seen on this context
    int hasher = BJ_HASHER_POSEIDON2;   // tree hasher of the bj_merkle_tree_* calls (bj_ctx_set_tree_hasher / bj_prove)
    bool in_proof = false;       // bj_prove_dev is running: temporaries come out of the arena instead of hipMalloc
    // pinned staging ring for the small host->device blocks between kernels (challenges, pointer tables, query indices):
    // a copy out of pageable memory costs a blocking staging pass plus a stream synchronisation, ~35 us of idle GPU each
    unsigned char *h_ring = nullptr;
    size_t ring_off = 0, ring_inflight = 0;
    // bj_prove (host witness): device staging of the witness columns, kept across proofs, and a copy stream with one event
    // per column group so that the PCIe transfer of group k+1 runs under the iNTT / LDE of group k
    gl::u64 *wit_stage = nullptr;
    size_t wit_stage_elems = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[64] = {};
    // collectives of the sharded proof in flight: an event pair around each (created on first use, reused), summed into
    // bj_proof_comm_stats when the proof has drained — what a scaling run needs to tell waiting-for-peers from computing
    hipEvent_t comm_ev[48][2] = {};
    unsigned comm_n = 0;          // pairs recorded by the proof in flight
    size_t comm_bytes = 0;        // bytes received by this rank in them
    float comm_host_ms = 0;       // synchronous host-callback transport: wall time inside the callbacks
    // per-kernel probes of the proof in flight (bj_proof_kernel_stats): an event pair on the launch stream around the first
    // launch of each named kernel, with the algorithmic bytes of that launch — no synchronisation added
    struct KernelProbe {
        const char *name = nullptr;
        double bytes = 0;
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool closed = false;      // the closing event was recorded in THIS proof (an unclosed probe still holds the previous proof's)
    } probes[BJ_MAX_KERNEL_PROBES];
    unsigned probe_n = 0;
    bj::Pipeline *pipe = nullptr;   // bj_prove_async: created on first use, destroyed with the context
};

namespace bj {
// first launch of `name` inside a proof: records the opening event and returns the probe's index (-1: not in a proof, name
// already probed in this proof, or table full — the caller just launches); probe_end records the closing event
int probe_begin(bj_ctx *ctx, const char *name, double algorithmic_bytes);
void probe_end(bj_ctx *ctx, int idx);
int fail(bj_ctx *ctx, int code, const char *fmt, ...);
int bind(bj_ctx *ctx);
// host block -> device, ordered on ctx->stream like a kernel launch; returns without waiting (h_src may be reused at once)
int h2d_async(bj_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int ensure_twiddles(bj_ctx *ctx, unsigned log_n, bool inverse);
int ensure_scratch(bj_ctx *ctx, size_t elems);
inline gl::u64 *front_table(bj_ctx *ctx) { return ctx->d_small + 64 + 64 * 32 + 4096; }   // BJ_FRONT_TABLE_WORDS (kernels.h)
unsigned setup_world(const bj_setup *s);         // ranks of the proof a setup belongs to (1: single device)
// mode 1: whole witness first, then as on a resident witness; mode 2: groups transferred and transformed as they land, hashed once
int prove_host_copy_first(bj_ctx *ctx, const bj_setup *S, const uint64_t *h_variables, const uint64_t *h_multiplicities,
                          const uint64_t *h_public_values, bj_proof **out, int mode);
void pipeline_destroy(bj_ctx *ctx);              // waits for the asynchronous proofs in flight, ends the lanes
int pipeline_release_workspace(bj_ctx *ctx);    // bj_ctx_release_workspace on every idle lane
int arena_reset(bj_ctx *ctx, size_t need_elems);
int arena_drop_slabs(bj_ctx *ctx);
gl::u64 *arena_alloc(bj_ctx *ctx, size_t elems);   // nullptr if the reservation was too small
// short-lived device memory: from the arena inside a proof (no hipMalloc/hipFree, no implicit syncs), hipMalloc otherwise
void *tmp_alloc(bj_ctx *ctx, size_t bytes, bool *from_arena);
void tmp_free(bj_ctx *ctx, void *p, bool from_arena);
int lde_cosets_strided(bj_ctx *ctx, const gl::u64 *d_mono, size_t in_col_stride, gl::u64 *d_out, size_t out_col_stride,
                       unsigned log_n, unsigned n_cols, unsigned log_lde, unsigned coset_begin, unsigned coset_count,
                       bool tiled_in = false);
bool mono_tiled(unsigned log_n);   // the monomial layout of bj_prove for this trace length (abi.hip)
int intt_to_tiled(bj_ctx *ctx, const gl::u64 *d_in, size_t in_col_stride, gl::u64 *d_out, size_t out_col_stride, unsigned log_n,
                  unsigned n_cols);
inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
inline unsigned log2_exact(size_t x) {
    unsigned r = 0;
    while (((size_t)1 << r) < x) r++;
    return r;
}
}  // namespace bj

#define BJ_HIP(ctx, call)                                                                                       \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return bj::fail(ctx, e_ == hipErrorOutOfMemory ? BJ_ERR_OOM : BJ_ERR_HIP, "%s failed: %s", #call,   \
                            hipGetErrorString(e_));                                                             \
    } while (0)

#define BJ_CHECK_LAUNCH(ctx)                                                                                    \
    do {                                                                                                        \
        hipError_t e_ = hipGetLastError();                                                                      \
        if (e_ != hipSuccess)                                                                                   \
            return bj::fail(ctx, BJ_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e_));                \
    } while (0)
