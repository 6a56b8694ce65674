// C-ABI layer of libboojum_hip.so: context management + argument validation + kernel orchestration.
// Public contract: include/boojum_hip.h.  No CPU fallback anywhere in this file: every entry point either
// enqueues HIP work on the context's device or returns an error.
#include "ctx.h"

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using gl::u64;

namespace bj {

int fail(bj_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int bind(bj_ctx *ctx) {
    if (!ctx) return BJ_ERR_INVALID_ARG;
    BJ_HIP(ctx, hipSetDevice(ctx->device));
    return BJ_OK;
}

int ensure_twiddles(bj_ctx *ctx, unsigned log_n, bool inverse) {
    u64 *&tab = inverse ? ctx->tw_inv : ctx->tw_fwd;
    unsigned &have = inverse ? ctx->tw_inv_log : ctx->tw_fwd_log;
    if (log_n <= have && tab) return BJ_OK;
    if (log_n == 0) log_n = 1;
    if (tab) {
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BJ_HIP(ctx, hipFree(tab));
        tab = nullptr;
        have = 0;
    }
    size_t half = (size_t)1 << (log_n - 1);
    BJ_HIP(ctx, hipMalloc((void **)&tab, half * sizeof(u64)));
    bj::launch_twiddles(tab, log_n, inverse, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    have = log_n;
    return BJ_OK;
}

int ensure_scratch(bj_ctx *ctx, size_t elems) {
    if (elems <= ctx->scratch_elems) return BJ_OK;
    if (ctx->d_scratch) {
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BJ_HIP(ctx, hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_elems = 0;
    }
    BJ_HIP(ctx, hipMalloc((void **)&ctx->d_scratch, elems * sizeof(u64)));
    ctx->scratch_elems = elems;
    return BJ_OK;
}

int arena_drop_slabs(bj_ctx *ctx) {
    if (ctx->arena_slabs.empty()) return BJ_OK;
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &sl : ctx->arena_slabs) (void)hipFree(sl.first);
    ctx->arena_slabs.clear();
    ctx->slab_off = 0;
    return BJ_OK;
}
int arena_reset(bj_ctx *ctx, size_t need_elems) {
    ctx->arena_off = 0;
    ctx->arena_high_water = 0;
    if (int rc = arena_drop_slabs(ctx)) return rc;
    if (need_elems <= ctx->arena_elems) return BJ_OK;
    if (ctx->arena) {
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BJ_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_elems = 0;
    }
    BJ_HIP(ctx, hipMalloc((void **)&ctx->arena, need_elems * sizeof(u64)));
    ctx->arena_elems = need_elems;
    return BJ_OK;
}

void *tmp_alloc(bj_ctx *ctx, size_t bytes, bool *from_arena) {
    *from_arena = false;
    if (ctx->in_proof) {
        if (u64 *p = arena_alloc(ctx, (bytes + 7) / 8)) {
            *from_arena = true;
            return p;
        }
    }
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return nullptr;
    return p;
}
void tmp_free(bj_ctx *ctx, void *p, bool from_arena) {
    (void)ctx;
    if (p && !from_arena) (void)hipFree(p);
}

constexpr size_t RING_BYTES = (size_t)1 << 20, RING_MAX_BLOCK = (size_t)128 << 10;
int h2d_async(bj_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (int rc = bind(ctx)) return rc;
    if (!bytes) return BJ_OK;
    if (!d_dst || !h_src) return fail(ctx, BJ_ERR_INVALID_ARG, "h2d_async: null pointer");
    if (bytes > RING_MAX_BLOCK) return bj_memcpy_h2d(ctx, d_dst, h_src, bytes);
    if (!ctx->h_ring) BJ_HIP(ctx, hipHostMalloc((void **)&ctx->h_ring, RING_BYTES, hipHostMallocDefault));
    const size_t slot = (bytes + 63) & ~(size_t)63;
    if (ctx->ring_off + slot > RING_BYTES) {   // wrap: the skipped tail counts as in flight, or a reset at a mid-ring offset
        ctx->ring_inflight += RING_BYTES - ctx->ring_off;   // would let the next lap overwrite copies still queued
        ctx->ring_off = 0;
    }
    if (ctx->ring_inflight + slot > RING_BYTES) {   // the ring wrapped onto copies that may not have run yet
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->ring_inflight = 0;
    }
    std::memcpy(ctx->h_ring + ctx->ring_off, h_src, bytes);
    BJ_HIP(ctx, hipMemcpyAsync(d_dst, ctx->h_ring + ctx->ring_off, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->ring_off += slot;
    ctx->ring_inflight += slot;
    return BJ_OK;
}

u64 *arena_alloc(bj_ctx *ctx, size_t elems) {
    const size_t start = (ctx->arena_off + 63) & ~(size_t)63;   // 512-byte alignment
    if (ctx->arena_slabs.empty() && start + elems <= ctx->arena_elems) {
        ctx->arena_off = start + elems;
        if (ctx->arena_off > ctx->arena_high_water) ctx->arena_high_water = ctx->arena_off;
        return ctx->arena + start;
    }
    // the reservation was too small: overflow slabs (only inside a proof: the arena belongs to the proof in flight)
    if (!ctx->in_proof) return nullptr;
    if (!ctx->arena_slabs.empty()) {
        auto &last = ctx->arena_slabs.back();
        const size_t st = (ctx->slab_off + 63) & ~(size_t)63;
        if (st + elems <= last.second) {
            ctx->slab_off = st + elems;
            ctx->arena_high_water += elems + 64;
            return last.first + st;
        }
    }
    const size_t slab = elems > ((size_t)1 << 25) ? elems : ((size_t)1 << 25);   // at least 256 MiB
    u64 *p = nullptr;
    if (hipMalloc((void **)&p, slab * sizeof(u64)) != hipSuccess) return nullptr;
    ctx->arena_slabs.emplace_back(p, slab);
    ctx->slab_off = elems;
    if (ctx->arena_high_water < ctx->arena_off) ctx->arena_high_water = ctx->arena_off;
    ctx->arena_high_water += elems + 64;
    return p;
}


int probe_begin(bj_ctx *ctx, const char *name, double algorithmic_bytes) {
    if (!ctx->in_proof || ctx->probe_n >= BJ_MAX_KERNEL_PROBES) return -1;
    for (unsigned i = 0; i < ctx->probe_n; i++)
        if (!strcmp(ctx->probes[i].name, name)) return -1;
    auto &p = ctx->probes[ctx->probe_n];
    if (!p.ev[0] && hipEventCreate(&p.ev[0]) != hipSuccess) return -1;
    if (!p.ev[1] && hipEventCreate(&p.ev[1]) != hipSuccess) return -1;
    p.name = name;
    p.bytes = algorithmic_bytes;
    p.closed = false;
    if (hipEventRecord(p.ev[0], ctx->stream) != hipSuccess) return -1;
    return (int)ctx->probe_n++;
}
void probe_end(bj_ctx *ctx, int idx) {
    if (idx >= 0) ctx->probes[idx].closed = hipEventRecord(ctx->probes[idx].ev[1], ctx->stream) == hipSuccess;
}

namespace {
EnvConfig g_env;
std::once_flag g_env_once;
void load_env() {
    EnvConfig e;
    auto set = [](const char *name) { return getenv(name) != nullptr; };
    auto str = [](const char *name) { const char *v = getenv(name); return std::string(v ? v : ""); };
    e.ntt_first_narrow = set("BJ_NTT_FIRST_NARROW");
    e.ntt_generic = str("BJ_NTT_GENERIC").rfind("1", 0) == 0;
    e.ntt_generic_remainder = set("BJ_NTT_GENERIC_REMAINDER");
    e.bitrev_gather = set("BJ_BITREV_GATHER");
    if (set("BJ_NTT_FRONT")) e.ntt_front = atoi(getenv("BJ_NTT_FRONT"));
    e.ntt_first4_v = str("BJ_NTT_FIRST4_V").rfind("1", 0) == 0 ? 1 : 2;
    if (set("BJ_NTT_FIRST4_MODE")) e.ntt_first4_mode = atoi(getenv("BJ_NTT_FIRST4_MODE"));
    e.ntt_two_pass = str("BJ_NTT_TWO_PASS").rfind("0", 0) != 0;
    e.mono_tiled = str("BJ_MONO_TILED").rfind("0", 0) != 0;
    if (set("BJ_ASYNC_MODE")) e.async_mode = atoi(getenv("BJ_ASYNC_MODE"));
    e.async_stagger = str("BJ_ASYNC_STAGGER").rfind("0", 0) != 0;
    e.gate_no_aot = set("BJ_GATE_NO_AOT");
    e.gate_no_fuse = set("BJ_GATE_NO_FUSE");
    e.gate_no_jit = set("BJ_GATE_NO_JIT");
    e.gates_windowed = str("BJ_GATES_WINDOWED").rfind("0", 0) != 0;
    e.prove_no_absorb = set("BJ_PROVE_NO_ABSORB");
    e.prove_uniform_groups = set("BJ_PROVE_UNIFORM_GROUPS");
    e.copy_perm_wide_k = set("BJ_COPY_PERM_WIDE_K");
    if (set("BJ_PROVE_H2D_GROUP")) {
        const unsigned v = (unsigned)strtoul(getenv("BJ_PROVE_H2D_GROUP"), nullptr, 10);
        e.prove_h2d_group = v ? v : 8u;
    }
    if (set("BJ_NODES_LANEPAR_MAX")) e.nodes_lanepar_max = (size_t)strtoull(getenv("BJ_NODES_LANEPAR_MAX"), nullptr, 10);
    e.jit_cache_dir = str("BJ_GATE_JIT_CACHE");
    e.rccl_lib = str("BJ_RCCL_LIB");
    g_env = e;
}
}  // namespace
const EnvConfig &env() {
    std::call_once(g_env_once, load_env);
    return g_env;
}
void env_reload() {
    (void)env();
    load_env();
}
}  // namespace bj

using bj::bind;
using bj::ensure_scratch;
using bj::ensure_twiddles;
using bj::fail;
using bj::is_pow2;
using bj::log2_exact;

extern "C" {

int bj_abi_version(void) { return BJ_ABI_VERSION; }

int bj_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *bj_status_string(int status) {
    switch (status) {
        case BJ_OK: return "ok";
        case BJ_ERR_INVALID_ARG: return "invalid argument";
        case BJ_ERR_NO_DEVICE: return "no usable HIP device";
        case BJ_ERR_HIP: return "HIP runtime error";
        case BJ_ERR_OOM: return "out of device memory";
        case BJ_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}

void bj_env_reload(void) { bj::env_reload(); }

int bj_ctx_create(int device, bj_ctx **out) {
    if (!out) return BJ_ERR_INVALID_ARG;
    *out = nullptr;
    (void)bj::env();      // the one place the environment is read
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return BJ_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return BJ_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return BJ_ERR_NO_DEVICE;
    bj_ctx *ctx = new bj_ctx();
    ctx->device = device;
    if (hipMalloc((void **)&ctx->d_small, (64 + 64 * 32 + 4096 + bj::BJ_FRONT_TABLE_WORDS) * sizeof(u64)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return BJ_ERR_HIP;
    }
    *out = ctx;
    return BJ_OK;
}

void bj_ctx_destroy(bj_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    bj::pipeline_destroy(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->tw_fwd) (void)hipFree(ctx->tw_fwd);
    if (ctx->tw_inv) (void)hipFree(ctx->tw_inv);
    if (ctx->tw_inv_scaled22) (void)hipFree(ctx->tw_inv_scaled22);
    if (ctx->d_small) (void)hipFree(ctx->d_small);
    if (ctx->d_ptrs) (void)hipFree((void *)ctx->d_ptrs);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->wit_stage) (void)hipFree(ctx->wit_stage);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (auto &pair : ctx->comm_ev)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->copy_ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &p : ctx->probes)
        for (hipEvent_t e : p.ev)
            if (e) (void)hipEventDestroy(e);
    if (ctx->arena) (void)hipFree(ctx->arena);
    for (auto &sl : ctx->arena_slabs) (void)hipFree(sl.first);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

int bj_ctx_release_workspace(bj_ctx *ctx) {
    if (int rc = bj::bind(ctx)) return rc;
    if (ctx->in_proof) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_ctx_release_workspace: a proof is running");
    if (int rc = bj::pipeline_release_workspace(ctx)) return rc;
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) BJ_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (int rc = bj::arena_drop_slabs(ctx)) return rc;
    if (ctx->arena) BJ_HIP(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_elems = ctx->arena_off = 0;
    ctx->arena_learned = 0;
    if (ctx->d_scratch) BJ_HIP(ctx, hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr;
    ctx->scratch_elems = 0;
    if (ctx->wit_stage) BJ_HIP(ctx, hipFree(ctx->wit_stage));
    ctx->wit_stage = nullptr;
    ctx->wit_stage_elems = 0;
    return BJ_OK;
}

int bj_ctx_set_stream(bj_ctx *ctx, void *hip_stream) {
    if (!ctx) return BJ_ERR_INVALID_ARG;
    if (ctx->ring_inflight) {   // staged copies are ordered on the old stream
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->ring_inflight = 0;
    }
    ctx->stream = (hipStream_t)hip_stream;
    return BJ_OK;
}

const char *bj_last_error(const bj_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bj_sync(bj_ctx *ctx) {
    if (int rc = bind(ctx)) return rc;
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJ_OK;
}

int bj_malloc(bj_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!d_ptr) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_malloc: null out pointer");
    if (int rc = bind(ctx)) return rc;
    BJ_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 8));
    return BJ_OK;
}
int bj_free(bj_ctx *ctx, void *d_ptr) {
    if (int rc = bind(ctx)) return rc;
    if (d_ptr) {
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        BJ_HIP(ctx, hipFree(d_ptr));
    }
    return BJ_OK;
}
int bj_memcpy_h2d(bj_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (int rc = bind(ctx)) return rc;
    if (!bytes) return BJ_OK;
    if (!d_dst || !h_src) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_memcpy_h2d: null pointer");
    BJ_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJ_OK;
}
int bj_memcpy_d2h(bj_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (int rc = bind(ctx)) return rc;
    if (!bytes) return BJ_OK;
    if (!h_dst || !d_src) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_memcpy_d2h: null pointer");
    if (bytes <= bj::RING_MAX_BLOCK) {   // small block: land in pinned memory (a pageable destination costs a staging pass in the runtime)
        if (!ctx->h_ring) BJ_HIP(ctx, hipHostMalloc((void **)&ctx->h_ring, bj::RING_BYTES, hipHostMallocDefault));
        BJ_HIP(ctx, hipMemcpyAsync(ctx->h_ring, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));   // runs after every staged upload
        BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(h_dst, ctx->h_ring, bytes);
        ctx->ring_off = ctx->ring_inflight = 0;
        return BJ_OK;
    }
    BJ_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ring_inflight = 0;
    return BJ_OK;
}

int bj_memcpy_d2d(bj_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    if (int rc = bind(ctx)) return rc;
    if (!bytes) return BJ_OK;
    if (!d_dst || !d_src) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_memcpy_d2d: null pointer");
    BJ_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJ_OK;
}

int bj_ctx_set_tree_hasher(bj_ctx *ctx, int hasher) {
    if (int rc = bind(ctx)) return rc;
    if (hasher < BJ_HASHER_POSEIDON2 || hasher > BJ_HASHER_KECCAK256)
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_ctx_set_tree_hasher: unknown hasher %d", hasher);
    ctx->hasher = hasher;
    return BJ_OK;
}

int bj_timer_start(bj_ctx *ctx) {
    if (int rc = bind(ctx)) return rc;
    BJ_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return BJ_OK;
}
int bj_timer_stop_ms(bj_ctx *ctx, float *ms) {
    if (int rc = bind(ctx)) return rc;
    if (!ms) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_timer_stop_ms: null out pointer");
    BJ_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    BJ_HIP(ctx, hipEventSynchronize(ctx->ev1));
    BJ_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return BJ_OK;
}

// ------------------------------------------------------------------------------------------------- NTT family

static int check_ntt_args(bj_ctx *ctx, const char *fn, const void *in, const void *out, unsigned log_n,
                          unsigned n_cols, size_t col_stride) {
    if (!in || !out) return fail(ctx, BJ_ERR_INVALID_ARG, "%s: null device pointer", fn);
    if (log_n > 32) return fail(ctx, BJ_ERR_INVALID_ARG, "%s: log_n %u exceeds the field's two-adicity (32)", fn, log_n);
    if (log_n > 30) return fail(ctx, BJ_ERR_UNSUPPORTED, "%s: log_n %u > 30 not supported", fn, log_n);
    if (n_cols > 65535) return fail(ctx, BJ_ERR_UNSUPPORTED, "%s: more than 65535 columns per call", fn);
    if (n_cols > 1 && col_stride < ((size_t)1 << log_n))
        return fail(ctx, BJ_ERR_INVALID_ARG, "%s: col_stride smaller than the column length", fn);
    return BJ_OK;
}

int bj_ntt_forward_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols,
                         size_t col_stride, uint64_t coset) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_ntt_forward_batch", d_in, d_out, log_n, n_cols, col_stride)) return rc;
    if (int rc = ensure_twiddles(ctx, log_n, false)) return rc;
    coset = gl::canon(coset);
    if (coset == 0) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_ntt_forward_batch: coset shift must be non-zero");
    const u64 *scales = nullptr;
    if (coset != 1 && log_n > 0) {
        bj::launch_round_scales(ctx->d_small + 64, &coset, 1, log_n, ctx->stream);
        scales = ctx->d_small + 64;
    }
    bj::launch_ntt_passes(d_in, d_out, ctx->tw_fwd, scales, log_n, n_cols, 1, col_stride, col_stride, ctx->stream, bj::front_table(ctx));
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_intt_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols,
                  size_t col_stride, uint64_t coset) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_intt_batch", d_in, d_out, log_n, n_cols, col_stride)) return rc;
    if (int rc = ensure_twiddles(ctx, log_n, true)) return rc;
    coset = gl::canon(coset);
    if (coset == 0) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_intt_batch: coset shift must be non-zero");
    const size_t n = (size_t)1 << log_n;
    // butterflies with inverse twiddles into scratch (bit-reversed), then un-reverse + scale into d_out — in groups of columns whose
    // scratch stays below 1 GiB (the groups of a wide batch run back to back on the stream; a 2^23-row witness would otherwise
    // keep 6 GB of scratch per context: eight sharded ranks on one device could not afford it)
    const size_t cap_elems = (size_t)1 << 27;
    unsigned group = n >= cap_elems ? 1u : (unsigned)(cap_elems / n);
    if (group > n_cols) group = n_cols;
    if (int rc = ensure_scratch(ctx, (size_t)group * n)) return rc;
    const u64 n_inv = log_n ? gl::inv(gl::canon((u64)n % gl::P)) : 1;
    const u64 step = coset == 1 ? 1 : gl::inv(coset);
    for (unsigned c0 = 0; c0 < n_cols; c0 += group) {
        const unsigned nc = n_cols - c0 < group ? n_cols - c0 : group;
        bj::launch_ntt_passes(d_in + (size_t)c0 * col_stride, ctx->d_scratch, ctx->tw_inv, nullptr, log_n, nc, 1, col_stride, n, ctx->stream,
                              bj::front_table(ctx));
        bj::launch_bitrev_scale(ctx->d_scratch, d_out + (size_t)c0 * col_stride, log_n, nc, n, col_stride, n_inv, step, ctx->stream);
    }
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

extern "C++" {
namespace bj {
// LDE of cosets [coset_begin, coset_begin+coset_count) of 2^log_lde with explicit input/output column strides;
// output column c holds its cosets back to back starting at d_out + c*out_col_stride.
int lde_cosets_strided(bj_ctx *ctx, const u64 *d_mono, size_t in_col_stride, u64 *d_out, size_t out_col_stride,
                       unsigned log_n, unsigned n_cols, unsigned log_lde, unsigned coset_begin, unsigned coset_count, bool tiled_in) {
    if (int rc = ensure_twiddles(ctx, log_n, false)) return rc;
    if (tiled_in && !bj::ntt_two_pass_applies(d_mono, d_out, log_n, coset_count, in_col_stride, out_col_stride))
        return fail(ctx, BJ_ERR_UNSUPPORTED, "LDE of tiled monomials: 2^22-word columns on 16-byte boundaries with the two-pass plan enabled only");
    u64 shifts[64];
    u64 w = gl::omega(log_n + log_lde);
    for (unsigned i = 0; i < coset_count; i++)
        shifts[i] = gl::mul(gl::GEN, gl::pow(w, gl::bitrev32(coset_begin + i, log_lde)));  // utils.rs:345-346, 370-373
    bj::launch_round_scales(ctx->d_small + 64, shifts, coset_count, log_n ? log_n : 1, ctx->stream);
    bj::launch_ntt_passes(d_mono, d_out, ctx->tw_fwd, log_n ? ctx->d_small + 64 : nullptr, log_n, n_cols, coset_count,
                          in_col_stride, out_col_stride, ctx->stream, bj::front_table(ctx), tiled_in);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

// The monomial layout bj_prove keeps for 2^log_n-row columns: tiled (ntt_r16.hip) where the two-pass plan runs, natural elsewhere.
bool mono_tiled(unsigned log_n) { return log_n == 22 && bj::env().ntt_two_pass && bj::env().mono_tiled && !bj::env().ntt_generic; }

// ifft_natural_to_natural (fft/mod.rs:464-491) on the main domain with the result left in the TILED layout: front pass into scratch,
// last pass storing the bit-reversed, 1/n-scaled positions directly — two HBM passes, no bit-reversal pass.  Columns on 16-byte
// boundaries; d_in may equal d_out.
int intt_to_tiled(bj_ctx *ctx, const u64 *d_in, size_t in_col_stride, u64 *d_out, size_t out_col_stride, unsigned log_n, unsigned n_cols) {
    if (n_cols == 0) return BJ_OK;
    if (!mono_tiled(log_n) || ((uintptr_t)d_in % 16) || ((uintptr_t)d_out % 16) || in_col_stride % 2 || out_col_stride % 2)
        return fail(ctx, BJ_ERR_UNSUPPORTED, "inverse transform into the tiled layout: 2^22-word columns on 16-byte boundaries only");
    if (int rc = ensure_twiddles(ctx, log_n, true)) return rc;
    const size_t n = (size_t)1 << log_n;
    const size_t cap_elems = (size_t)1 << 27;
    unsigned group = (unsigned)(cap_elems / n);
    if (group > n_cols) group = n_cols;
    if (int rc = ensure_scratch(ctx, (size_t)group * n)) return rc;
    const u64 n_inv = gl::inv(gl::canon((u64)n % gl::P));
    if (!ctx->tw_inv_scaled22) {   // a table for 2^22 is a prefix of every larger one: built once per context, whatever tw_inv grows to
        BJ_HIP(ctx, hipMalloc((void **)&ctx->tw_inv_scaled22, (n / 2) * sizeof(u64)));
        bj::launch_scale_table(ctx->tw_inv, ctx->tw_inv_scaled22, n / 2, n_inv, ctx->stream);
    }
    for (unsigned c0 = 0; c0 < n_cols; c0 += group) {
        const unsigned nc = n_cols - c0 < group ? n_cols - c0 : group;
        bj::launch_ntt_front10(d_in + (size_t)c0 * in_col_stride, ctx->d_scratch, ctx->tw_inv, nullptr, bj::front_table(ctx), log_n, nc, 1,
                               in_col_stride, n, ctx->stream);
        bj::launch_ntt_local12_pair_tiled(ctx->d_scratch, d_out + (size_t)c0 * out_col_stride, ctx->tw_inv, ctx->tw_inv_scaled22, n_inv, nc, n, out_col_stride, ctx->stream);
    }
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}
}  // namespace bj
}  // extern "C++"

int bj_monomials_tiled(unsigned log_n) { return bj::mono_tiled(log_n) ? 1 : 0; }

int bj_intt_batch_tiled(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols, size_t col_stride) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_intt_batch_tiled", d_in, d_out, log_n, n_cols, col_stride)) return rc;
    return bj::intt_to_tiled(ctx, d_in, col_stride, d_out, col_stride, log_n, n_cols);
}

int bj_lde_cosets_batch_tiled(bj_ctx *ctx, const uint64_t *d_mono_tiled, size_t col_stride, uint64_t *d_out, unsigned log_n,
                              unsigned n_cols, unsigned log_lde, unsigned coset_begin, unsigned coset_count) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0 || coset_count == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_lde_cosets_batch_tiled", d_mono_tiled, d_out, log_n, n_cols, col_stride)) return rc;
    if (log_lde == 0 || log_lde > 6 || log_n + log_lde > 32) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch_tiled: lde factor must be 2..64 inside the two-adicity");
    const unsigned L = 1u << log_lde;
    if (coset_begin >= L || coset_count > L - coset_begin)
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch_tiled: coset range outside [0, lde_factor)");
    if (d_out == d_mono_tiled) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch_tiled: output must not alias the monomials");
    return bj::lde_cosets_strided(ctx, d_mono_tiled, col_stride, d_out, ((size_t)coset_count) << log_n, log_n, n_cols, log_lde,
                                  coset_begin, coset_count, true);
}

int bj_tiled_permute_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols, size_t col_stride,
                           int to_tiled) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_tiled_permute_batch", d_in, d_out, log_n, n_cols, col_stride)) return rc;
    if (log_n != 22) return fail(ctx, BJ_ERR_UNSUPPORTED, "bj_tiled_permute_batch: the tiled layout is defined for 2^22-word columns");
    if (d_in == d_out) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_tiled_permute_batch: out of place only");
    bj::launch_tiled_permute(d_in, d_out, n_cols, col_stride, col_stride, to_tiled != 0, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_lde_cosets_batch(bj_ctx *ctx, const uint64_t *d_mono, size_t col_stride, uint64_t *d_out, unsigned log_n,
                        unsigned n_cols, unsigned log_lde, unsigned coset_begin, unsigned coset_count) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0 || coset_count == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_lde_cosets_batch", d_mono, d_out, log_n, n_cols, col_stride)) return rc;
    if (log_lde == 0 || log_lde > 6) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch: lde factor must be 2..64");
    if (log_n + log_lde > 32) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch: LDE domain exceeds two-adicity");
    const unsigned L = 1u << log_lde;
    if (coset_begin >= L || coset_count > L - coset_begin)
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch: coset range outside [0, lde_factor)");
    if (d_out == d_mono) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_cosets_batch: output must not alias the monomials");
    return bj::lde_cosets_strided(ctx, d_mono, col_stride, d_out, ((size_t)coset_count) << log_n, log_n, n_cols, log_lde,
                                  coset_begin, coset_count);
}

int bj_lde_batch(bj_ctx *ctx, const uint64_t *d_mono, size_t col_stride, uint64_t *d_out, unsigned log_n,
                 unsigned n_cols, unsigned log_lde) {
    if (log_lde == 0 || log_lde > 6) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_lde_batch: lde factor must be 2..64");
    return bj_lde_cosets_batch(ctx, d_mono, col_stride, d_out, log_n, n_cols, log_lde, 0, 1u << log_lde);
}

int bj_trace_to_lde_batch(bj_ctx *ctx, uint64_t *d_cols, size_t col_stride, uint64_t *d_out, unsigned log_n,
                          unsigned n_cols, unsigned log_lde) {
    if (int rc = bj_intt_batch(ctx, d_cols, d_cols, log_n, n_cols, col_stride, 1)) return rc;
    return bj_lde_batch(ctx, d_cols, col_stride, d_out, log_n, n_cols, log_lde);
}

int bj_bitreverse_batch(bj_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, unsigned log_n, unsigned n_cols,
                        size_t col_stride) {
    if (int rc = bind(ctx)) return rc;
    if (n_cols == 0) return BJ_OK;
    if (int rc = check_ntt_args(ctx, "bj_bitreverse_batch", d_in, d_out, log_n, n_cols, col_stride)) return rc;
    const size_t n = (size_t)1 << log_n;
    if (d_in == d_out) {
        if (int rc = ensure_scratch(ctx, (size_t)n_cols * n)) return rc;
        bj::launch_bitrev_scale(d_in, ctx->d_scratch, log_n, n_cols, col_stride, n, 1, 1, ctx->stream);
        if (col_stride == n) {
            BJ_HIP(ctx, hipMemcpyAsync(d_out, ctx->d_scratch, (size_t)n_cols * n * sizeof(u64),
                                       hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            BJ_HIP(ctx, hipMemcpy2DAsync(d_out, col_stride * sizeof(u64), ctx->d_scratch, n * sizeof(u64),
                                         n * sizeof(u64), n_cols, hipMemcpyDeviceToDevice, ctx->stream));
        }
    } else {
        bj::launch_bitrev_scale(d_in, d_out, log_n, n_cols, col_stride, col_stride, 1, 1, ctx->stream);
    }
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_field_op_batch(bj_ctx *ctx, int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, size_t n) {
    if (int rc = bind(ctx)) return rc;
    if (n == 0) return BJ_OK;
    if (op < BJ_FIELD_ADD || op > BJ_FIELD_EXT2_MUL_LAZY) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_field_op_batch: unknown operator");
    if ((op == BJ_FIELD_BUTTERFLY || op == BJ_FIELD_ADDSUB) && (n & 1))
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_field_op_batch: the butterfly operators take an even number of pairs");
    const bool unary = op == BJ_FIELD_SQUARE || op == BJ_FIELD_INVERSE || op == BJ_FIELD_ADDSUB;
    if (!d_a || !d_out || (!unary && !d_b)) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_field_op_batch: null device pointer");
    bj::launch_field_op(op, d_a, unary ? nullptr : d_b, d_out, n, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_canonicalize(bj_ctx *ctx, uint64_t *d_data, size_t n) {
    if (int rc = bind(ctx)) return rc;
    if (n == 0) return BJ_OK;
    if (!d_data) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_canonicalize: null device pointer");
    bj::launch_canonicalize(d_data, n, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

static int host_roundtrip(bj_ctx *ctx, uint64_t *h, unsigned log_n, unsigned n_cols, uint64_t coset, bool inverse) {
    if (int rc = bind(ctx)) return rc;
    if (!h) return fail(ctx, BJ_ERR_INVALID_ARG, "host NTT: null host pointer");
    if (log_n > 30) return fail(ctx, BJ_ERR_UNSUPPORTED, "host NTT: log_n > 30");
    size_t bytes = ((size_t)n_cols << log_n) * sizeof(u64);
    if (!bytes) return BJ_OK;
    u64 *d = nullptr;
    BJ_HIP(ctx, hipMalloc((void **)&d, bytes));
    int rc = bj_memcpy_h2d(ctx, d, h, bytes);
    if (!rc)
        rc = inverse ? bj_intt_batch(ctx, d, d, log_n, n_cols, (size_t)1 << log_n, coset)
                     : bj_ntt_forward_batch(ctx, d, d, log_n, n_cols, (size_t)1 << log_n, coset);
    if (!rc) rc = bj_memcpy_d2h(ctx, h, d, bytes);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
}
int bj_ntt_forward_host(bj_ctx *ctx, uint64_t *h_inout, unsigned log_n, unsigned n_cols, uint64_t coset) {
    return host_roundtrip(ctx, h_inout, log_n, n_cols, coset, false);
}
int bj_intt_host(bj_ctx *ctx, uint64_t *h_inout, unsigned log_n, unsigned n_cols, uint64_t coset) {
    return host_roundtrip(ctx, h_inout, log_n, n_cols, coset, true);
}

// ------------------------------------------------------------------------------------------- Poseidon2 Merkle

size_t bj_merkle_tree_digests(size_t num_leaves, size_t cap_size) { return 2 * num_leaves - cap_size; }

static int check_tree_args(bj_ctx *ctx, const char *fn, size_t num_leaves, size_t cap_size, const void *d_tree) {
    if (!d_tree) return fail(ctx, BJ_ERR_INVALID_ARG, "%s: null tree pointer", fn);
    if (!is_pow2(num_leaves) || !is_pow2(cap_size))
        return fail(ctx, BJ_ERR_INVALID_ARG, "%s: num_leaves and cap_size must be powers of two", fn);
    if (cap_size > num_leaves) return fail(ctx, BJ_ERR_INVALID_ARG, "%s: cap_size larger than the tree", fn);
    return BJ_OK;
}

int bj_merkle_tree_nodes(bj_ctx *ctx, uint64_t *d_tree, size_t num_leaves, size_t cap_size) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_tree_args(ctx, "bj_merkle_tree_nodes", num_leaves, cap_size, d_tree)) return rc;
    bj::launch_tree_node_layers(ctx->hasher, d_tree, num_leaves, cap_size, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_merkle_tree_build(bj_ctx *ctx, const uint64_t *d_cols, size_t col_stride, unsigned n_cols, size_t num_leaves,
                         size_t cap_size, uint64_t *d_tree) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_tree_args(ctx, "bj_merkle_tree_build", num_leaves, cap_size, d_tree)) return rc;
    if (!d_cols || n_cols == 0) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_build: no columns");
    if (n_cols > 1 && col_stride < num_leaves)
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_build: col_stride smaller than num_leaves");
    bj::launch_tree_leaves(ctx->hasher, d_cols, col_stride, nullptr, n_cols, num_leaves, d_tree, ctx->stream);
    bj::launch_tree_node_layers(ctx->hasher, d_tree, num_leaves, cap_size, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_merkle_tree_build_ptrs(bj_ctx *ctx, const uint64_t *const *h_col_ptrs, unsigned n_cols, size_t num_leaves,
                              size_t cap_size, uint64_t *d_tree) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_tree_args(ctx, "bj_merkle_tree_build_ptrs", num_leaves, cap_size, d_tree)) return rc;
    if (!h_col_ptrs || n_cols == 0) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_build_ptrs: no columns");
    for (unsigned c = 0; c < n_cols; c++)
        if (!h_col_ptrs[c]) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_build_ptrs: null column %u", c);
    if (n_cols > ctx->d_ptrs_cap) {
        if (ctx->d_ptrs) {
            BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
            BJ_HIP(ctx, hipFree((void *)ctx->d_ptrs));
            ctx->d_ptrs = nullptr;
            ctx->d_ptrs_cap = 0;
        }
        size_t cap = n_cols < 1024 ? 1024 : n_cols;
        BJ_HIP(ctx, hipMalloc((void **)&ctx->d_ptrs, cap * sizeof(u64 *)));
        ctx->d_ptrs_cap = cap;
    }
    if (int rc = bj::h2d_async(ctx, (void *)ctx->d_ptrs, h_col_ptrs, n_cols * sizeof(u64 *))) return rc;  // staged: the caller's array may be transient
    bj::launch_tree_leaves(ctx->hasher, nullptr, 0, ctx->d_ptrs, n_cols, num_leaves, d_tree, ctx->stream);
    bj::launch_tree_node_layers(ctx->hasher, d_tree, num_leaves, cap_size, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_merkle_tree_build_chunked(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, size_t len,
                                 unsigned log_elems_per_leaf, size_t cap_size, uint64_t *d_tree) {
    if (int rc = bind(ctx)) return rc;
    if (!d_c0 || !d_c1) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_build_chunked: null source");
    if (!is_pow2(len) || log_elems_per_leaf > 6 || ((size_t)1 << log_elems_per_leaf) > len)
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_build_chunked: bad length / elements per leaf");
    size_t num_leaves = len >> log_elems_per_leaf;
    if (int rc = check_tree_args(ctx, "bj_merkle_tree_build_chunked", num_leaves, cap_size, d_tree)) return rc;
    bj::launch_tree_leaves_chunked(ctx->hasher, d_c0, d_c1, 2, log_elems_per_leaf, num_leaves, d_tree, ctx->stream);
    bj::launch_tree_node_layers(ctx->hasher, d_tree, num_leaves, cap_size, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_merkle_tree_cap(bj_ctx *ctx, const uint64_t *d_tree, size_t num_leaves, size_t cap_size, uint64_t *h_cap) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_tree_args(ctx, "bj_merkle_tree_cap", num_leaves, cap_size, d_tree)) return rc;
    if (!h_cap) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_cap: null host pointer");
    return bj_memcpy_d2h(ctx, h_cap, d_tree + 4 * (2 * num_leaves - 2 * cap_size), 4 * cap_size * sizeof(u64));
}

int bj_merkle_tree_proof(bj_ctx *ctx, const uint64_t *d_tree, size_t num_leaves, size_t cap_size, size_t idx,
                         uint64_t *h_leaf_digest, uint64_t *h_path) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_tree_args(ctx, "bj_merkle_tree_proof", num_leaves, cap_size, d_tree)) return rc;
    if (idx >= num_leaves) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_proof: leaf index out of range");
    if (!h_leaf_digest || (!h_path && num_leaves > cap_size))
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_merkle_tree_proof: null host pointer");
    BJ_HIP(ctx, hipMemcpyAsync(h_leaf_digest, d_tree + 4 * idx, 32, hipMemcpyDeviceToHost, ctx->stream));
    const u64 *layer = d_tree;
    size_t len = num_leaves, depth = 0;
    while (len > cap_size) {
        BJ_HIP(ctx, hipMemcpyAsync(h_path + 4 * depth, layer + 4 * (idx ^ 1), 32, hipMemcpyDeviceToHost, ctx->stream));
        layer += 4 * len;
        len /= 2;
        idx >>= 1;
        depth++;
    }
    BJ_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BJ_OK;
}

int bj_poseidon2_permute(bj_ctx *ctx, uint64_t *d_states, size_t n_states) {
    if (int rc = bind(ctx)) return rc;
    if (n_states == 0) return BJ_OK;
    if (!d_states) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_poseidon2_permute: null device pointer");
    bj::launch_poseidon2_permute_states(d_states, n_states, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

// --------------------------------------------------------------------------------------------------------- FRI

int bj_fri_fold(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, size_t len, uint64_t *d_o0, uint64_t *d_o1,
                unsigned log_full, uint64_t coset_inv, uint64_t ch0, uint64_t ch1) {
    if (int rc = bind(ctx)) return rc;
    if (!d_c0 || !d_c1 || !d_o0 || !d_o1) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold: null device pointer");
    if (!is_pow2(len) || len < 2) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold: length must be a power of two >= 2");
    if (log_full > 32 || len > ((size_t)1 << log_full))
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold: array longer than the initial domain");
    if (int rc = ensure_twiddles(ctx, log_full, true)) return rc;
    (void)log2_exact;
    bj::launch_fri_fold(d_c0, d_c1, len, d_o0, d_o1, ctx->tw_inv, coset_inv, ch0, ch1, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

int bj_fri_fold_step(bj_ctx *ctx, const uint64_t *d_c0, const uint64_t *d_c1, size_t len, unsigned k, uint64_t *d_o0,
                     uint64_t *d_o1, unsigned log_full, uint64_t coset_inv, uint64_t ch0, uint64_t ch1) {
    if (int rc = bind(ctx)) return rc;
    if (!d_c0 || !d_c1 || !d_o0 || !d_o1) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold_step: null device pointer");
    if (k < 1 || k > 3) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold_step: k must be 1..3");
    if (!is_pow2(len) || (len >> k) == 0) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold_step: bad length");
    if (log_full > 32 || len > ((size_t)1 << log_full))
        return fail(ctx, BJ_ERR_INVALID_ARG, "bj_fri_fold_step: array longer than the initial domain");
    if (int rc = ensure_twiddles(ctx, log_full, true)) return rc;
    bj::launch_fri_fold_step(d_c0, d_c1, len, k, d_o0, d_o1, ctx->tw_inv, coset_inv, ch0, ch1, ctx->stream);
    BJ_CHECK_LAUNCH(ctx);
    return BJ_OK;
}

}  // extern "C"
