// bj_prove_async / bj_proof_wait: two proofs in flight on ONE device from ONE host thread (include/boojum_hip.h).
//
// The call site replaced is a host loop over witnesses around prove_cpu_basic (src/cs/implementations/prover.rs:153-168,
// convenience.rs:119-196).  A proof ends in a latency-bound tail — ~100 lane-parallel node layers, the FRI tail oracles, ~15
// transcript round trips between host and device — during which most of the chip idles, and it starts with a PCIe transfer
// during which the CUs only transform the columns that have landed.  Neither can be filled from inside the proof (the transcript
// serialises it); the next proof's head can fill both.  A context therefore owns two LANES: each a private sub-context (its own
// non-blocking HIP stream, copy stream, workspace arena, witness staging, twiddle tables, staging ring) driven by its own worker
// thread.  bj_prove_async hands the witness to the next lane in turn and returns; the worker runs the ordinary bj_prove on the
// lane's sub-context.  Nothing is shared between the lanes but the (read-only) setup, so every proof is the bytes bj_prove gives.
// At most two proofs are in flight: a third submission waits for the lane it is due on.
#include "ctx.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

struct bj_ticket {
    bj_ctx *parent = nullptr;
    unsigned lane = 0;
    const bj_setup *setup = nullptr;
    const uint64_t *h_variables = nullptr, *h_multiplicities = nullptr;
    std::vector<uint64_t> public_values;
    bool has_public = false;
    // result (written by the lane's worker under the lane's mutex)
    bool done = false;
    int rc = BJ_OK;
    bj_proof *proof = nullptr;
    std::string err;
};

namespace bj {

struct Lane {
    bj_ctx *sub = nullptr;
    hipStream_t stream = nullptr;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    bj_ticket *job = nullptr;      // submitted, not finished
    bool quit = false;
    Lane *sibling = nullptr;
    // when the proof in flight started, how long the last one took, and for which setup (for the stagger below)
    std::chrono::steady_clock::time_point started;
    double last_ms = 0;
    const bj_setup *running = nullptr, *last_setup = nullptr;

    bool busy() {
        std::lock_guard<std::mutex> lk(m);
        return job != nullptr;
    }
    // Two lanes that prove the same circuit take the same time, so whatever offset they start with stays: started together they stay in
    // phase — both hash at once, both sit in their latency-bound tails at once — and the second proof in flight fills nothing.  A lane
    // that is about to start within a quarter period of its sibling's start (same setup, period known from the last proof) therefore
    // waits until half a period after it; the device is busy with the sibling meanwhile, and from then on the lanes alternate.
    void stagger(const bj_setup *setup) {
        if (!sibling || !bj::env().async_stagger) return;
        double wait_ms = 0;
        {
            std::lock_guard<std::mutex> lk(sibling->m);
            const double period = sibling->last_setup == setup && sibling->last_ms > 0 ? sibling->last_ms : (last_setup == setup ? last_ms : 0);
            if (sibling->job && sibling->running == setup && period > 0) {
                const double off = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sibling->started).count();
                if (off < 0.25 * period) wait_ms = 0.5 * period - off;
            }
        }
        if (wait_ms > 0) std::this_thread::sleep_for(std::chrono::duration<double, std::milli>(wait_ms));
    }

    void run() {
        for (;;) {
            bj_ticket *t;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || (job && !job->done); });
                if (quit) return;
                t = job;
            }
            // What a lane does while its sibling proves, by witness size (measured, tools/async_small.py; BJ_ASYNC_MODE overrides):
            //   below 1 GiB (<= 2^20 rows of the bench geometry): bj_prove as is — the group-wise transfer hides behind the lane's own
            //   transforms — with the lanes STAGGERED half a period apart: 2^16 / 2^18 / 2^20 rows 1.10 / 1.18 / 1.09 x the serial rate
            //   (0.91 x at 2^20 when both lanes start together);
            //   from 1 GiB on: the whole witness first (55 ms of PCIe at 2^22 rows, under the sibling's kernels — which also sets the
            //   lanes apart), then the proof as on a resident witness: one leaf kernel instead of group-wise absorption; 1.03 x serial.
            unsigned v = 0, w = 0, ln = 0;
            (void)bj_setup_shape(t->setup, &ln, &v, &w, nullptr);
            const size_t witness_bytes = ((size_t)(v + w + 1) << ln) * 8;
            int mode = bj::env().async_mode;
            if (mode < 0) mode = witness_bytes >= ((size_t)1 << 30) ? 1 : 0;
            static const bool dbg = getenv("BJ_ASYNC_DEBUG") != nullptr;
            const auto t_pick = std::chrono::steady_clock::now();
            if (mode == 0) stagger(t->setup);
            {
                std::lock_guard<std::mutex> lk(m);
                started = std::chrono::steady_clock::now();
                running = t->setup;
            }
            bj_proof *p = nullptr;
            const uint64_t *pub = t->has_public ? t->public_values.data() : nullptr;
            const bool overlapped = sibling && sibling->busy() && mode != 0;
            const int rc = overlapped ? prove_host_copy_first(sub, t->setup, t->h_variables, t->h_multiplicities, pub, &p, mode)
                                      : bj_prove(sub, t->setup, t->h_variables, t->h_multiplicities, pub, &p);
            if (dbg)
                fprintf(stderr, "[lane %p] mode %d overlapped %d: picked up, ran %.1f ms (stagger %.1f ms)\n", (void *)this, mode, (int)overlapped,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - started).count(),
                        std::chrono::duration<double, std::milli>(started - t_pick).count());
            {
                std::lock_guard<std::mutex> lk(m);
                last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - started).count();
                last_setup = t->setup;
                running = nullptr;
                t->rc = rc;
                t->proof = p;
                if (rc) t->err = sub->err;
                t->done = true;
                job = nullptr;
            }
            cv.notify_all();
        }
    }
};

struct Pipeline {
    Lane lanes[2];
    unsigned next = 0;
    unsigned created = 0;
};

static int lane_start(bj_ctx *ctx, Lane &L) {
    if (int rc = bj_ctx_create(ctx->device, &L.sub)) return fail(ctx, rc, "bj_prove_async: creating a lane's context failed");
    if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) {
        bj_ctx_destroy(L.sub);
        L.sub = nullptr;
        return fail(ctx, BJ_ERR_HIP, "bj_prove_async: creating a lane's stream failed");
    }
    L.sub->stream = L.stream;
    L.sub->hasher = ctx->hasher;
    L.worker = std::thread([&L] { L.run(); });
    return BJ_OK;
}

void pipeline_destroy(bj_ctx *ctx) {
    Pipeline *P = ctx->pipe;
    if (!P) return;
    for (unsigned i = 0; i < P->created; i++) {      // first every worker ends (nothing of a lane may go while a sibling still proves) ...
        Lane &L = P->lanes[i];
        {
            std::unique_lock<std::mutex> lk(L.m);
            L.cv.wait(lk, [&] { return L.job == nullptr; });   // a proof in flight finishes first (its ticket stays valid)
            L.quit = true;
        }
        L.cv.notify_all();
        if (L.worker.joinable()) L.worker.join();
    }
    for (unsigned i = 0; i < P->created; i++) {      // ... then the contexts go
        Lane &L = P->lanes[i];
        if (L.sub) bj_ctx_destroy(L.sub);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    delete P;
    ctx->pipe = nullptr;
}

int pipeline_release_workspace(bj_ctx *ctx) {
    Pipeline *P = ctx->pipe;
    if (!P) return BJ_OK;
    for (unsigned i = 0; i < P->created; i++) {
        Lane &L = P->lanes[i];
        std::unique_lock<std::mutex> lk(L.m);
        if (L.job) return fail(ctx, BJ_ERR_INVALID_ARG, "bj_ctx_release_workspace: an asynchronous proof is running");
        if (int rc = bj_ctx_release_workspace(L.sub)) return fail(ctx, rc, "%s", L.sub->err.c_str());
    }
    return BJ_OK;
}

}  // namespace bj

extern "C" {

int bj_prove_async(bj_ctx *ctx, const bj_setup *setup, const uint64_t *h_variables, const uint64_t *h_multiplicities,
                   const uint64_t *h_public_values, bj_ticket **out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!out) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_async: null out pointer");
    *out = nullptr;
    if (!setup || !h_variables) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_async: null argument");
    unsigned n_pub = 0;
    if (int rc = bj_setup_shape(setup, nullptr, nullptr, nullptr, &n_pub)) return bj::fail(ctx, rc, "bj_prove_async: bad setup");
    if (bj::setup_world(setup) != 1)
        return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_prove_async: single-device setups only (the ranks of a sharded proof are concurrent already)");
    if (n_pub && !h_public_values) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_prove_async: public input values required");
    if (!ctx->pipe) {
        ctx->pipe = new bj::Pipeline();
        ctx->pipe->lanes[0].sibling = &ctx->pipe->lanes[1];
        ctx->pipe->lanes[1].sibling = &ctx->pipe->lanes[0];
    }
    bj::Pipeline *P = ctx->pipe;
    const unsigned li = P->next;
    bj::Lane &L = P->lanes[li];
    if (li >= P->created) {
        if (int rc = bj::lane_start(ctx, L)) return rc;
        P->created = li + 1;
    }
    bj_ticket *t = new bj_ticket();
    t->parent = ctx;
    t->lane = li;
    t->setup = setup;
    t->h_variables = h_variables;
    t->h_multiplicities = h_multiplicities;
    if (n_pub) {
        t->public_values.assign(h_public_values, h_public_values + n_pub);   // the caller's array may go away; the witness may not
        t->has_public = true;
    }
    {
        std::unique_lock<std::mutex> lk(L.m);
        L.cv.wait(lk, [&] { return L.job == nullptr; });   // at most two in flight: the proof this lane still runs comes first
        L.job = t;
    }
    L.cv.notify_all();
    P->next = (li + 1) % 2;
    *out = t;
    return BJ_OK;
}

int bj_proof_wait(bj_ticket *t, bj_proof **out) {
    if (!t) return BJ_ERR_INVALID_ARG;
    bj_ctx *ctx = t->parent;
    bj::Lane &L = ctx->pipe->lanes[t->lane];
    {
        std::unique_lock<std::mutex> lk(L.m);
        L.cv.wait(lk, [&] { return t->done; });
    }
    const int rc = t->rc;
    if (rc) ctx->err = t->err;
    if (out)
        *out = t->proof;
    else if (t->proof)
        bj_proof_destroy(t->proof);
    delete t;
    return rc;
}

int bj_proof_poll(const bj_ticket *t) {
    if (!t) return BJ_ERR_INVALID_ARG;
    bj::Lane &L = t->parent->pipe->lanes[t->lane];
    std::lock_guard<std::mutex> lk(L.m);
    return t->done ? 1 : 0;
}

}  // extern "C"
