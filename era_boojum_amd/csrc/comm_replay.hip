// Recorded-peer transport of the sharded prover: ONE rank of a `world`-rank proof runs ALONE on its GPU and every all-gather is
// served by a device-to-device copy of the buffer that collective produced in a real `world`-rank run of the same proof (every
// rank of a sharded proof sees the same gathered bytes, and proofs of one witness are deterministic).  What it is for: the
// critical path of one rank — its kernels, launches, host round trips, transcript work — measured on a single GPU, with the
// link time replaced by an HBM copy (bench.py --replay-world).  It is a bj_comm like any other: the prover cannot
// tell it from RCCL, the proof that comes out is bit for bit the single-GPU proof, and in verifying mode every contribution
// of the replayed rank is compared with the slice of it the recording holds.
#include "ctx.h"

#include <cstring>
#include <vector>

namespace {
struct ReplayComm {
    unsigned rank = 0, world = 1;
    std::vector<const void *> blobs;   // device buffers, world * bytes each, in call order
    std::vector<size_t> sizes;
    size_t n_setup = 0, n_cycle = 0;   // the first n_setup collectives once (bj_setup_create_sharded), then n_cycle per proof
    size_t calls = 0, bytes = 0, mismatches = 0;
    int verify = 0;
    std::vector<unsigned char> h_a, h_b;
    // capture (bj_comm_replay_capture): the FIRST collective past the recorded ones is not served — this rank's contribution
    // to it is copied out and the call fails with BJ_REPLAY_CAPTURED, which ends the proof.  That is how a recording is made
    // one collective at a time with ONE rank on the device at a time (era_boojum_amd/scale_replay.py): collective k of rank r is
    // what r contributes after k replayed ones, and every rank receives the same gathered bytes.
    void *d_capture = nullptr;
    size_t capture_capacity = 0, captured = 0;
};
constexpr int BJ_REPLAY_CAPTURED = 77;

int replay_on_stream(void *user, const void *d_send, void *d_recv, size_t bytes, void *stream) {
    ReplayComm *c = (ReplayComm *)user;
    size_t k = c->calls;
    hipStream_t st = (hipStream_t)stream;
    if (c->d_capture && k == c->blobs.size()) {   // the first collective nobody has recorded yet
        if (bytes > c->capture_capacity) return -5;
        if (hipMemcpyAsync(c->d_capture, d_send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return -4;
        if (hipStreamSynchronize(st) != hipSuccess) return -4;
        c->captured = bytes;
        return BJ_REPLAY_CAPTURED;
    }
    if (k >= c->n_setup && !c->d_capture) {
        if (!c->n_cycle) return -2;
        k = c->n_setup + (k - c->n_setup) % c->n_cycle;
    }
    if (k >= c->blobs.size() || c->sizes[k] != bytes * c->world) return -3;   // not the collective that was recorded here
    c->calls++;
    c->bytes += bytes * c->world;
    // without `verify` the first collective of every proof is compared all the same (a cap fragment: a few hundred bytes): a
    // recording of another witness or configuration of the same shape does not go unnoticed
    const bool first_of_proof = c->n_cycle && k == c->n_setup;
    if (c->verify || first_of_proof) {   // this rank's own contribution must be the slice the recording has for it
        c->h_a.resize(bytes);
        c->h_b.resize(bytes);
        if (hipMemcpyAsync(c->h_a.data(), d_send, bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return -4;
        if (hipMemcpyAsync(c->h_b.data(), (const char *)c->blobs[k] + (size_t)c->rank * bytes, bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
            return -4;
        if (hipStreamSynchronize(st) != hipSuccess) return -4;
        if (std::memcmp(c->h_a.data(), c->h_b.data(), bytes) != 0) c->mismatches++;
    }
    return hipMemcpyAsync(d_recv, c->blobs[k], bytes * c->world, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : -1;
}
int replay_blocking(void *user, const void *d_send, void *d_recv, size_t bytes) {
    if (int rc = replay_on_stream(user, d_send, d_recv, bytes, nullptr)) return rc;
    return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -1;
}
}  // namespace

extern "C" {

int bj_comm_replay_create(bj_ctx *ctx, unsigned rank, unsigned world, const void *const *d_gathered, const size_t *gathered_bytes,
                          size_t n_setup, size_t n_per_proof, int verify, bj_comm *out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!out || world < 2 || rank >= world || (n_setup + n_per_proof && (!d_gathered || !gathered_bytes)))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_comm_replay_create: bad arguments");
    ReplayComm *c = new ReplayComm();
    c->rank = rank;
    c->world = world;
    c->n_setup = n_setup;
    c->n_cycle = n_per_proof;
    c->verify = verify;
    for (size_t i = 0; i < n_setup + n_per_proof; i++) {
        if (!d_gathered[i] || gathered_bytes[i] == 0 || gathered_bytes[i] % world) {
            delete c;
            return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_comm_replay_create: recorded buffer %zu is empty or not a multiple of the world size", i);
        }
        c->blobs.push_back(d_gathered[i]);
        c->sizes.push_back(gathered_bytes[i]);
    }
    std::memset(out, 0, sizeof(*out));
    out->rank = rank;
    out->world = world;
    out->all_gather = replay_blocking;
    out->all_gather_stream = replay_on_stream;
    out->user = c;
    return BJ_OK;
}

void bj_comm_replay_destroy(bj_comm *comm) {
    if (!comm || comm->all_gather_stream != replay_on_stream || !comm->user) return;
    delete (ReplayComm *)comm->user;
    std::memset(comm, 0, sizeof(*comm));
}

int bj_comm_replay_capture(bj_comm *comm, void *d_capture, size_t capacity_bytes) {
    if (!comm || comm->all_gather_stream != replay_on_stream || !comm->user || !d_capture) return BJ_ERR_INVALID_ARG;
    ReplayComm *c = (ReplayComm *)comm->user;
    c->d_capture = d_capture;
    c->capture_capacity = capacity_bytes;
    c->captured = 0;
    return BJ_OK;
}
int bj_comm_replay_captured(const bj_comm *comm, size_t *bytes) {
    if (!comm || comm->all_gather_stream != replay_on_stream || !comm->user || !bytes) return BJ_ERR_INVALID_ARG;
    *bytes = ((const ReplayComm *)comm->user)->captured;
    return BJ_OK;
}
int bj_comm_replay_stats(const bj_comm *comm, size_t *calls, size_t *bytes_received, size_t *mismatches) {
    if (!comm || comm->all_gather_stream != replay_on_stream || !comm->user) return BJ_ERR_INVALID_ARG;
    const ReplayComm *c = (const ReplayComm *)comm->user;
    if (calls) *calls = c->calls;
    if (bytes_received) *bytes_received = c->bytes;
    if (mismatches) *mismatches = c->mismatches;
    return BJ_OK;
}

}  // extern "C"
