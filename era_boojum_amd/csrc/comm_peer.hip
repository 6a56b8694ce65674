// Full-mesh peer transport for the BULK exchanges of a sharded proof ("direct full-mesh peer copies so all 7 xGMI
// links carry traffic").  A ring all-gather moves the 268 MB of quotient residues (and the 67 MB of the first folded FRI layer and
// of the DEEP numerator slices) over ONE link's bandwidth; on a fully connected node every rank can instead write its contribution
// straight into every peer's receive buffer, one copy per link, all links at once.  This transport wraps a base bj_comm (RCCL or a
// host callback) and handles the exchanges of at least `bulk_threshold` bytes per rank itself:
//   1. the proof's stream is drained (the contribution is complete; this rank's MAILBOX — a receive buffer of world slots owned by
//      the transport, grown on demand, never freed before the transport — is no longer read by the copy-out of the previous exchange);
//   2. every rank publishes the IPC handle of its mailbox through the host's control channel — a small blocking all-gather of
//      host bytes (MPI, a TCP store, torch.distributed's gloo group);
//   3. peers' mailboxes are mapped once per handle (hipIpcOpenMemHandle) and the rank copies its contribution to slot `rank` of
//      every peer's mailbox — world copies on world streams, i.e. one per link — and to its own slot of the receive buffer;
//      (handles, mappings and the vote happen once per exchange size; the mailbox has two halves used in turn);
//   4. when its copies have completed the rank enters a control exchange; when that returns every peer's copies into THIS
//      rank's mailbox have completed too, and one device copy on the proof's stream moves the world slots into the prover's buffer
//      (an HBM-speed pass over data that crossed the links at a tenth of that).
// The mailbox rather than the prover's own buffer is what peers map: that buffer lives in the proof arena, tens of GB in one
// allocation — mapping a 2.5 GB one into a sibling process did not return on the test box (smaller ones did) — and moves when the
// arena grows; a mailbox is sized by the exchange (<= 2 q n x 8 bytes: 268 MB at 2^22 rows) and stable across proofs.
// Everything below the threshold (cap fragments, values at z, query openings: latency-bound) goes to the base transport unchanged.
// The proof cannot tell the difference: same bytes in the same slots.  Not timed on a multi-GPU node by its author (one GPU
// here: ranks of the tests share the device, where IPC mapping of a sibling process's buffer works the same way); selectable
// beside RCCL (bench.py --bulk-transport, BJ_COMM_BULK=peer) and never the default.
#include "ctx.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
#define PEER_DBG(...)                                                         \
    do {                                                                      \
        if (getenv("BJ_PEER_DEBUG")) {                                        \
            fprintf(stderr, "[peer r%u] ", c->rank);                          \
            fprintf(stderr, __VA_ARGS__);                                     \
            fprintf(stderr, "\n");                                            \
            fflush(stderr);                                                   \
        }                                                                     \
    } while (0)
struct Mapped {
    hipIpcMemHandle_t handle;
    void *base = nullptr;
    size_t last_use = 0;
};
struct PeerRecord {               // what a rank publishes per bulk exchange
    hipIpcMemHandle_t handle;     // of its mailbox
    uint64_t capacity;            // bytes of the mailbox (>= world * bytes)
    uint64_t bytes;               // per-rank contribution (must agree across ranks)
    uint64_t ok;                  // 1: the mailbox exists and could be exported (a rank that could not sends 0 and everyone falls back)
};
constexpr size_t MAILBOX_MAX = (size_t)1 << 30;   // larger exchanges go to the base transport
struct PeerComm {
    bj_comm base;
    bj_host_exchange_fn exchange = nullptr;
    void *exchange_user = nullptr;
    unsigned rank = 0, world = 1;
    size_t threshold = 0;
    int device = 0;
    std::vector<hipStream_t> streams;            // one per peer
    std::vector<std::vector<Mapped>> mapped;     // per peer: mailboxes opened so far
    void *mailbox = nullptr;                     // this rank's receive slots, mapped by the peers
    size_t mailbox_bytes = 0;
    hipIpcMemHandle_t mailbox_handle;
    bool mailbox_ok = false;
    std::vector<void *> retired;                 // outgrown mailboxes: peers may still have them mapped, freed with the transport
    std::vector<void *> peer_box;                // per peer: its current mailbox as mapped here
    std::vector<size_t> peer_half;               // per peer: bytes of one half of it
    std::vector<std::pair<size_t, int>> sizes;   // per-rank sizes negotiated so far: 1 = peer copies, 2 = base transport
    size_t seq = 0;                              // bulk exchanges served by peer copies: parity selects the mailbox half
    size_t tick = 0;
    size_t bulk_calls = 0, bulk_bytes = 0, small_calls = 0, fallbacks = 0;
    std::string error;
};
constexpr size_t MAX_MAPPED_PER_PEER = 8;

int base_gather(PeerComm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t st) {
    if (c->base.all_gather_stream) return c->base.all_gather_stream(c->base.user, d_send, d_recv, bytes, st);
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    return c->base.all_gather(c->base.user, d_send, d_recv, bytes);
}

void *map_peer(PeerComm *c, unsigned p, const hipIpcMemHandle_t &h) {
    auto &v = c->mapped[p];
    for (auto &m : v)
        if (std::memcmp(&m.handle, &h, sizeof(h)) == 0) {
            m.last_use = ++c->tick;
            return m.base;
        }
    if (v.size() >= MAX_MAPPED_PER_PEER) {       // the peer has outgrown several mailboxes: unmap the one used longest ago (never the current one)
        size_t old = v.size();
        for (size_t i = 0; i < v.size(); i++)
            if (v[i].base != c->peer_box[p] && (old == v.size() || v[i].last_use < v[old].last_use)) old = i;
        if (old < v.size()) {
            (void)hipIpcCloseMemHandle(v[old].base);
            v.erase(v.begin() + (long)old);
        }
    }
    void *base = nullptr;
    if (hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    Mapped m;
    m.handle = h;
    m.base = base;
    m.last_use = ++c->tick;
    v.push_back(m);
    return base;
}

int peer_on_stream(void *user, const void *d_send, void *d_recv, size_t bytes, void *stream) {
    PeerComm *c = (PeerComm *)user;
    hipStream_t st = (hipStream_t)stream;
    if (bytes < c->threshold) {
        c->small_calls++;
        return base_gather(c, d_send, d_recv, bytes, st);
    }
    const unsigned W = c->world;
    PEER_DBG("bulk %zu bytes: draining the stream", bytes);
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    // A size is negotiated ONCE (handles published, mailboxes mapped, a vote that every rank could): all ranks see the same sequence
    // of sizes, so they take the same branch here without talking.  Afterwards an exchange of that size costs one control round
    // trip (the completion barrier).  The mailbox has two halves used in turn: a peer writes exchange k + 1 into the half that this
    // rank's copy-out of exchange k is not reading (that copy-out is drained before this rank enters the barrier of k + 1, i.e.
    // before anybody can start k + 2).
    const size_t need = bytes * W;
    int state = 0;                                  // 0: never seen, 1: peer copies, 2: base transport
    for (auto &kv : c->sizes)
        if (kv.first == bytes) state = kv.second;
    if (state == 2) {
        c->fallbacks++;
        return base_gather(c, d_send, d_recv, bytes, st);
    }
    if (state == 0) {
        PeerRecord mine;
        std::memset(&mine, 0, sizeof(mine));
        mine.bytes = bytes;
        // grow: a candidate allocation with a handle of its own; it replaces the mailbox only if EVERY rank's negotiation succeeds
        // (otherwise peers keep writing the sizes they negotiated earlier into the mailbox they mapped then)
        void *cand = nullptr;
        size_t cand_bytes = 0;
        hipIpcMemHandle_t cand_handle;
        bool cand_ok = false;
        if (2 * need <= MAILBOX_MAX && c->mailbox_bytes < 2 * need) {
            cand_bytes = (2 * need + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
            if (hipMalloc(&cand, cand_bytes) == hipSuccess) cand_ok = hipIpcGetMemHandle(&cand_handle, cand) == hipSuccess;
            else cand = nullptr;
            (void)hipGetLastError();
        }
        if (cand && cand_ok) {
            mine.handle = cand_handle;
            mine.capacity = cand_bytes;
            mine.ok = 1;
        } else if (!cand && c->mailbox && c->mailbox_ok && c->mailbox_bytes >= 2 * need) {
            mine.handle = c->mailbox_handle;
            mine.capacity = c->mailbox_bytes;
            mine.ok = 1;
        }
        if (const char *f = getenv("BJ_PEER_TEST_FAIL_RANK"))      // test hook: this rank pretends it could not export its mailbox
            if ((unsigned)atoi(f) == c->rank) mine.ok = 0;
        PEER_DBG("mailbox %p (%zu bytes), ok %d: exchanging", c->mailbox, c->mailbox_bytes, (int)mine.ok);
        std::vector<PeerRecord> all(W);
        if (c->exchange(c->exchange_user, &mine, all.data(), sizeof(PeerRecord))) return -2;
        bool ok = true;
        for (unsigned p = 0; p < W; p++) ok = ok && all[p].ok == 1 && all[p].bytes == bytes && all[p].capacity >= 2 * need;
        std::vector<void *> boxes(W, nullptr);
        for (unsigned p = 0; p < W && ok; p++) {
            if (p == c->rank) continue;
            PEER_DBG("mapping peer %u", p);
            boxes[p] = map_peer(c, p, all[p].handle);
            PEER_DBG("peer %u mapped at %p", p, boxes[p]);
            if (!boxes[p]) ok = false;
        }
        // the decision must be the same everywhere: a rank that could not map a peer says so in a second exchange
        uint64_t vote = ok ? 1 : 0;
        std::vector<uint64_t> votes(W);
        if (c->exchange(c->exchange_user, &vote, votes.data(), sizeof(vote))) return -2;
        for (unsigned p = 0; p < W; p++) ok = ok && votes[p] == 1;
        c->sizes.push_back({bytes, ok ? 1 : 2});
        if (ok) {                                   // commit: everyone writes to the mailboxes published in THIS negotiation from now on
            for (unsigned p = 0; p < W; p++)
                if (p != c->rank) {
                    c->peer_box[p] = boxes[p];
                    c->peer_half[p] = all[p].capacity / 2;
                }
            if (cand) {
                if (c->mailbox) c->retired.push_back(c->mailbox);
                c->mailbox = cand;
                c->mailbox_bytes = cand_bytes;
                c->mailbox_handle = cand_handle;
                c->mailbox_ok = true;
            }
        } else if (cand) {
            c->retired.push_back(cand);             // a peer may have mapped it already: freed with the transport
        }
        if (!ok) {
            c->fallbacks++;
            return base_gather(c, d_send, d_recv, bytes, st);
        }
    }
    const size_t half_sel = c->seq++ & 1;
    for (unsigned k = 0; k < W; k++) {               // start with the next rank: at any moment every link carries one copy
        const unsigned p = (c->rank + 1 + k) % W;
        char *dst = p == c->rank ? (char *)d_recv : (char *)c->peer_box[p] + half_sel * c->peer_half[p];   // the own slot goes straight home
        if (dst + (size_t)c->rank * bytes == (const char *)d_send) continue;                                // in-place gather: it is there already
        if (hipMemcpyAsync(dst + (size_t)c->rank * bytes, d_send, bytes, hipMemcpyDeviceToDevice, c->streams[p]) != hipSuccess) return -3;
    }
    PEER_DBG("copies queued");
    for (unsigned p = 0; p < W; p++)
        if (hipStreamSynchronize(c->streams[p]) != hipSuccess) return -3;
    PEER_DBG("copies done");
    uint64_t done = 1;
    std::vector<uint64_t> dones(W);
    if (c->exchange(c->exchange_user, &done, dones.data(), sizeof(done))) return -2;   // every rank's copies have landed
    const char *box = (const char *)c->mailbox + half_sel * (c->mailbox_bytes / 2);
    for (unsigned p = 0; p < W; p++) {           // mailbox -> receive buffer, stream-ordered from here on (two runs around the own slot)
        if (p == c->rank) continue;
        unsigned q = p;
        while (q + 1 < W && q + 1 != c->rank) q++;
        if (hipMemcpyAsync((char *)d_recv + (size_t)p * bytes, box + (size_t)p * bytes, (size_t)(q - p + 1) * bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return -3;
        p = q;
    }
    c->bulk_calls++;
    c->bulk_bytes += bytes * W;
    return 0;
}
int peer_blocking(void *user, const void *d_send, void *d_recv, size_t bytes) {
    PeerComm *c = (PeerComm *)user;
    if (bytes < c->threshold && !c->base.all_gather_stream) {
        c->small_calls++;
        return c->base.all_gather(c->base.user, d_send, d_recv, bytes);
    }
    if (int rc = peer_on_stream(user, d_send, d_recv, bytes, nullptr)) return rc;
    return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -1;
}
}  // namespace

extern "C" {

int bj_comm_peer_create(bj_ctx *ctx, const bj_comm *base, bj_host_exchange_fn exchange, void *exchange_user, size_t bulk_threshold_bytes,
                        bj_comm *out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!base || !out || !exchange || base->world < 2 || base->rank >= base->world || (!base->all_gather && !base->all_gather_stream))
        return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_comm_peer_create: a base transport of world >= 2 and a control exchange are required");
    PeerComm *c = new PeerComm();
    c->base = *base;
    c->exchange = exchange;
    c->exchange_user = exchange_user;
    c->rank = base->rank;
    c->world = base->world;
    c->threshold = bulk_threshold_bytes ? bulk_threshold_bytes : ((size_t)1 << 20);
    c->device = ctx->device;
    c->mapped.resize(c->world);
    c->peer_box.assign(c->world, nullptr);
    c->peer_half.assign(c->world, 0);
    for (unsigned p = 0; p < c->world; p++) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            for (hipStream_t t : c->streams) (void)hipStreamDestroy(t);
            delete c;
            return bj::fail(ctx, BJ_ERR_HIP, "bj_comm_peer_create: stream creation failed");
        }
        c->streams.push_back(s);
    }
    std::memset(out, 0, sizeof(*out));
    out->rank = c->rank;
    out->world = c->world;
    out->all_gather = peer_blocking;
    // stream-ordered for the small exchanges when the base is; the bulk path drains the stream itself
    out->all_gather_stream = peer_on_stream;
    out->user = c;
    return BJ_OK;
}

void bj_comm_peer_destroy(bj_comm *comm) {
    if (!comm || comm->all_gather_stream != peer_on_stream || !comm->user) return;
    PeerComm *c = (PeerComm *)comm->user;
    (void)hipSetDevice(c->device);
    for (auto &v : c->mapped)
        for (auto &m : v) (void)hipIpcCloseMemHandle(m.base);
    if (c->mailbox) (void)hipFree(c->mailbox);
    for (void *r : c->retired) (void)hipFree(r);
    for (hipStream_t s : c->streams) (void)hipStreamDestroy(s);
    delete c;
    std::memset(comm, 0, sizeof(*comm));
}

int bj_comm_peer_stats(const bj_comm *comm, size_t *bulk_calls, size_t *bulk_bytes_received, size_t *small_calls, size_t *fallbacks) {
    if (!comm || comm->all_gather_stream != peer_on_stream || !comm->user) return BJ_ERR_INVALID_ARG;
    const PeerComm *c = (const PeerComm *)comm->user;
    if (bulk_calls) *bulk_calls = c->bulk_calls;
    if (bulk_bytes_received) *bulk_bytes_received = c->bulk_bytes;
    if (small_calls) *small_calls = c->small_calls;
    if (fallbacks) *fallbacks = c->fallbacks;
    return BJ_OK;
}

}  // extern "C"
