// In-library transport of the sharded prover: ONE RCCL all-gather per exchange, enqueued on the context's HIP stream straight
// on the prover's device buffers (no staging copies, no host trampoline, no device-wide synchronisation) — north_star's "a single
// RCCL all-gather over xGMI for the tree root and FRI commitments" (SURVEY §8e).  librccl is loaded at run time (dlopen, local
// scope): hosts that never shard need no communication library, and a process that already carries a copy (PyTorch ships one)
// shares it.  The unique id travels between the processes by whatever channel the host has (MPI, a file, torch.distributed's
// store): bj_rccl_unique_id on rank 0, then bj_comm_rccl_create on every rank (collective).
#include "ctx.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
struct UniqueId {
    char internal[BJ_RCCL_UNIQUE_ID_BYTES];
};
typedef void *Comm;
struct Api {
    void *handle = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};
Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const std::string rccl_lib = bj::env().rccl_lib;
        const char *env = rccl_lib.empty() ? nullptr : rccl_lib.c_str();
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            a.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (a.handle) break;
            a.error = dlerror();
        }
        if (!a.handle) return;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) {
            a.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            dlclose(a.handle);
            a.handle = nullptr;
        }
    });
    return a;
}
struct RcclComm {
    Comm comm = nullptr;
    unsigned rank = 0, world = 1;
    size_t calls = 0, bytes = 0;
};
constexpr int kNcclUint8 = 1;   // ncclDataType_t (rccl.h)

// bj_comm::all_gather_stream: stream-ordered, returns as soon as the collective is enqueued
int gather_on_stream(void *user, const void *d_send, void *d_recv, size_t bytes, void *stream) {
    RcclComm *c = (RcclComm *)user;
    c->calls++;
    c->bytes += bytes * c->world;
    return api().AllGather(d_send, d_recv, bytes, kNcclUint8, c->comm, (hipStream_t)stream);
}
// bj_comm::all_gather (the synchronous contract of the host-callback transport): the same collective on the null stream
int gather_blocking(void *user, const void *d_send, void *d_recv, size_t bytes) {
    if (int rc = gather_on_stream(user, d_send, d_recv, bytes, nullptr)) return rc;
    return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -1;
}
}  // namespace

extern "C" {

int bj_rccl_available(void) { return api().handle ? 1 : 0; }

int bj_rccl_unique_id(void *out_id) {
    if (!out_id) return BJ_ERR_INVALID_ARG;
    Api &a = api();
    if (!a.handle) return BJ_ERR_UNSUPPORTED;
    UniqueId id;
    if (a.GetUniqueId(&id) != 0) return BJ_ERR_HIP;
    std::memcpy(out_id, &id, sizeof(id));
    return BJ_OK;
}

int bj_comm_rccl_create(bj_ctx *ctx, const void *unique_id, unsigned rank, unsigned world, bj_comm *out) {
    if (int rc = bj::bind(ctx)) return rc;
    if (!unique_id || !out || world == 0 || rank >= world) return bj::fail(ctx, BJ_ERR_INVALID_ARG, "bj_comm_rccl_create: bad arguments");
    Api &a = api();
    if (!a.handle) return bj::fail(ctx, BJ_ERR_UNSUPPORTED, "bj_comm_rccl_create: librccl could not be loaded (%s); set BJ_RCCL_LIB", a.error.c_str());
    UniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    RcclComm *c = new RcclComm();
    c->rank = rank;
    c->world = world;
    const int r = a.CommInitRank(&c->comm, (int)world, id, (int)rank);   // collective over the `world` processes; binds the current device
    if (r != 0) {
        const char *msg = a.GetErrorString ? a.GetErrorString(r) : "?";
        delete c;
        return bj::fail(ctx, BJ_ERR_HIP, "bj_comm_rccl_create: ncclCommInitRank failed: %s", msg);
    }
    std::memset(out, 0, sizeof(*out));
    out->rank = rank;
    out->world = world;
    out->all_gather = gather_blocking;
    out->all_gather_stream = gather_on_stream;
    out->user = c;
    return BJ_OK;
}

void bj_comm_rccl_destroy(bj_comm *comm) {
    if (!comm || comm->all_gather_stream != gather_on_stream || !comm->user) return;
    RcclComm *c = (RcclComm *)comm->user;
    if (c->comm) (void)api().CommDestroy(c->comm);
    delete c;
    std::memset(comm, 0, sizeof(*comm));
}

int bj_comm_rccl_stats(const bj_comm *comm, size_t *calls, size_t *bytes_received) {
    if (!comm || comm->all_gather_stream != gather_on_stream || !comm->user) return BJ_ERR_INVALID_ARG;
    const RcclComm *c = (const RcclComm *)comm->user;
    if (calls) *calls = c->calls;
    if (bytes_received) *bytes_received = c->bytes;
    return BJ_OK;
}

}  // extern "C"
