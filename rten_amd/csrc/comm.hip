// RCCL behind the boundary: communicator + broadcast for the one collective of the path -- the one-time
// broadcast of the prepacked weight arena from rank 0 to every GPU of the node over xGMI.  The reference has no
// analogue (single-process CPU executor, src/model.rs:308-550 callers); a Rust host binds these next to Model::load.
//
// librccl.so is loaded on first use with dlopen (a single-GPU process never maps it) and only the four entry points the
// path needs are resolved; the types below restate the public RCCL ABI (rccl.h: ncclUniqueId = 128 opaque bytes,
// ncclComm_t = opaque pointer, ncclResult_t 0 = success, ncclUint8 = 1).
#include <dlfcn.h>

#include <cstring>

#include "internal.h"

namespace {

struct UniqueId { char internal[RTEN_HIP_COMM_ID_BYTES]; };
typedef void *comm_t;
typedef int (*get_unique_id_fn)(UniqueId *);
typedef int (*comm_init_rank_fn)(comm_t *, int, UniqueId, int);
typedef int (*broadcast_fn)(const void *, void *, size_t, int, int, comm_t, hipStream_t);
typedef int (*comm_destroy_fn)(comm_t);
typedef const char *(*get_error_string_fn)(int);
constexpr int kNcclUint8 = 1;

struct Rccl {
    void *handle = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    broadcast_fn broadcast = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    get_error_string_fn error_string = nullptr;
    std::string load_error;
};

Rccl *rccl() {
    static Rccl *r = [] {
        Rccl *x = new Rccl();
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            x->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (x->handle) break;
        }
        if (!x->handle) {
            const char *e = dlerror();
            x->load_error = std::string("cannot load librccl.so: ") + (e ? e : "unknown error");
            return x;
        }
        x->get_unique_id = (get_unique_id_fn)dlsym(x->handle, "ncclGetUniqueId");
        x->comm_init_rank = (comm_init_rank_fn)dlsym(x->handle, "ncclCommInitRank");
        x->broadcast = (broadcast_fn)dlsym(x->handle, "ncclBroadcast");
        x->comm_destroy = (comm_destroy_fn)dlsym(x->handle, "ncclCommDestroy");
        x->error_string = (get_error_string_fn)dlsym(x->handle, "ncclGetErrorString");
        if (!x->get_unique_id || !x->comm_init_rank || !x->broadcast || !x->comm_destroy) x->load_error = "librccl.so lacks a required symbol";
        return x;
    }();
    return r;
}

int32_t rccl_fail(rten_hip_ctx *ctx, Rccl *r, int rc, const char *what) {
    return rten_set_error(ctx, RTEN_HIP_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, r->error_string ? r->error_string(rc) : "?");
}

} // namespace

struct rten_hip_comm {
    comm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

RTEN_EXPORT int32_t rten_hip_comm_get_unique_id(rten_hip_ctx *ctx, uint8_t id[RTEN_HIP_COMM_ID_BYTES]) {
    RTEN_CHECK_CTX(ctx);
    if (!id) return RTEN_HIP_ERR_INVALID_VALUE;
    Rccl *r = rccl();
    if (!r->load_error.empty()) return rten_set_error(ctx, RTEN_HIP_ERR_NO_DEVICE, "%s", r->load_error.c_str());
    UniqueId u;
    const int rc = r->get_unique_id(&u);
    if (rc != 0) return rccl_fail(ctx, r, rc, "ncclGetUniqueId");
    memcpy(id, u.internal, RTEN_HIP_COMM_ID_BYTES);
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_comm_init_rank(rten_hip_ctx *ctx, const uint8_t id[RTEN_HIP_COMM_ID_BYTES], int32_t world_size, int32_t rank,
                                            rten_hip_comm **out_comm) {
    RTEN_CHECK_CTX(ctx);
    if (!out_comm) return RTEN_HIP_ERR_INVALID_VALUE;
    *out_comm = nullptr;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size)
        return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "comm_init_rank: need 0 <= rank < world_size and an id from rten_hip_comm_get_unique_id");
    Rccl *r = rccl();
    if (!r->load_error.empty()) return rten_set_error(ctx, RTEN_HIP_ERR_NO_DEVICE, "%s", r->load_error.c_str());
    UniqueId u;
    memcpy(u.internal, id, RTEN_HIP_COMM_ID_BYTES);
    comm_t c = nullptr;
    const int rc = r->comm_init_rank(&c, world_size, u, rank); // one rank per GPU: the context's device is current (RTEN_CHECK_CTX)
    if (rc != 0) return rccl_fail(ctx, r, rc, "ncclCommInitRank");
    rten_hip_comm *cm = new rten_hip_comm();
    cm->comm = c; cm->world = world_size; cm->rank = rank; cm->device = ctx->device;
    *out_comm = cm;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_broadcast(rten_hip_ctx *ctx, rten_hip_comm *comm, void *buf, size_t bytes, int32_t root) {
    RTEN_CHECK_CTX(ctx);
    if (!comm || !comm->comm) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "broadcast: NULL communicator");
    if (comm->device != ctx->device) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "broadcast: communicator belongs to another device");
    if (root < 0 || root >= comm->world) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "broadcast: root out of range");
    if (ctx->capturing) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "broadcast: not capturable into a hipGraph (load-time collective)");
    if (bytes == 0) return RTEN_HIP_OK;
    if (!buf) return rten_set_error(ctx, RTEN_HIP_ERR_INVALID_VALUE, "broadcast: NULL buffer");
    Rccl *r = rccl();
    const int rc = r->broadcast(buf, buf, bytes, kNcclUint8, root, comm->comm, ctx->stream);
    if (rc != 0) return rccl_fail(ctx, r, rc, "ncclBroadcast");
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_comm_world_size(rten_hip_comm *comm, int32_t *world_size, int32_t *rank) {
    if (!comm) return RTEN_HIP_ERR_INVALID_VALUE;
    if (world_size) *world_size = comm->world;
    if (rank) *rank = comm->rank;
    return RTEN_HIP_OK;
}

RTEN_EXPORT int32_t rten_hip_comm_destroy(rten_hip_ctx *ctx, rten_hip_comm *comm) {
    RTEN_CHECK_CTX(ctx);
    if (!comm) return RTEN_HIP_OK;
    int32_t st = RTEN_HIP_OK;
    if (comm->comm) {
        hipStreamSynchronize(ctx->stream); // a broadcast still in flight on the stream uses the communicator
        Rccl *r = rccl();
        const int rc = r->comm_destroy(comm->comm);
        if (rc != 0) st = rccl_fail(ctx, r, rc, "ncclCommDestroy");
    }
    delete comm;
    return st;
}
