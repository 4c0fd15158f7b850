// Data-parallel collectives behind the C-ABI: RCCL over xGMI, one process per GPU (include/ssdseg.h, "data parallelism").
//
// The reference has no distributed code (single-process Keras `fit`, NB03#cell16); this is the one new boundary of the build
// (SURVEY.md 8(e)).  One communicator per context, created with ncclCommInitRank from a 128-byte unique id that rank 0
// generates and the host side hands to the other ranks (file rendezvous in ssdseglib/_parallel.py -- no torch anywhere).
// librccl.so is opened lazily (dlopen) at the first ssdseg_comm_* call: single-GPU users never load it, and the library
// still loads on machines without RCCL.  Every collective is enqueued on the context's main stream AFTER joining the
// weight-gradient side stream, so it is ordered behind the backward pass and in front of the optimizer with no host sync.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.handle != nullptr) return 0;
    const char* names[] = {getenv("SSDSEG_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (n == nullptr || n[0] == 0) continue;
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h != nullptr) break;
    }
    if (h == nullptr) {
        ssdseg_set_error("librccl.so could not be loaded (%s): multi-GPU data parallelism needs RCCL", dlerror());
        return SSDSEG_EINVAL(0);
    }
    RcclApi a;
    a.handle = h;
#define SSDSEG_SYM(field, name)                                                         \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                       \
    if (a.field == nullptr) {                                                            \
        ssdseg_set_error("librccl.so does not export %s", name);                         \
        dlclose(h);                                                                      \
        return SSDSEG_EINVAL(0);                                                         \
    }
    SSDSEG_SYM(GetUniqueId, "ncclGetUniqueId")
    SSDSEG_SYM(CommInitRank, "ncclCommInitRank")
    SSDSEG_SYM(CommDestroy, "ncclCommDestroy")
    SSDSEG_SYM(AllReduce, "ncclAllReduce")
    SSDSEG_SYM(Broadcast, "ncclBroadcast")
    SSDSEG_SYM(GroupStart, "ncclGroupStart")
    SSDSEG_SYM(GroupEnd, "ncclGroupEnd")
    SSDSEG_SYM(GetErrorString, "ncclGetErrorString")
#undef SSDSEG_SYM
    g_rccl = a;
    return 0;
}

int rccl_fail(ncclResult_t r, const char* what) {
    ssdseg_set_error("%s: %s (%d)", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error", (int)r);
    return -2000 - (int)r;
}

#define SSDSEG_RCCL(call)                                   \
    do {                                                    \
        ncclResult_t _r = (call);                           \
        if (_r != ncclSuccess) return rccl_fail(_r, #call); \
    } while (0)

__global__ void __launch_bounds__(256) scale_inplace_kernel(float* __restrict__ x, size_t count, float a) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) x[i] *= a;
}

}  // namespace

extern "C" {

int ssdseg_comm_unique_id(void* id_host, size_t id_bytes) {
    SSDSEG_ARG(id_host != nullptr, 1);
    SSDSEG_ARG(id_bytes == SSDSEG_COMM_ID_BYTES, 2);
    static_assert(sizeof(ncclUniqueId) == SSDSEG_COMM_ID_BYTES, "ncclUniqueId size");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    SSDSEG_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(id_host, &id, sizeof(id));
    return 0;
}

int ssdseg_comm_init_rank(ssdseg_ctx* ctx, const void* id_host, size_t id_bytes, int rank, int world) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(id_host != nullptr, 2);
    SSDSEG_ARG(id_bytes == SSDSEG_COMM_ID_BYTES, 3);
    SSDSEG_ARG(world >= 1, 5);
    SSDSEG_ARG(rank >= 0 && rank < world, 4);
    SSDSEG_ARG(ctx->comm == nullptr, 1);
    int rc = rccl_load();
    if (rc) return rc;
    SSDSEG_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t comm = nullptr;
    SSDSEG_RCCL(g_rccl.CommInitRank(&comm, world, id, rank));   // collective: returns once every rank has joined
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return 0;
}

int ssdseg_comm_destroy(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ctx->comm == nullptr) return 0;
    (void)ssdseg_join(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    SSDSEG_RCCL(g_rccl.CommDestroy(comm));
    return 0;
}

int ssdseg_comm_info(ssdseg_ctx* ctx, int* rank_host, int* world_host) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (rank_host) *rank_host = ctx->comm ? ctx->comm_rank : 0;
    if (world_host) *world_host = ctx->comm ? ctx->comm_world : 1;
    return 0;
}

// The step's one collective: sum of the flat gradient bucket (Adam applies 1/world through its grad_scale) and, in the same
// RCCL group, the MEAN of the non-trainable state bucket (BatchNormalization moving mean / variance: every replica updated them
// from its own shard's batch statistics; averaging keeps the replicas -- and a rank-0 checkpoint -- identical).
int ssdseg_allreduce_grads(ssdseg_ctx* ctx, float* grads, size_t count, float* state, size_t state_count) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(ctx->comm != nullptr, 1);
    SSDSEG_ARG(grads != nullptr || count == 0, 2);
    SSDSEG_ARG(state != nullptr || state_count == 0, 4);
    int rc = ssdseg_join(ctx);   // weight gradients are produced on the side stream
    if (rc) return rc;
    // (no short cut for a one-rank communicator: that is the configuration in which a one-GPU box exercises the RCCL calls)
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    const double wire = 2.0 * (ctx->comm_world - 1) / ctx->comm_world * 4.0 * ((double)count + (double)state_count);   // ring bytes per rank
    if (ctx->timing) ssdseg_timing_begin(ctx, "rccl_allreduce_grads", wire, 0.0);
    // first error wins, but the group is ALWAYS closed and the timing bracket ended: a return from inside the pair would leave
    // this thread's RCCL group open and the next ssdseg_comm_* call silently queued into it
    ncclResult_t first = g_rccl.GroupStart();
    const char* where = "ncclGroupStart";
    const bool opened = first == ncclSuccess;
    if (first == ncclSuccess && count) {
        first = g_rccl.AllReduce(grads, grads, count, ncclFloat32, ncclSum, comm, ctx->stream);
        where = "ncclAllReduce(grads)";
    }
    if (first == ncclSuccess && state_count) {
        first = g_rccl.AllReduce(state, state, state_count, ncclFloat32, ncclSum, comm, ctx->stream);
        where = "ncclAllReduce(state)";
    }
    if (opened) {
        const ncclResult_t end = g_rccl.GroupEnd();
        if (first == ncclSuccess && end != ncclSuccess) {
            first = end;
            where = "ncclGroupEnd";
        }
    }
    if (ctx->timing) ssdseg_timing_end(ctx);
    if (first != ncclSuccess) return rccl_fail(first, where);
    if (state_count) {
        const int blocks = (int)((state_count + 255) / 256 < 1024 ? (state_count + 255) / 256 : 1024);
        SSDSEG_LAUNCH(ctx, 8.0 * state_count, 0.0, scale_inplace_kernel, dim3(blocks), dim3(256), 0, state, state_count, 1.0f / (float)ctx->comm_world);
        SSDSEG_LAUNCH_CHECK();
    }
    return 0;
}

// generic in-place all-reduce of a small device buffer (bench.py: max-over-ranks of the timed region, barrier)
int ssdseg_allreduce(ssdseg_ctx* ctx, void* buf, size_t count, int dtype, int op) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(ctx->comm != nullptr, 1);
    SSDSEG_ARG(buf != nullptr, 2);
    SSDSEG_ARG(dtype == SSDSEG_COMM_F32 || dtype == SSDSEG_COMM_F64, 4);
    SSDSEG_ARG(op == SSDSEG_COMM_SUM || op == SSDSEG_COMM_MAX, 5);
    int rc = ssdseg_join(ctx);
    if (rc) return rc;
    if (count == 0) return 0;
    SSDSEG_RCCL(g_rccl.AllReduce(buf, buf, count, dtype == SSDSEG_COMM_F32 ? ncclFloat32 : ncclFloat64, op == SSDSEG_COMM_SUM ? ncclSum : ncclMax,
                                 (ncclComm_t)ctx->comm, ctx->stream));
    return 0;
}

// replicate rank `root`'s buffer (initial weights / optimizer state when the ranks were not seeded identically)
int ssdseg_broadcast(ssdseg_ctx* ctx, float* buf, size_t count, int root) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(ctx->comm != nullptr, 1);
    SSDSEG_ARG(buf != nullptr || count == 0, 2);
    SSDSEG_ARG(root >= 0 && root < ctx->comm_world, 4);
    int rc = ssdseg_join(ctx);
    if (rc) return rc;
    if (count == 0) return 0;
    SSDSEG_RCCL(g_rccl.Broadcast(buf, buf, count, ncclFloat32, root, (ncclComm_t)ctx->comm, ctx->stream));
    return 0;
}

}  // extern "C"
