// C1: slab reassembly with ONE RCCL all-gather over xGMI (the only collective on this path).
// RCCL is loaded lazily with dlopen so that single-GPU use never maps the (large) library and the
// C-ABI loads on machines without it.  One process per GPU; the launcher distributes the unique id.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "pb3d_internal.h"

namespace {

struct RcclApi {
    void* lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
    const char* (*GetErrorString)(ncclResult_t);
};

RcclApi g_rccl = {};

int load_rccl() {
    if (g_rccl.lib) return PB3D_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) {
        pb3d_set_error("RCCL not found: %s", dlerror());
        return PB3D_ECOMM;
    }
    RcclApi a = {};
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))dlsym(lib, "ncclCommUserRank");
    if (!a.CommCount || !a.CommUserRank || !a.GetUniqueId || !a.CommInitRank || !a.AllGather || !a.AllReduce || !a.CommDestroy || !a.GetErrorString) {
        pb3d_set_error("RCCL library lacks a required symbol");
        dlclose(lib);
        return PB3D_ECOMM;
    }
    g_rccl = a;
    return PB3D_OK;
}

#define PB3D_NCCL(call)                                                                 \
    do {                                                                                \
        ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess) {                                                        \
            pb3d_set_error("%s failed: %s", #call, g_rccl.GetErrorString(r_));          \
            return PB3D_ECOMM;                                                          \
        }                                                                               \
    } while (0)

}  // namespace

extern "C" {

int pb3d_comm_unique_id(uint8_t id[128]) {
    PB3D_REQUIRE(id != nullptr, "pb3d_comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    PB3D_TRY(load_rccl());
    ncclUniqueId u;
    PB3D_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, 128);
    return PB3D_OK;
}

int pb3d_comm_init(pb3d_ctx* ctx, const uint8_t id[128], int rank, int nranks) {
    PB3D_REQUIRE(ctx != nullptr && id != nullptr, "pb3d_comm_init: null argument");
    PB3D_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "pb3d_comm_init: bad rank %d of %d", rank, nranks);
    PB3D_REQUIRE(ctx->rccl_comm == nullptr, "pb3d_comm_init: communicator already initialised");
    PB3D_TRY(load_rccl());
    PB3D_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclComm_t comm;
    PB3D_NCCL(g_rccl.CommInitRank(&comm, nranks, u, rank));
    ctx->rccl_comm = (void*)comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return PB3D_OK;
}

int pb3d_comm_info(pb3d_ctx* ctx, int* rank, int* nranks) {
    PB3D_REQUIRE(ctx != nullptr && rank && nranks, "pb3d_comm_info: null argument");
    PB3D_REQUIRE(ctx->rccl_comm != nullptr, "pb3d_comm_info: call pb3d_comm_init first");
    // asked of RCCL itself, not echoed from pb3d_comm_init's arguments
    PB3D_NCCL(g_rccl.CommCount((ncclComm_t)ctx->rccl_comm, nranks));
    PB3D_NCCL(g_rccl.CommUserRank((ncclComm_t)ctx->rccl_comm, rank));
    return PB3D_OK;
}

int pb3d_allgather_dev(pb3d_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_allgather: null context");
    PB3D_REQUIRE(ctx->rccl_comm != nullptr, "pb3d_allgather: call pb3d_comm_init first");
    if (bytes_per_rank == 0) return PB3D_OK;
    PB3D_REQUIRE(d_send && d_recv, "pb3d_allgather: null buffer");
    PB3D_NCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, (ncclComm_t)ctx->rccl_comm, ctx->stream));
    return PB3D_OK;
}

int pb3d_allreduce_max_u64_dev(pb3d_ctx* ctx, void* d_buf, size_t count) {
    PB3D_REQUIRE(ctx != nullptr, "pb3d_allreduce_max_u64: null context");
    PB3D_REQUIRE(ctx->rccl_comm != nullptr, "pb3d_allreduce_max_u64: call pb3d_comm_init first");
    if (count == 0) return PB3D_OK;
    PB3D_REQUIRE(d_buf != nullptr, "pb3d_allreduce_max_u64: null buffer");
    PB3D_NCCL(g_rccl.AllReduce(d_buf, d_buf, count, ncclUint64, ncclMax, (ncclComm_t)ctx->rccl_comm, ctx->stream));
    return PB3D_OK;
}

int pb3d_comm_destroy(pb3d_ctx* ctx) {
    if (!ctx || !ctx->rccl_comm) return PB3D_OK;
    (void)hipStreamSynchronize(ctx->stream);
    g_rccl.CommDestroy((ncclComm_t)ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
    return PB3D_OK;
}

}  // extern "C"
