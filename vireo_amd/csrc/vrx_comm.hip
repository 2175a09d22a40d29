// Restart shard: the only inter-GPU exchange on the path.
// vireo_wrap (vireoSNP/utils/vireo_wrap.py:74-91) fits n_init independent restarts and
// keeps argmax(ELBO_[-1]).  With one process per GPU each rank fits its share and the
// per-restart ELBOs are all-gathered over RCCL (xGMI); the winner's state is broadcast.
// RCCL is bound lazily with dlopen so that single-GPU use never loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "vrx_common.h"

namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                              hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_api;
std::once_flag g_once;
bool g_ok = false;

bool load_rccl() {
    std::call_once(g_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_api.lib) break;
        }
        if (!g_api.lib) return;
#define VRX_SYM(field, name)                                                   \
    g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(g_api.lib, name)); \
    if (!g_api.field) return;
        VRX_SYM(GetUniqueId, "ncclGetUniqueId")
        VRX_SYM(CommInitRank, "ncclCommInitRank")
        VRX_SYM(CommDestroy, "ncclCommDestroy")
        VRX_SYM(AllGather, "ncclAllGather")
        VRX_SYM(AllReduce, "ncclAllReduce")
        VRX_SYM(Broadcast, "ncclBroadcast")
        VRX_SYM(GetErrorString, "ncclGetErrorString")
#undef VRX_SYM
        g_ok = true;
    });
    if (!g_ok) vrx_set_error("RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    return g_ok;
}
}  // namespace

#define VRX_NCCL(expr)                                                                        \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess) {                                                             \
            vrx_set_error("%s failed: %s", #expr, g_api.GetErrorString(r__));                 \
            return VRX_ERR_COMM;                                                              \
        }                                                                                     \
    } while (0)

struct vrx_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<double> send, recv;
};

static_assert(sizeof(ncclUniqueId) == VRX_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" int vrx_comm_unique_id(uint8_t* id) {
    VRX_REQUIRE(id, "vrx_comm_unique_id: null output");
    if (!load_rccl()) return VRX_ERR_COMM;
    ncclUniqueId u;
    VRX_NCCL(g_api.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return VRX_OK;
}

extern "C" int vrx_comm_create(int device, int rank, int world, const uint8_t* id, vrx_comm** out) {
    VRX_REQUIRE(out && id, "vrx_comm_create: null argument");
    *out = nullptr;
    VRX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "vrx_comm_create: bad rank/world");
    if (!load_rccl()) return VRX_ERR_COMM;
    VRX_HIP(hipSetDevice(device));
    vrx_comm* c = new vrx_comm();
    c->device = device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t r = g_api.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        vrx_set_error("ncclCommInitRank failed: %s", g_api.GetErrorString(r));
        delete c;
        return VRX_ERR_COMM;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        vrx_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        g_api.CommDestroy(c->comm);
        delete c;
        return VRX_ERR_HIP;
    }
    *out = c;
    return VRX_OK;
}

extern "C" void vrx_comm_destroy(vrx_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    if (c->comm) g_api.CommDestroy(c->comm);
    delete c;
}

extern "C" int vrx_comm_allgather_f64(vrx_comm* c, const double* local, int64_t n_local,
                                      double* out) {
    VRX_REQUIRE(c && local && out && n_local >= 1, "vrx_comm_allgather_f64: bad argument");
    VRX_HIP(hipSetDevice(c->device));
    if (c->send.n < (size_t)n_local) VRX_HIP(c->send.alloc((size_t)n_local));
    if (c->recv.n < (size_t)(n_local * c->world)) VRX_HIP(c->recv.alloc((size_t)(n_local * c->world)));
    VRX_HIP(hipMemcpyAsync(c->send.p, local, (size_t)n_local * sizeof(double), hipMemcpyHostToDevice,
                           c->stream));
    VRX_NCCL(g_api.AllGather(c->send.p, c->recv.p, (size_t)n_local, ncclDouble, c->comm, c->stream));
    VRX_HIP(hipMemcpyAsync(out, c->recv.p, (size_t)(n_local * c->world) * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
    VRX_HIP(hipStreamSynchronize(c->stream));
    return VRX_OK;
}

extern "C" int vrx_comm_barrier(vrx_comm* c) {
    VRX_REQUIRE(c, "vrx_comm_barrier: null comm");
    VRX_HIP(hipSetDevice(c->device));
    if (c->send.n < 1) VRX_HIP(c->send.alloc(1));
    VRX_HIP(hipMemsetAsync(c->send.p, 0, sizeof(double), c->stream));
    VRX_NCCL(g_api.AllReduce(c->send.p, c->send.p, 1, ncclDouble, ncclSum, c->comm, c->stream));
    VRX_HIP(hipStreamSynchronize(c->stream));
    return VRX_OK;
}

extern "C" int vrx_comm_bcast_f64(vrx_comm* c, double* buf, int64_t n, int root) {
    VRX_REQUIRE(c && buf && n >= 1 && root >= 0 && root < c->world, "vrx_comm_bcast_f64: bad argument");
    VRX_HIP(hipSetDevice(c->device));
    if (c->recv.n < (size_t)n) VRX_HIP(c->recv.alloc((size_t)n));
    if (c->rank == root)
        VRX_HIP(hipMemcpyAsync(c->recv.p, buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice,
                               c->stream));
    VRX_NCCL(g_api.Broadcast(c->recv.p, c->recv.p, (size_t)n, ncclDouble, root, c->comm, c->stream));
    if (c->rank != root)
        VRX_HIP(hipMemcpyAsync(buf, c->recv.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost,
                               c->stream));
    VRX_HIP(hipStreamSynchronize(c->stream));
    return VRX_OK;
}
