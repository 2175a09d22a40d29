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
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // optional (absent from very old librccl builds): the detail text behind an error code, the version
    const char* (*GetLastError)(ncclComm_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
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
        VRX_SYM(GroupStart, "ncclGroupStart")
        VRX_SYM(GroupEnd, "ncclGroupEnd")
        VRX_SYM(GetErrorString, "ncclGetErrorString")
#undef VRX_SYM
        g_api.GetLastError = reinterpret_cast<decltype(g_api.GetLastError)>(dlsym(g_api.lib, "ncclGetLastError"));
        g_api.GetVersion = reinterpret_cast<decltype(g_api.GetVersion)>(dlsym(g_api.lib, "ncclGetVersion"));
        g_ok = true;
    });
    if (!g_ok) vrx_set_error("RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    return g_ok;
}
// what librccl recorded behind the last error code ("" when it has nothing / cannot say)
const char* rccl_detail() {
    const char* d = g_api.GetLastError ? g_api.GetLastError(nullptr) : nullptr;
    return d ? d : "";
}
}  // namespace

#define VRX_NCCL(expr)                                                                        \
    do {                                                                                      \
        ncclResult_t r__ = (expr);                                                            \
        if (r__ != ncclSuccess) {                                                             \
            vrx_set_error("%s failed: %s%s%s", #expr, g_api.GetErrorString(r__),              \
                          rccl_detail()[0] ? " -- " : "", rccl_detail());                     \
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
        vrx_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s%s%s", rank, world, device,
                      g_api.GetErrorString(r), rccl_detail()[0] ? " -- " : "", rccl_detail());
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

// rank, world, device of the communicator and librccl's version code (ncclGetVersion: 10000 major +
// 100 minor + patch from 2.9 on; 0 when the library does not say): bench.py's `comm` block
extern "C" int vrx_comm_info(vrx_comm* c, int32_t* info4) {
    VRX_REQUIRE(c && info4, "vrx_comm_info: null argument");
    int v = 0;
    if (g_api.GetVersion && g_api.GetVersion(&v) != ncclSuccess) v = 0;
    info4[0] = c->rank;
    info4[1] = c->world;
    info4[2] = c->device;
    info4[3] = v;
    return VRX_OK;
}

// The winner's state, device to device (vireo_wrap.py:90-94 hands `_models_all[_idx]` on; with one
// process per GPU the other ranks need it too): ncclBroadcast straight from / into the model's HBM
// buffers -- ID_prob, GT_prob (Vireo), beta_mu, beta_sum in ONE group call -- with no host staging.
// Every rank passes a model of the same problem shape and configuration.
extern "C" int vrx_comm_bcast_model(vrx_comm* c, vrx_model* m, int root) {
    VRX_REQUIRE(c && m && root >= 0 && root < c->world, "vrx_comm_bcast_model: bad argument");
    VrxModelBuffers b;
    int rc = vrx_model_state_buffers(m, c->rank != root, &b);  // (drains the model's stream)
    if (rc) return rc;
    VRX_REQUIRE(b.device == c->device, "vrx_comm_bcast_model: the model lives on device %d, the communicator on %d",
                b.device, c->device);
    VRX_HIP(hipSetDevice(c->device));
    VRX_NCCL(g_api.GroupStart());
    for (int i = 0; i < 4; ++i)
        if (b.n[i] > 0) {
            const ncclResult_t r = g_api.Broadcast(b.p[i], b.p[i], b.n[i], ncclDouble, root, c->comm, c->stream);
            if (r != ncclSuccess) {
                (void)g_api.GroupEnd();
                vrx_set_error("ncclBroadcast (model state %d) failed: %s", i, g_api.GetErrorString(r));
                return VRX_ERR_COMM;
            }
        }
    VRX_NCCL(g_api.GroupEnd());
    VRX_HIP(hipStreamSynchronize(c->stream));
    return VRX_OK;
}
