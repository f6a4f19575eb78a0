// Pose-record all-gather over RCCL (xGMI) for the C ABI: the only collective of the multi-GPU path.
// RCCL is resolved at run time with dlopen, so libdvo_hip.so has no link-time dependency on it and shares the RCCL
// instance a host framework (e.g. torch.distributed) may already have loaded.  Host-side only.
#include "../../include/dvo_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

typedef struct { char internal[128]; } rcclUniqueId;     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rcclComm_t;
typedef int (*fn_get_id)(rcclUniqueId*);
typedef int (*fn_init_rank)(rcclComm_t*, int, rcclUniqueId, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int /*ncclDataType_t*/, rcclComm_t, hipStream_t);
typedef int (*fn_destroy)(rcclComm_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* handle = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_allgather allgather = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    char err[256] = "";
};

Rccl g_rccl;
thread_local char g_comm_error[512] = "";

bool load_rccl()
{
    if (g_rccl.handle) return true;
    const char* env = getenv("DVO_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_rccl.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
        g_rccl.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
        g_rccl.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
        g_rccl.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
        g_rccl.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
        if (g_rccl.get_id && g_rccl.init_rank && g_rccl.allgather && g_rccl.destroy) { g_rccl.handle = h; return true; }
        dlclose(h);
    }
    snprintf(g_comm_error, sizeof(g_comm_error), "RCCL not found (tried DVO_RCCL_LIB, librccl.so.1, librccl.so, /opt/rocm/lib)");
    return false;
}

int comm_fail(int code, const char* what, int rc)
{
    snprintf(g_comm_error, sizeof(g_comm_error), "%s failed: %s", what, (g_rccl.errstr && rc) ? g_rccl.errstr(rc) : "error");
    return code;
}

}  // namespace

struct dvo_comm {
    rcclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;
    float* d_send = nullptr;
    float* d_recv = nullptr;
    size_t cap_records = 0;
};

extern "C" {

const char* dvo_comm_last_error(void) { return g_comm_error; }

int dvo_comm_unique_id(void* id128)
{
    if (!id128) return DVO_ERR_INVALID_ARG;
    if (!load_rccl()) return DVO_ERR_BAD_STATE;
    rcclUniqueId id;
    int rc = g_rccl.get_id(&id);
    if (rc) return comm_fail(DVO_ERR_HIP, "ncclGetUniqueId", rc);
    memcpy(id128, id.internal, 128);
    return DVO_OK;
}

int dvo_comm_create(int device, int rank, int nranks, const void* id128, dvo_comm** out)
{
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return DVO_ERR_INVALID_ARG;
    *out = nullptr;
    if (!load_rccl()) return DVO_ERR_BAD_STATE;
    if (hipSetDevice(device) != hipSuccess) { snprintf(g_comm_error, sizeof(g_comm_error), "hipSetDevice(%d) failed", device); return DVO_ERR_NO_DEVICE; }
    dvo_comm* c = new (std::nothrow) dvo_comm();
    if (!c) return DVO_ERR_OUT_OF_MEMORY;
    c->rank = rank; c->nranks = nranks; c->device = device;
    rcclUniqueId id;
    memcpy(id.internal, id128, 128);
    int rc = g_rccl.init_rank(&c->comm, nranks, id, rank);
    if (rc) { delete c; return comm_fail(DVO_ERR_HIP, "ncclCommInitRank", rc); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_rccl.destroy(c->comm); delete c; return DVO_ERR_HIP; }
    *out = c;
    return DVO_OK;
}

int dvo_allgather_poses(dvo_comm* c, const float* mine, int n_records, float* all)
{
    if (!c || !mine || !all || n_records < 1) return DVO_ERR_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return DVO_ERR_HIP;
    if ((size_t)n_records > c->cap_records) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr;
        c->cap_records = 0;
        if (hipMalloc((void**)&c->d_send, (size_t)n_records * 16 * sizeof(float)) != hipSuccess ||
            hipMalloc((void**)&c->d_recv, (size_t)n_records * 16 * sizeof(float) * c->nranks) != hipSuccess)
            return DVO_ERR_OUT_OF_MEMORY;
        c->cap_records = (size_t)n_records;
    }
    const size_t count = (size_t)n_records * 16;
    if (hipMemcpyAsync(c->d_send, mine, count * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) return DVO_ERR_HIP;
    int rc = g_rccl.allgather(c->d_send, c->d_recv, count, 7 /* ncclFloat32 */, c->comm, c->stream);
    if (rc) return comm_fail(DVO_ERR_HIP, "ncclAllGather", rc);
    if (hipMemcpyAsync(all, c->d_recv, count * sizeof(float) * c->nranks, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return DVO_ERR_HIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return DVO_ERR_HIP;
    return DVO_OK;
}

void dvo_comm_destroy(dvo_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.destroy) g_rccl.destroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C"
