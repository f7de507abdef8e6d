// trsim_comm.hip — the one exchange of the multi-GPU path, and stream ordering with a caller's own streams.
//
// north_star: "independent env shards fan out across the 8 GPUs of one node with a single RCCL all-gather over xGMI of
// episode returns".  Shards never exchange state; this file holds that single collective behind the C ABI so
// that a host in any language can do it without torch: trs_comm_get_unique_id / trs_comm_init / trs_allgather_returns.
// RCCL is bound at run time (dlopen) — libtrsim.so has no link-time dependency on it, a process that never calls
// trs_comm_init never loads it, and a process that already holds an RCCL (PyTorch bundles one) shares that copy.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/trsim.h"
#include "trsim_env.hpp"
#include "trsim_internal.hpp"

#define TRS_EXPORT extern "C" __attribute__((visibility("default")))

#define CCHK(call)                                                                                \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) return trs_internal_fail(TRS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace trsim {

// the five entry points of rccl.h this file uses (signatures of /opt/rocm/include/rccl/rccl.h:187,220,260,339,678)
struct RcclId { char internal[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, RcclId id, int rank) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

struct Comm {
    int rank = 0, world = 1;
    void* nccl = nullptr;                // ncclComm_t; nullptr at world size 1 (the gather is a copy)
    float* gathered = nullptr;           // device [world * n_envs]
};

}  // namespace trsim

namespace {

using namespace trsim;

RcclApi g_rccl;
std::string g_rccl_err;

// one process-wide binding; a copy that is already mapped (PyTorch's) is preferred over loading a second one
bool load_rccl()
{
    if (g_rccl.lib) return true;
    const char* env = std::getenv("TRS_RCCL_LIB");                   // an explicit path wins (deployment knob, the only one this library reads)
    // a copy some other module of the process has already mapped — PyTorch bundles its own librccl under torch/lib, which a lookup
    // by soname would miss — is found by its PATH in /proc/self/maps and shared, so that one process never runs two RCCLs
    std::string mapped;
    if (!env) {
        if (FILE* f = std::fopen("/proc/self/maps", "r")) {
            char line[4096];
            while (std::fgets(line, sizeof line, f)) {
                const char* path = std::strchr(line, '/');
                if (!path || !std::strstr(path, "librccl.so")) continue;
                mapped.assign(path);
                while (!mapped.empty() && (mapped.back() == '\n' || mapped.back() == ' ')) mapped.pop_back();
                break;
            }
            std::fclose(f);
        }
    }
    const char* names[] = {env, mapped.empty() ? nullptr : mapped.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { if (n && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break; }
    if (!h) for (const char* n : names) { if (n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
    if (!h) { const char* why = dlerror(); g_rccl_err = std::string("librccl not found (set TRS_RCCL_LIB): ") + (why ? why : ""); return false; }
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) { g_rccl_err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather"; return false; }
    g_rccl = a;
    return true;
}

int rccl_fail(const char* what, int rc)
{
    return trs_internal_fail(TRS_ERR_DEVICE, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}

constexpr int kNcclFloat32 = 7;          // rccl.h:466

}  // namespace

namespace trsim {

void comm_destroy(trs_env* e)
{
    Comm* c = e->comm;
    if (!c) return;
    if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
    (void)hipFree(c->gathered);
    delete c;
    e->comm = nullptr;
}

}  // namespace trsim

TRS_EXPORT int trs_comm_get_unique_id(void* id_out)
{
    if (!id_out) return trs_internal_fail(TRS_ERR_ARG, "null argument");
    if (!load_rccl()) return trs_internal_fail(TRS_ERR_DEVICE, g_rccl_err);
    RcclId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id_out, &id, sizeof id);
    return TRS_OK;
}

TRS_EXPORT int trs_comm_init(trs_env* e, int rank, int world, const void* unique_id)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (world < 1 || rank < 0 || rank >= world) return trs_internal_fail(TRS_ERR_ARG, "bad rank / world size");
    if (world > 1 && !unique_id) return trs_internal_fail(TRS_ERR_ARG, "a communicator of more than one rank needs rank 0's unique id (trs_comm_get_unique_id)");
    CCHK(hipSetDevice(e->device));
    { int rq = sync_handle(e); if (rq) return rq; }
    comm_destroy(e);
    Comm* c = new (std::nothrow) Comm();
    if (!c) return trs_internal_fail(TRS_ERR_NOMEM, "out of memory");
    e->comm = c;
    c->rank = rank; c->world = world;
    if (hipMalloc((void**)&c->gathered, (size_t)world * e->n * sizeof(float)) != hipSuccess) { comm_destroy(e); return trs_internal_fail(TRS_ERR_NOMEM, "out of device memory"); }
    if (unique_id) {                                         // also at world size 1 when an id is given: the RCCL path itself is exercised
        if (!load_rccl()) { comm_destroy(e); return trs_internal_fail(TRS_ERR_DEVICE, g_rccl_err); }
        RcclId id;
        std::memcpy(&id, unique_id, sizeof id);
        const int rc = g_rccl.CommInitRank(&c->nccl, world, id, rank);
        if (rc) { c->nccl = nullptr; comm_destroy(e); return rccl_fail("ncclCommInitRank", rc); }
    }
    return TRS_OK;
}

TRS_EXPORT int trs_comm_destroy(trs_env* e)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    CCHK(hipSetDevice(e->device));
    { int rq = sync_handle(e); if (rq) return rq; }
    comm_destroy(e);
    return TRS_OK;
}

TRS_EXPORT int trs_allgather_returns(trs_env* e, const float** d_out_all, float* h_out_all)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    if (!e->comm) return trs_internal_fail(TRS_ERR_STATE, "no communicator: call trs_comm_init first");
    CCHK(hipSetDevice(e->device));
    { int rq = quiesce_handle(e); if (rq) return rq; }       // the collective is queued on the handle's stream, behind the steps
    Comm* c = e->comm;
    const size_t n = (size_t)e->n;
    if (c->nccl) {
        const int rc = g_rccl.AllGather(e->pp.ep_return, c->gathered, n, kNcclFloat32, c->nccl, e->sP);
        if (rc) return rccl_fail("ncclAllGather", rc);
    } else {
        CCHK(hipMemcpyAsync(c->gathered, e->pp.ep_return, n * sizeof(float), hipMemcpyDeviceToDevice, e->sP));
    }
    if (d_out_all) *d_out_all = c->gathered;
    if (h_out_all) {
        CCHK(hipMemcpyAsync(h_out_all, c->gathered, (size_t)c->world * n * sizeof(float), hipMemcpyDeviceToHost, e->sP));
        CCHK(hipStreamSynchronize(e->sP));
        return check_fault(e);
    }
    return TRS_OK;
}

// ---- ordering against a caller's own HIP streams ----------------------------------------------------------------------
// The handle works on its own non-blocking stream.  A producer of device-resident controls (a policy on torch's stream) and a
// consumer of device-resident frames must be ordered against it; these two calls do that without a host synchronisation
// (launch mode) or with the cheapest one that is correct (resident mode, where steps are posted by the host).
TRS_EXPORT int trs_stream_wait_external(trs_env* e, void* hip_stream)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    CCHK(hipSetDevice(e->device));
    hipStream_t ext = static_cast<hipStream_t>(hip_stream);
    if (resident_on(e) || resident_running(e)) { CCHK(hipStreamSynchronize(ext)); return TRS_OK; }   // a posted step reads its controls as soon as the worker sees the post
    CCHK(hipEventRecord(e->ev_order, ext));
    CCHK(hipStreamWaitEvent(e->sP, e->ev_order, 0));
    return TRS_OK;
}

TRS_EXPORT int trs_stream_signal_external(trs_env* e, void* hip_stream)
{
    if (!e) return trs_internal_fail(TRS_ERR_ARG, "null handle");
    CCHK(hipSetDevice(e->device));
    hipStream_t ext = static_cast<hipStream_t>(hip_stream);
    if (resident_running(e)) { int rw = resident_wait(e); if (rw) return rw; return check_fault(e); }   // completion flags: the frames are in memory; an event would wait for the worker to leave
    CCHK(hipEventRecord(e->ev_order, e->sP));
    CCHK(hipStreamWaitEvent(ext, e->ev_order, 0));
    return TRS_OK;
}
