// comm.cpp -- thin C-ABI wrappers over RCCL for the two collectives of the sharded RIME step
// (rime_comm_init / allgather_vis / reduce_grads).  They replace the per-device Python
// loop of DistributedLogProb.closure (optim.py:1539-1566: parameters copied to every device, gradients summed
// on device 0).  The product's Python path uses torch.distributed (backend "nccl" == RCCL), which drives the
// same RCCL calls; these entry points give a non-torch host (or a torch host that wants its own communicator)
// the same operations behind the library's ABI: raw device pointers, a stream, no allocation.
//
// RCCL is resolved at first use with dlopen / dlsym -- the copy already loaded into the process (torch ships
// its own librccl.so) wins, so that two RCCL instances never coexist; the library itself has no link-time
// dependency on RCCL and loads on machines without it (every entry point then returns RIME_EUNSUPPORTED).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include "rime_common.h"

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);             // already in the process (torch's copy)?
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce;
    });
    return r;
}

int fail(ncclResult_t e)
{
    const Rccl& r = rccl();
    std::snprintf(rime::g_last_error, sizeof(rime::g_last_error), "RCCL: %s",
                  r.GetErrorString ? r.GetErrorString(e) : "error");
    return RIME_ELAUNCH;
}

} // namespace

extern "C" int rime_comm_unique_id(void* id128)
{
    if (!id128) return RIME_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) return RIME_EUNSUPPORTED;
    ncclUniqueId id;
    const ncclResult_t e = r.GetUniqueId(&id);
    if (e != ncclSuccess) return fail(e);
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return RIME_OK;
}

extern "C" int rime_comm_init(void** comm_out, int nranks, int rank, const void* id128)
{
    if (!comm_out || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return RIME_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) return RIME_EUNSUPPORTED;
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t e = r.CommInitRank(&c, nranks, id, rank);
    if (e != ncclSuccess) return fail(e);
    *comm_out = c;
    return RIME_OK;
}

extern "C" int rime_comm_destroy(void* comm)
{
    if (!comm) return RIME_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) return RIME_EUNSUPPORTED;
    const ncclResult_t e = r.CommDestroy(static_cast<ncclComm_t>(comm));
    return e == ncclSuccess ? RIME_OK : fail(e);
}

extern "C" int rime_comm_allgather_vis(void* comm, int dtype, const void* vis_local, void* vis_all,
                                       size_t complex_per_rank, void* stream)
{
    if (!comm || !vis_local || !vis_all || (dtype != RIME_F32 && dtype != RIME_F64)) return RIME_EINVAL;
    if (complex_per_rank == 0) return RIME_OK;
    Rccl& r = rccl();
    if (!r.ok) return RIME_EUNSUPPORTED;
    const ncclResult_t e = r.AllGather(vis_local, vis_all, 2 * complex_per_rank, dtype == RIME_F32 ? ncclFloat32 : ncclFloat64,
                                       static_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream));
    return e == ncclSuccess ? RIME_OK : fail(e);
}

extern "C" int rime_comm_reduce_grads(void* comm, int dtype, void* grads, size_t count, void* stream)
{
    if (!comm || !grads || (dtype != RIME_F32 && dtype != RIME_F64)) return RIME_EINVAL;
    if (count == 0) return RIME_OK;
    Rccl& r = rccl();
    if (!r.ok) return RIME_EUNSUPPORTED;
    const ncclResult_t e = r.AllReduce(grads, grads, count, dtype == RIME_F32 ? ncclFloat32 : ncclFloat64, ncclSum,
                                       static_cast<ncclComm_t>(comm), reinterpret_cast<hipStream_t>(stream));
    return e == ncclSuccess ? RIME_OK : fail(e);
}
