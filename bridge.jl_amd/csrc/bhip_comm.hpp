// bhip_comm.hpp -- the ONE collective of the path, inside the product: an RCCL all-gather over xGMI of the per-GPU
// statistics block (chains are sharded by contiguous global id, replicated guide, no data-path collective;
// acceptance / log-weight statistics -- and, optionally, pointwise Welford states, src/mclog.jl:31-38 -- are gathered).
//
// The reference has no communication at all (single-threaded Julia); this is the multi-GPU boundary a `ccall` caller
// gets: bhip_comm_init_rank (one process per GPU, id handshake by the launcher), bhip_comm_init_all (one process,
// every GPU of the node), bhip_comm_allgather[_stats], bhip_comm_destroy.
//
// RCCL is loaded lazily with dlopen("librccl.so.1"): libbridgehip.so itself has no link-time dependency on it (a process
// that already carries an RCCL -- PyTorch-ROCm bundles one under the same soname -- keeps using that one), and every
// bhip_comm_* call fails loudly with BHIP_EHIP when the library cannot be found.
#pragma once
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <rccl/rccl.h>

namespace bhip {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;              // what bhip_comm_query reports: RCCL's own view of the communicator
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    std::string err;
};

inline RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.err = std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char *s) {
            void *p = dlsym(api.handle, s);
            if (!p && api.err.empty()) api.err = std::string("RCCL symbol missing: ") + s;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
    });
    return api;
}

}  // namespace bhip
