// Gradient exchange over RCCL behind the C-ABI: one communicator per process (one process per GPU),
// in-place all-reduce / broadcast on a caller-supplied HIP stream.  RCCL is resolved at run time
// with dlopen (the copy PyTorch already mapped is reused when there is one), so the kernel library
// has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);     // already mapped by the host framework?
            if (r.handle) break;
        }
        for (const char* n : names) {
            if (r.handle) break;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.handle) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.Broadcast = (decltype(r.Broadcast))dlsym(r.handle, "ncclBroadcast");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.Broadcast && r.CommDestroy;
    });
    return r;
}

inline int nccl_rc(ncclResult_t e) { return e == ncclSuccess ? 0 : VITK_ERR_COMM - (int)e; }

inline bool nccl_type(int dtype, ncclDataType_t* t) {
    if (dtype == VITK_F32) { *t = ncclFloat32; return true; }
    if (dtype == VITK_BF16) { *t = ncclBfloat16; return true; }
    return false;
}

}  // namespace

extern "C" int vitk_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out) return VITK_ERR_NULL;
    if (bytes < sizeof(ncclUniqueId)) return VITK_ERR_SHAPE;
    Rccl& r = rccl();
    if (!r.ok) return VITK_ERR_UNSUPPORTED;
    return nccl_rc(r.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id_out)));
}

extern "C" int vitk_comm_init(int rank, int world, const void* unique_id, void** comm_out) {
    if (!unique_id || !comm_out) return VITK_ERR_NULL;
    if (world < 1 || rank < 0 || rank >= world) return VITK_ERR_SHAPE;
    Rccl& r = rccl();
    if (!r.ok) return VITK_ERR_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = nccl_rc(r.CommInitRank(&c, world, id, rank));
    if (rc) return rc;
    *comm_out = c;
    return 0;
}

extern "C" int vitk_allreduce(void* comm, void* buf, size_t count, int dtype, int average, void* stream) {
    if (!comm || !buf) return VITK_ERR_NULL;
    ncclDataType_t t;
    if (!nccl_type(dtype, &t)) return VITK_ERR_DTYPE;
    Rccl& r = rccl();
    if (!r.ok) return VITK_ERR_UNSUPPORTED;
    return nccl_rc(r.AllReduce(buf, buf, count, t, average ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int vitk_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream) {
    if (!comm || !buf) return VITK_ERR_NULL;
    ncclDataType_t t;
    if (!nccl_type(dtype, &t)) return VITK_ERR_DTYPE;
    Rccl& r = rccl();
    if (!r.ok) return VITK_ERR_UNSUPPORTED;
    return nccl_rc(r.Broadcast(buf, buf, count, t, root, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int vitk_comm_destroy(void* comm) {
    if (!comm) return VITK_ERR_NULL;
    Rccl& r = rccl();
    if (!r.ok) return VITK_ERR_UNSUPPORTED;
    return nccl_rc(r.CommDestroy((ncclComm_t)comm));
}
