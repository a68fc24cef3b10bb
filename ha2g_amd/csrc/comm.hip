// Data-parallel gradient exchange for hosts that bind the C-ABI directly (no torch.distributed): a thin layer over RCCL -- the collective
// library of ROCm, rings over xGMI inside a node -- with plain pointers in the signatures.  One process per GPU; rank 0 creates a 128-byte
// id (ha2g_comm_unique_id), hands it to the other ranks by whatever channel the host owns (file, socket, MPI), every rank calls
// ha2g_comm_init on its own device, and the flat gradient buffers of ha2g_amd/optim.py's layout go through ha2g_allreduce_bucket on the
// stream the backward ran on (SUM or mean, in place).  Replaces nn.DataParallel's gather / scatter (reference scripts/train.py:133-143).
//
// RCCL is resolved at the first call with dlopen("librccl.so.1"): libha2g_hip.so itself carries no link-time dependency on it (a
// single-GPU host never loads it), and a process that already holds RCCL (PyTorch's bundled copy has the same soname) shares that instance
// instead of loading a second one.
#include <dlfcn.h>
#include <string.h>
#include <mutex>

#include "common.h"

namespace {

// the slice of rccl.h this file uses (ABI-stable NCCL 2 entry points)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                                  // 0 = ncclSuccess
constexpr int kNcclFloat32 = 7, kNcclSum = 0, kNcclAvg = 4;

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
        r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy && r.CommCount && r.GetErrorString;
    });
    return r;
}

int fail(const char* what, ncclResult_t rc) { return ha2g_set_error(-3, "%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString(rc)); }

}  // namespace

extern "C" {

int ha2g_comm_available(void) { return rccl().ok ? 1 : 0; }

// rank 0: fill id128 (128 bytes, host memory) -- distribute it to every rank before ha2g_comm_init
int ha2g_comm_unique_id(void* id128) {
    HA2G_REQUIRE(rccl().ok, "comm: librccl.so.1 not found");
    HA2G_REQUIRE(id128 != nullptr, "comm_unique_id: null");
    ncclUniqueId id;
    ncclResult_t rc = rccl().GetUniqueId(&id);
    if (rc) return fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof id);
    return 0;
}
// collective over all `world` ranks, each on its own process and current HIP device; *comm receives the handle
int ha2g_comm_init(const void* id128, int rank, int world, void** comm) {
    HA2G_REQUIRE(rccl().ok, "comm: librccl.so.1 not found");
    HA2G_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments (rank %d of %d)", rank, world);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t rc = rccl().CommInitRank(&c, world, id, rank);
    if (rc) return fail("ncclCommInitRank", rc);
    *comm = c;
    return 0;
}
// ranks RCCL itself sees behind the handle (what bench.py prints as rccl_world)
int ha2g_comm_world(void* comm) {
    HA2G_REQUIRE(rccl().ok && comm, "comm_world: no communicator");
    int n = 0;
    ncclResult_t rc = rccl().CommCount((ncclComm_t)comm, &n);
    if (rc) return fail("ncclCommCount", rc);
    return n;
}
// buf[0..n) (fp32, device) <- sum over ranks (average = 0) or mean over ranks (average = 1), in place, enqueued on `stream`: one call per
// bucket, the collective overlaps whatever the caller enqueues on other streams
int ha2g_allreduce_bucket(void* comm, float* buf, long n, int average, void* stream) {
    HA2G_REQUIRE(rccl().ok && comm, "allreduce_bucket: no communicator");
    HA2G_REQUIRE(n >= 0 && (buf != nullptr || n == 0), "allreduce_bucket: bad buffer");
    if (n == 0) return 0;
    ncclResult_t rc = rccl().AllReduce(buf, buf, (size_t)n, kNcclFloat32, average ? kNcclAvg : kNcclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (rc) return fail("ncclAllReduce", rc);
    return 0;
}
int ha2g_comm_destroy(void* comm) {
    if (!comm) return 0;
    HA2G_REQUIRE(rccl().ok, "comm: librccl.so.1 not found");
    ncclResult_t rc = rccl().CommDestroy((ncclComm_t)comm);
    if (rc) return fail("ncclCommDestroy", rc);
    return 0;
}

}  // extern "C"
