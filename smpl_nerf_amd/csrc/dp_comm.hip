// Data-parallel training inside the boundary ("ncclAllReduce on the compute stream right after the backward"): the
// reference trains on one GPU (solver/nerf_solver.py:39); rays of independent images shard over the GPUs of a node and the only
// exchange of the path is the average of the replicated nets' gradients.  Here the step stays ONE call with more than one rank:
// forward -> loss -> backward -> ncclAllReduce(ncclAvg) of the flat gradient buffer on the caller's stream(s) -> Adam, nothing
// synchronised, graph-capturable like the single-GPU step.
//
// RCCL is bound at run time (dlopen / dlsym of librccl.so.1 - the copy the process already holds, e.g. PyTorch's, else the
// system's): a host that never trains data-parallel never loads it, and the C hosts of the single-GPU entry points link as before.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "snerf_common.h"

namespace snerf {

// the part of rccl.h this file needs (ncclResult_t = int, 0 = success; ncclFloat32 = 7; ncclAvg = 4; the unique id is 128 bytes)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == SNERF_COMM_ID_BYTES, "snerf_comm_unique_id hands out an ncclUniqueId");
constexpr int NCCL_FLOAT32 = 7, NCCL_AVG = 4;

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

static const Rccl *rccl(const char *what) {
    static Rccl r;
    static std::once_flag once;
    static char err[256] = "";
    std::call_once(once, [] {
        // SNERF_RCCL_LIB=<path>: this library and no other (a site build of RCCL, or the recording communicator of
        // tests/native/fake_rccl.cpp through which the GPU suite checks what the data-parallel steps reduce, and on which stream)
        if (const char *path = getenv("SNERF_RCCL_LIB"); path && path[0]) {
            r.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) {
                snprintf(err, sizeof err, "cannot load SNERF_RCCL_LIB=%s (%s)", path, dlerror());
                return;
            }
        }
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)   // the copy this process already holds (PyTorch ships one), if any
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            snprintf(err, sizeof err, "cannot load librccl.so.1 (%s)", dlerror());
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p && !err[0]) snprintf(err, sizeof err, "librccl has no symbol %s", name);
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    if (err[0]) {
        fail(SNERF_E_LAUNCH, "%s: %s", what, err);
        return nullptr;
    }
    return &r;
}
static int nccl_fail(const Rccl *r, const char *what, int code) {
    return fail(SNERF_E_LAUNCH, "%s: RCCL error %d (%s)", what, code, r->GetErrorString ? r->GetErrorString(code) : "?");
}

// averages buf[begin, end) minus [skip_begin, skip_end) over the ranks of `comm`, in place, on `stream` (one grouped launch)
int dp_allreduce_avg(snerf_comm_t comm, float *buf, int64_t begin, int64_t end, int64_t skip_begin, int64_t skip_end, hipStream_t stream,
                     const char *what) {
    const Rccl *r = rccl(what);
    if (!r) return SNERF_E_LAUNCH;
    if (!comm || !buf) return fail(SNERF_E_BADARG, "%s: null communicator / buffer", what);
    int64_t seg[2][2] = {{begin, end}, {0, 0}};
    if (skip_begin < skip_end && skip_begin < end && skip_end > begin) {
        seg[0][1] = skip_begin > begin ? skip_begin : begin;
        seg[1][0] = skip_end < end ? skip_end : end;
        seg[1][1] = end;
    }
    int rc = r->GroupStart();
    if (rc) return nccl_fail(r, what, rc);
    for (auto &sg : seg)
        if (sg[1] > sg[0] && (rc = r->AllReduce(buf + sg[0], buf + sg[0], (size_t)(sg[1] - sg[0]), NCCL_FLOAT32, NCCL_AVG,
                                                  reinterpret_cast<ncclComm_t>(comm), stream))) {
            (void)r->GroupEnd();
            return nccl_fail(r, what, rc);
        }
    if ((rc = r->GroupEnd())) return nccl_fail(r, what, rc);
    return SNERF_OK;
}

}  // namespace snerf

extern "C" int snerf_comm_unique_id(void *id_host) {
    using namespace snerf;
    const Rccl *r = rccl("comm_unique_id");
    if (!r) return SNERF_E_LAUNCH;
    if (!id_host) return fail(SNERF_E_BADARG, "comm_unique_id: id_host is null");
    const int rc = r->GetUniqueId(reinterpret_cast<ncclUniqueId *>(id_host));
    return rc ? nccl_fail(r, "comm_unique_id", rc) : SNERF_OK;
}

extern "C" int snerf_comm_init_rank(const void *id_host, int world_size, int rank, snerf_comm_t *comm) {
    using namespace snerf;
    const Rccl *r = rccl("comm_init_rank");
    if (!r) return SNERF_E_LAUNCH;
    if (!id_host || !comm || world_size < 1 || rank < 0 || rank >= world_size) return fail(SNERF_E_BADARG, "comm_init_rank: bad arguments");
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof id);
    ncclComm_t c = nullptr;
    const int rc = r->CommInitRank(&c, world_size, id, rank);
    if (rc) return nccl_fail(r, "comm_init_rank", rc);
    *comm = reinterpret_cast<snerf_comm_t>(c);
    return SNERF_OK;
}

extern "C" int snerf_comm_destroy(snerf_comm_t comm) {
    using namespace snerf;
    if (!comm) return SNERF_OK;
    const Rccl *r = rccl("comm_destroy");
    if (!r) return SNERF_E_LAUNCH;
    const int rc = r->CommDestroy(reinterpret_cast<ncclComm_t>(comm));
    return rc ? nccl_fail(r, "comm_destroy", rc) : SNERF_OK;
}

extern "C" int snerf_comm_info(snerf_comm_t comm, int32_t *world_size, int32_t *rank) {
    using namespace snerf;
    const Rccl *r = rccl("comm_info");
    if (!r) return SNERF_E_LAUNCH;
    if (!comm) return fail(SNERF_E_BADARG, "comm_info: comm is null");
    int w = 0, k = 0, rc;
    if ((rc = r->CommCount(reinterpret_cast<ncclComm_t>(comm), &w)) || (rc = r->CommUserRank(reinterpret_cast<ncclComm_t>(comm), &k)))
        return nccl_fail(r, "comm_info", rc);
    if (world_size) *world_size = w;
    if (rank) *rank = k;
    return SNERF_OK;
}

extern "C" int snerf_comm_allreduce_avg_f32(snerf_comm_t comm, float *buf, int64_t n, snerf_stream_t stream) {
    if (n < 0) return snerf::fail(SNERF_E_BADARG, "comm_allreduce_avg: negative n");
    if (n == 0) return SNERF_OK;
    return snerf::dp_allreduce_avg(comm, buf, 0, n, 0, 0, (hipStream_t)stream, "comm_allreduce_avg");
}
