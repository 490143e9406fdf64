// Recording communicator - TEST INFRASTRUCTURE, not part of the product.
//
// A stand-in for librccl.so.1 that the library binds when SNERF_RCCL_LIB points here (csrc/dp_comm.hip).  It exports the nine
// nccl* symbols the library resolves.  ncclAllReduce records (buffer, count, stream, group) and scales the range by
// 1 / FAKE_RCCL_NRANKS on the stream it was given - the average over FAKE_RCCL_NRANKS ranks of which all but this one brought
// a zero gradient.  With it the GPU suite checks, on a 1-GPU box, what a world-size-1 RCCL communicator cannot show (its average is
// the identity; VERDICT r05 "What's weak" #1): that every element of the flat gradient buffer is reduced exactly once, in which
// order, on which stream, and that the optimiser runs behind the reduction (tests/test_gpu_round6.py).
//
// build: hipcc --offload-arch=gfx950 -shared -fPIC -O2 tests/native/fake_rccl.cpp -o tests/native/libfake_rccl.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace {
struct Comm {
    int nranks, rank;
};
struct Record {
    int64_t ptr, count, stream, group, seq;
};
std::mutex g_mu;
std::vector<Record> g_log;
int g_depth = 0;
int64_t g_group = 0, g_seq = 0;

int nranks_override() {
    const char *e = getenv("FAKE_RCCL_NRANKS");
    return e && atoi(e) > 0 ? atoi(e) : 0;
}

__global__ void scale_kernel(const float *src, float *dst, size_t n, float f) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] * f;
}
}  // namespace

#define FAKE_API extern "C" __attribute__((visibility("default")))

struct ncclUniqueId {
    char internal[128];
};

FAKE_API int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return 4;
    for (int i = 0; i < 128; ++i) id->internal[i] = (char)(i * 7 + 1);
    return 0;
}
FAKE_API int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return 4;
    for (int i = 0; i < 128; ++i)
        if (id.internal[i] != (char)(i * 7 + 1)) return 4;   // (the id the caller carried between the ranks)
    *comm = new Comm{nranks_override() ? nranks_override() : nranks, rank};
    return 0;
}
FAKE_API int ncclCommDestroy(void *comm) {
    delete static_cast<Comm *>(comm);
    return 0;
}
FAKE_API int ncclCommCount(const void *comm, int *count) {
    if (!comm || !count) return 4;
    *count = static_cast<const Comm *>(comm)->nranks;
    return 0;
}
FAKE_API int ncclCommUserRank(const void *comm, int *rank) {
    if (!comm || !rank) return 4;
    *rank = static_cast<const Comm *>(comm)->rank;
    return 0;
}
FAKE_API int ncclGroupStart() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth++ == 0) ++g_group;
    return 0;
}
FAKE_API int ncclGroupEnd() {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_depth <= 0) return 5;
    --g_depth;
    return 0;
}
FAKE_API int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream) {
    if (!send || !recv || !comm) return 4;
    if (dtype != 7 || op != 4) return 4;   // ncclFloat32, ncclAvg: the only form the library issues
    {
        std::lock_guard<std::mutex> lock(g_mu);
        g_log.push_back(Record{(int64_t)(uintptr_t)send, (int64_t)count, (int64_t)(uintptr_t)stream, g_depth > 0 ? g_group : ++g_group, g_seq++});
    }
    if (count == 0) return 0;
    const float f = 1.0f / (float)static_cast<Comm *>(comm)->nranks;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, static_cast<const float *>(send),
                       static_cast<float *>(recv), count, f);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
FAKE_API const char *ncclGetErrorString(int code) { return code == 0 ? "no error" : "fake_rccl: invalid argument or launch failure"; }

// ---- what the tests read ----------------------------------------------------------------------
// out: up to `max` records of five int64 each (ptr, count, stream, group, seq); returns the number of records logged so far
FAKE_API int64_t fake_rccl_log(int64_t *out, int64_t max) {
    std::lock_guard<std::mutex> lock(g_mu);
    const int64_t n = (int64_t)g_log.size();
    for (int64_t i = 0; i < n && i < max; ++i) memcpy(out + 5 * i, &g_log[(size_t)i], sizeof(Record));
    return n;
}
FAKE_API void fake_rccl_reset() {
    std::lock_guard<std::mutex> lock(g_mu);
    g_log.clear();
}
