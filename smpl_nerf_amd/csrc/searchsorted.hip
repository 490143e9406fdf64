// snerf_searchsorted / snerf_searchsorted_f32 - batched binary search (a6), every scalar type the reference dispatches
// (AT_DISPATCH_ALL_TYPES at searchsorted_cpu_wrapper.cpp:100 / searchsorted_cuda_kernel.cu:132: uint8, int8, int16, int32,
// int64, float, double; `a` and `v` share the type).
//
// Replaces torchsearchsorted's native op (searchsorted_cpu_wrapper.cpp:82-122 /
// searchsorted_cuda_kernel.cu:84-141): out[r,c] = number of entries of the sorted row a[r] that
// are < v[r,c] (side left) or <= v[r,c] (side right), with row broadcast when either operand has
// one row.  HBM-bound: a row of `a` is read once into LDS per workgroup, `v` is streamed, and the
// int64 result is the dominant traffic (8 B per query).
//
// Layout: one workgroup owns ROWS consecutive result rows x all their queries when the rows are
// short (the NeRF shape: a[B,63], v[B,128]) so that a 256-thread group always has 4 full waves
// of work; long rows fall back to one row per workgroup, grid.y tiling over the queries.
#include "snerf_common.h"

namespace snerf {

constexpr int SS_THREADS = 256;
constexpr int SS_LDS_BYTES = 32768;  // 32 KiB of `a` rows per workgroup

// Branch-free bisection over a sorted LDS/global row: returns #{k: a[k] < v} or #{k: a[k] <= v}.
template <bool LEFT, typename T>
__device__ __forceinline__ int bisect(const T *__restrict__ row, int n, T v) {
    int lo = 0, len = n;
    while (len > 0) {
        int half = len >> 1;
        T m = row[lo + half];
        bool go_right = LEFT ? (m < v) : (m <= v);
        lo = go_right ? lo + half + 1 : lo;
        len = go_right ? len - half - 1 : half;
    }
    return lo;
}

template <bool LEFT, typename T>
__global__ __launch_bounds__(SS_THREADS) void searchsorted_rows_kernel(
    const T *__restrict__ a, int64_t nrow_a, int ncol_a, const T *__restrict__ v, int64_t nrow_v,
    int ncol_v, int64_t *__restrict__ out, int64_t nrow, int rows_per_block) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[SS_LDS_BYTES];
    T *s_a = reinterpret_cast<T *>(s_raw);
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
    const int nrows = (int)min((int64_t)rows_per_block, nrow - row0);
    // stage the rows of `a` this group needs (one row only if `a` is broadcast)
    const int a_rows = nrow_a == 1 ? 1 : nrows;
    const T *a_src = a + (nrow_a == 1 ? 0 : row0 * ncol_a);
    for (int i = threadIdx.x; i < a_rows * ncol_a; i += SS_THREADS) s_a[i] = a_src[i];
    __syncthreads();
    const int total = nrows * ncol_v;
    for (int i = threadIdx.x; i < total; i += SS_THREADS) {
        const int r = i / ncol_v, c = i - r * ncol_v;
        const T q = v[(nrow_v == 1 ? 0 : (row0 + r) * (int64_t)ncol_v) + c];
        const T *row = s_a + (nrow_a == 1 ? 0 : r * ncol_a);
        out[(row0 + r) * (int64_t)ncol_v + c] = bisect<LEFT, T>(row, ncol_a, q);
    }
}

// long rows: one result row per blockIdx.x, queries tiled over blockIdx.y; `a` read from L2/HBM
template <bool LEFT, typename T>
__global__ __launch_bounds__(SS_THREADS) void searchsorted_long_kernel(
    const T *__restrict__ a, int64_t nrow_a, int64_t ncol_a, const T *__restrict__ v, int64_t nrow_v,
    int64_t ncol_v, int64_t *__restrict__ out) {
    const int64_t r = blockIdx.x;
    const T *row = a + (nrow_a == 1 ? 0 : r * ncol_a);
    const T *vr = v + (nrow_v == 1 ? 0 : r * ncol_v);
    for (int64_t c = (int64_t)blockIdx.y * SS_THREADS + threadIdx.x; c < ncol_v; c += (int64_t)gridDim.y * SS_THREADS) {
        const T q = vr[c];
        int64_t lo = 0, len = ncol_a;
        while (len > 0) {
            int64_t half = len >> 1;
            T m = row[lo + half];
            bool go_right = LEFT ? (m < q) : (m <= q);
            lo = go_right ? lo + half + 1 : lo;
            len = go_right ? len - half - 1 : half;
        }
        out[r * ncol_v + c] = lo;
    }
}

template <typename T>
static int launch_searchsorted(const T *a, int64_t nrow_a, int64_t ncol_a, const T *v, int64_t nrow_v, int64_t ncol_v,
                               int64_t *out, int side_left, hipStream_t s) {
    if (nrow_a < 0 || nrow_v < 0 || ncol_a < 0 || ncol_v < 0) return fail(SNERF_E_BADARG, "searchsorted: negative size");
    if (!(nrow_a == nrow_v || nrow_a == 1 || nrow_v == 1))
        return fail(SNERF_E_BADARG, "searchsorted: `a` and `v` must have the same number of rows or one of them one row");
    const int64_t nrow = nrow_a > nrow_v ? nrow_a : nrow_v;
    if (nrow == 0 || ncol_v == 0) return SNERF_OK;
    if (!a && ncol_a > 0) return fail(SNERF_E_BADARG, "searchsorted: a is null");
    if (!v || !out) return fail(SNERF_E_BADARG, "searchsorted: v/out is null");
    if (!aligned(a, sizeof(T)) || !aligned(v, sizeof(T)) || !aligned(out, 8))
        return fail(SNERF_E_ALIGN, "searchsorted: pointers must be aligned to their element size");
    constexpr int64_t LDS_ELEMS = SS_LDS_BYTES / (int)sizeof(T);
    if (ncol_a <= LDS_ELEMS && ncol_v <= (1 << 20)) {
        // rows per group: enough queries for 4 waves x 4 items, bounded by the LDS budget
        int64_t rpb = (4 * SS_THREADS + ncol_v - 1) / ncol_v;
        if (ncol_a > 0) rpb = rpb < LDS_ELEMS / ncol_a ? rpb : LDS_ELEMS / ncol_a;
        if (rpb < 1) rpb = 1;
        if (rpb > nrow) rpb = nrow;
        const int64_t grid = (nrow + rpb - 1) / rpb;
        if (grid > 0x7fffffffLL) return fail(SNERF_E_BADARG, "searchsorted: too many rows");
        if (side_left)
            hipLaunchKernelGGL((searchsorted_rows_kernel<true, T>), dim3((unsigned)grid), dim3(SS_THREADS), 0, s, a, nrow_a,
                               (int)ncol_a, v, nrow_v, (int)ncol_v, out, nrow, (int)rpb);
        else
            hipLaunchKernelGGL((searchsorted_rows_kernel<false, T>), dim3((unsigned)grid), dim3(SS_THREADS), 0, s, a, nrow_a,
                               (int)ncol_a, v, nrow_v, (int)ncol_v, out, nrow, (int)rpb);
    } else {
        if (nrow > 0x7fffffffLL) return fail(SNERF_E_BADARG, "searchsorted: too many rows");
        int64_t gy = (ncol_v + SS_THREADS - 1) / SS_THREADS;
        if (gy > 65535) gy = 65535;
        if (side_left)
            hipLaunchKernelGGL((searchsorted_long_kernel<true, T>), dim3((unsigned)nrow, (unsigned)gy), dim3(SS_THREADS), 0, s,
                               a, nrow_a, ncol_a, v, nrow_v, ncol_v, out);
        else
            hipLaunchKernelGGL((searchsorted_long_kernel<false, T>), dim3((unsigned)nrow, (unsigned)gy), dim3(SS_THREADS), 0, s,
                               a, nrow_a, ncol_a, v, nrow_v, ncol_v, out);
    }
    return check_launch("searchsorted");
}

}  // namespace snerf

extern "C" int snerf_searchsorted_f32(const float *a, int64_t nrow_a, int64_t ncol_a, const float *v,
                                      int64_t nrow_v, int64_t ncol_v, int64_t *out, int side_left,
                                      snerf_stream_t stream) {
    return snerf::launch_searchsorted<float>(a, nrow_a, ncol_a, v, nrow_v, ncol_v, out, side_left, (hipStream_t)stream);
}

extern "C" int snerf_searchsorted(int dtype, const void *a, int64_t nrow_a, int64_t ncol_a, const void *v, int64_t nrow_v,
                                  int64_t ncol_v, int64_t *out, int side_left, snerf_stream_t stream) {
    using namespace snerf;
    hipStream_t s = (hipStream_t)stream;
#define SNERF_SS_CASE(code, T) \
    case code:                 \
        return launch_searchsorted<T>(static_cast<const T *>(a), nrow_a, ncol_a, static_cast<const T *>(v), nrow_v, ncol_v, out, side_left, s)
    switch (dtype) {
        SNERF_SS_CASE(SNERF_DTYPE_F32, float);
        SNERF_SS_CASE(SNERF_DTYPE_F64, double);
        SNERF_SS_CASE(SNERF_DTYPE_I32, int32_t);
        SNERF_SS_CASE(SNERF_DTYPE_I64, int64_t);
        SNERF_SS_CASE(SNERF_DTYPE_I16, int16_t);
        SNERF_SS_CASE(SNERF_DTYPE_I8, int8_t);
        SNERF_SS_CASE(SNERF_DTYPE_U8, uint8_t);
    }
#undef SNERF_SS_CASE
    return fail(SNERF_E_BADARG, "searchsorted: unknown dtype code %d", dtype);
}
