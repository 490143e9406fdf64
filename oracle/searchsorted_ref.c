/* CPU oracle (TEST INFRASTRUCTURE ONLY): C restatement of the reference's native searchsorted.
 *
 * Follows torchsearchsorted/src/cpu/searchsorted_cpu_wrapper.cpp of HannesStark/SMPL-NeRF:
 *   eval()                      :5-39   three-way test  a[col] (<|<=) val (<=|<) a[col+1]
 *   binary_search()             :42-80  bisection that returns -1 below the row, ncol-1 above
 *   searchsorted_cpu_wrapper()  :82-122 serial row/column loops, row broadcast when nrow == 1,
 *                                       result = binary_search(...) + 1
 * The reference cannot be compiled in this image (ATen API break at :100 against torch 2.10), so
 * this file restates the algorithm for float32 and is pinned in tests/test_oracle_searchsorted.py
 * against numpy.searchsorted - the reference's own test oracle
 * (torchsearchsorted/test/test_searchsorted.py:41-44).
 *
 * Built by oracle/Makefile into oracle/_build/libsearchsorted_ref.so.  Never linked by the product.
 */
#include <stdint.h>

static int ss_eval(float val, const float *a, int64_t row, int64_t col, int64_t ncol, int side_left)
{
    if (col == ncol - 1) {                 /* right border (:10-17) */
        return (a[row * ncol + col] <= val) ? 1 : -1;
    }
    int is_lower, is_next_higher;
    if (side_left) {                       /* a[col] < v <= a[col+1] (:21-24) */
        is_lower = a[row * ncol + col] < val;
        is_next_higher = a[row * ncol + col + 1] >= val;
    } else {                               /* a[col] <= v < a[col+1] (:25-29) */
        is_lower = a[row * ncol + col] <= val;
        is_next_higher = a[row * ncol + col + 1] > val;
    }
    if (is_lower && is_next_higher) return 0;
    return is_lower ? 1 : -1;
}

static int64_t ss_binary_search(const float *a, int64_t row, float val, int64_t ncol, int side_left)
{
    int64_t right = ncol, left = 0;        /* :60-61 */
    while (right >= left) {
        int64_t mid = left + (right - left) / 2;
        int rel = ss_eval(val, a, row, mid, ncol, side_left);
        if (rel == 0) return mid;
        if (rel > 0) {
            if (mid == ncol - 1) return ncol - 1;
            left = mid;
        } else {
            if (mid == 0) return -1;
            right = mid;
        }
    }
    return -1;
}

/* out[max(nrow_a,nrow_v), ncol_v] int64; returns 0. */
int searchsorted_ref_f32(const float *a, int64_t nrow_a, int64_t ncol_a,
                         const float *v, int64_t nrow_v, int64_t ncol_v,
                         int64_t *out, int side_left)
{
    int64_t nrow = nrow_a > nrow_v ? nrow_a : nrow_v;
    for (int64_t row = 0; row < nrow; ++row) {
        for (int64_t col = 0; col < ncol_v; ++col) {
            int64_t rv = (nrow_v == 1) ? 0 : row;
            int64_t ra = (nrow_a == 1) ? 0 : row;
            out[row * ncol_v + col] =
                ss_binary_search(a, ra, v[rv * ncol_v + col], ncol_a, side_left) + 1;
        }
    }
    return 0;
}
