// Shared pieces of the exact top-k search (search.hip, search_f16.hip).
#pragma once
#include "kernels.h"

namespace pfann {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int CAP = 8192;   // survivor slots per query row

__device__ __forceinline__ unsigned f2ord(float f) {   // monotone float -> uint
    const unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
// ascending sort of packed keys == descending score, ascending row
__device__ __forceinline__ unsigned long long pack_key(float score, unsigned row) {
    return ((unsigned long long)(~f2ord(score)) << 32) | row;
}


struct ScanParams {
    const float *q, *db;
    int64_t nq, nrows;       // nrows = rows scanned at this level = ceil(N / stride)
    int64_t row_stride;      // db row step (level stride)
    int d;
    const float *thr;        // [nq] or nullptr (= emit everything, densely: slot = row index)
    int *cnt;                // [nq]
    unsigned long long *keys;  // [nq][CAP]
    int n_tiles_m;
    int nsub;                // survivor sub-lists per query row (small-batch kernel: 32), else 1
    float *gmax;             // small-batch kernel, group-maximum mode: [nq][gridDim.x * 4]
};


__device__ inline void bitonic_sort_u64(unsigned long long *s, int P, int tid, int nt) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nt) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s[i], b = s[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// ---- fp16 pre-filter path (search_f16.hip) ------------------------------------------------
int launch_rows_to_half(const float *x, int64_t n, int d, void *xh, float *norm_max_dev, hipStream_t s);
int launch_q_prep(const float *q, int64_t nq, int d, float xnorm_max, void *qh, float *eps, int *row_ovf, hipStream_t s);
int launch_scan_f16_gmax(const void *dbh, int64_t n, int d, int64_t stride, const void *qh, int64_t nq, int k,
                         SearchWorkspace &ws, int *n_groups_out, hipStream_t s);
int launch_scan_f16(const void *dbh, int64_t n, int d, int64_t stride, const void *qh, int64_t nq,
                    const float *thr_adj, SearchWorkspace &ws, bool allow_sublists, int *nsub_out, hipStream_t s);
// rescore = 1: survivors within 2 eps of the k-th best approximate score are re-scored in exact fp32 from db32;
// rescore = 0: the keys' scores are final (eps is not read)
int launch_select_rescore(SearchWorkspace &ws, int64_t nq, int k, int mode, float *D, int64_t *I, int64_t label_base,
                          const float *q32, const float *db32, int d, int nsub, int rescore, hipStream_t s);


}  // namespace pfann
