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
    // small-batch group-maximum pass with the query preparation folded in (ELT = 2: q32 = the fp32 query rows, p.q unused;
    // what q_prep_kernel does, one launch less): workgroup 0 leaves qh / eps / row_ovf for the launches that follow
    const float *q32 = nullptr;
    float xnorm_max = 0.f;
    void *qh_out = nullptr;  // [nq][d] fp16 query rows
    float *eps = nullptr;    // [nq]
    int *row_ovf = nullptr;
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

// Exact fallback for a flagged row (body of topk_fallback_kernel; NT threads).  Returns at once when row_ovf[m] == 0.
template <int ELT, int NT>
__device__ inline void topk_fallback_body(int64_t m, int *row_ovf, const float *__restrict__ q, const void *__restrict__ dbv,
                                          int64_t n, int d, int k, float *D, int64_t *I, int64_t label_base) {
    constexpr int FB = 2048, RPP = NT / 2;        // buffer slots; rows per pass (NT / 4 row groups x 2)
    __shared__ unsigned long long buf[FB];
    __shared__ float qs[1024];
    __shared__ int s_cnt;
    __shared__ unsigned long long s_T;
    if (row_ovf[m] == 0) return;
    // FOUR lanes per row, partial sums over e = 4 sub, 4 sub + 16, ..., combined as (p0 + p1) + (p2 + p3): the summation
    // order of the select kernels' exact re-scoring (search_f16.hip), so a row's fp32 score has the same bits whether it comes
    // out of a select or out of this fallback (round 6: with eight lanes per row a list overflow on one path of a sharded
    // search moved a score by one ulp against the single-shard run)
    const int tid = threadIdx.x, sub = tid & 3, grp = tid >> 2;
    for (int e = tid; e < d; e += NT) qs[e] = ELT == 4 ? q[m * d + e] : (float)(_Float16)q[m * d + e];
    if (tid == 0) { s_cnt = 0; s_T = ~0ull; }
    __syncthreads();
    for (int64_t base = 0; base < n; base += RPP) {
        const unsigned long long T = s_T;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t row = base + u * (NT / 4) + grp;
            float part = 0.f;
            if (row < n) {
                for (int e = sub * 4; e < d; e += 16) {
                    float x0, x1, x2, x3;
                    if (ELT == 4) {
                        const float4 x4 = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(dbv) + row * d + e);
                        x0 = x4.x; x1 = x4.y; x2 = x4.z; x3 = x4.w;
                    } else {
                        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                        const f16x4 h4 = *reinterpret_cast<const f16x4 *>(reinterpret_cast<const _Float16 *>(dbv) + row * d + e);
                        x0 = (float)h4[0]; x1 = (float)h4[1]; x2 = (float)h4[2]; x3 = (float)h4[3];
                    }
                    part = fmaf(x0, qs[e], part); part = fmaf(x1, qs[e + 1], part);
                    part = fmaf(x2, qs[e + 2], part); part = fmaf(x3, qs[e + 3], part);
                }
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            if (sub == 0 && row < n) {
                const unsigned long long key = pack_key(part, (unsigned)row);
                if (key < T) buf[atomicAdd(&s_cnt, 1)] = key;       // s_cnt <= FB - RPP before the pass
            }
        }
        __syncthreads();
        if (s_cnt > FB - RPP || base + RPP >= n) {                    // block-uniform
            const int c = s_cnt;
            for (int i = c + tid; i < FB; i += NT) buf[i] = ~0ull;
            __syncthreads();
            bitonic_sort_u64(buf, FB, tid, NT);
            if (tid == 0) {
                s_cnt = c < k ? c : k;
                s_T = c >= k ? buf[k - 1] : ~0ull;
            }
            __syncthreads();
        }
    }
    const int c = s_cnt;
    for (int i = tid; i < k; i += NT) {
        if (i < c) {
            D[m * k + i] = ord2f(~(unsigned)(buf[i] >> 32));
            I[m * k + i] = (int64_t)(unsigned)(buf[i] & 0xFFFFFFFFu) + label_base;
        } else {
            D[m * k + i] = -3.4028234663852886e38f;
            I[m * k + i] = -1;
        }
    }
    if (tid == 0) row_ovf[m] = 0;
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
                          const float *q32, const float *db32, int d, int nsub, int rescore, hipStream_t s,
                          bool few_survivors = false);
// small-batch path (search_small): the 256-thread select alone, then big select (rows with > SMALL_N survivors) + exact
// fallback of flagged rows in ONE launch
int launch_select_rescore_small(SearchWorkspace &ws, int64_t nq, int k, float *D, int64_t *I, int64_t label_base,
                                const float *q32, const float *db32, int d, int nsub, int rescore, hipStream_t s);
int launch_select_tail(SearchWorkspace &ws, int64_t nq, int k, float *D, int64_t *I, int64_t label_base, const float *q32,
                       const float *db32, const void *fb_rows, int fb_elt, int64_t n, int d, int nsub, int rescore, hipStream_t s);


}  // namespace pfann
