// fp16 pre-filter for the batched exact top-k (gfx950).
//
// With thousands of query rows per pass the exact-fp32 scan is bound by the fp32 MFMA rate
// (1/16 of the fp16 rate).  This path keeps the RESULT exact while moving the O(Q*N*d) work to
// v_mfma_f32_32x32x16_f16:
//
//   1. db rows and query rows are rounded to fp16 once (db at load, queries per call);
//      s16 = sum fl16(q_i) * fl16(x_i) with exact products and fp32 accumulation;
//   2. |s16 - s| <= eps_m := 1.05e-3 * ||q_m|| * X + 3.1e-8 * sqrt(d) * (||q_m|| + X),  X = max_n ||x_n||:
//      each rounding is |dv| <= max(2^-11 |v|, 2^-25) (normal / subnormal fp16), so
//      sum |fl(q_i) fl(x_i) - q_i x_i| <= 2^-10 sum |q_i x_i| + 2^-25 (||q||_1 + ||x||_1) + 2nd order
//      <= 2^-10 ||q|| ||x|| + 2^-25 sqrt(d) (||q|| + ||x||) (Cauchy-Schwarz); the factor 1.05e-3 > 2^-10 = 9.77e-4
//      leaves room for the second-order term (2^-22) and the fp32 accumulation (d 2^-24 relative, d <= 1024).
//      Both terms scale with the norms: the bound holds for any db scale, not only unit-norm rows.  A query row
//      whose fp16 image would overflow (||q|| >= 6e4, or NaN) is flagged for the exact device fallback instead;
//   3. the scan emits row n for query m iff  s16 >= tau_m - eps_m,  tau_m being a lower bound of
//      the query's k-th best EXACT score (from the sampled levels, as in search.hip): every row
//      of the true top-k survives;
//   4. the select kernel sorts the survivors by s16, re-scores in exact fp32 only those with
//      s16 >= (k-th best s16) - 2 eps  (k rows have s16 >= a, hence exact >= a - eps, so the true
//      k-th best is >= a - eps and every true top-k row has s16 >= a - 2 eps), sorts that handful
//      by exact score and emits the top k.  Thresholds handed to the next level are exact scores.
//
// The answer is the exact fp32 top-k; only the arithmetic that cannot change it runs in fp16.
// Reference precedent for fp16 search: faiss GpuMultipleClonerOptions.useFloat16
// (database.py:102-104), there without re-scoring, i.e. approximate.
#include "search_common.h"
#include <stdlib.h>

namespace pfann {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------- conversions
__global__ void rows_to_half_kernel(const float *__restrict__ x, int64_t n, int d, _Float16 *__restrict__ xh,
                                    float *norm_max) {
    // one wave per row: convert, and fold the row's L2 norm into a global max (non-negative
    // floats order like their bit patterns)
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    float ss = 0.f;
    for (int e = lane; e < d; e += 64) {
        const float v = x[row * d + e];
        xh[row * d + e] = (_Float16)v;
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(norm_max), __float_as_uint(sqrtf(ss)));
}

int launch_rows_to_half(const float *x, int64_t n, int d, void *xh, float *norm_max_dev, hipStream_t s) {
    if (n <= 0) return 0;
    PF_LAUNCH(rows_to_half_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, s, x, n, d,
                       reinterpret_cast<_Float16 *>(xh), norm_max_dev);
    PF_HIP(hipGetLastError());
    return 0;
}

__global__ void q_prep_kernel(const float *__restrict__ q, int64_t nq, int d, float xnorm_max,
                              _Float16 *__restrict__ qh, float *__restrict__ eps, int *__restrict__ row_ovf) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= nq) return;
    float ss = 0.f;
    for (int e = lane; e < d; e += 64) {
        const float v = q[row * d + e];
        ss = fmaf(v, v, ss);
    }
    ss = wave_sum(ss);
    const bool bad = !(ss < 3.6e9f);                       // ||q|| >= 6e4 or NaN: outside fp16's range
    for (int e = lane; e < d; e += 64) qh[row * d + e] = bad ? (_Float16)0.f : (_Float16)q[row * d + e];
    if (lane == 0) {
        const float nq2 = sqrtf(ss);
        eps[row] = 1.05e-3f * nq2 * xnorm_max + 3.1e-8f * sqrtf((float)d) * (nq2 + xnorm_max);
        if (bad) row_ovf[row] = 1;                          // recomputed exactly by topk_fallback_kernel
    }
}

int launch_q_prep(const float *q, int64_t nq, int d, float xnorm_max, void *qh, float *eps, int *row_ovf, hipStream_t s) {
    PF_LAUNCH(q_prep_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, s, q, nq, d, xnorm_max,
                       reinterpret_cast<_Float16 *>(qh), eps, row_ovf);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// fp16 scan: same tiling/pipeline as scan_emit_kernel<128,128,64,64,QT> (search.hip) with the
// operands stored as halves: a 128-byte K-tile holds 64 k (vs 32 floats), the 16-byte fragment a
// lane reads feeds ONE v_mfma_f32_32x32x16_f16 (lane half h holds k = 8h..8h+7 of the 16-step).
// p.q / p.db point to fp16 rows, p.thr holds tau - eps (or nullptr on the dense top level).
// ------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(256, 2) void scan_f16_kernel(ScanParams p) {
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64;
    constexpr int LDK = 36;                   // LDS row pitch in dwords (128 B of k + 16 B pad)
    constexpr int WAVES_N = BN / WN, TM = WM / 32, TN = WN / 32, AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BN * LDK];
    __shared__ __attribute__((aligned(16))) float thr_s[QT * BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int n_mg = (p.n_tiles_m + QT - 1) / QT;
    const int nt = L / n_mg, mg = L - nt * n_mg;
    const int mt0 = mg * QT;
    const int n_q = min(QT, p.n_tiles_m - mt0);
    const int64_t n0 = (int64_t)nt * BN;
    const int col4 = tid & 7, rowq = tid >> 3;
    const unsigned row_bytes = (unsigned)p.d * 2u;             // one fp16 row

    // thr_s = -(tau - eps): the accumulators START there, so the MFMA chain yields s16 - (tau - eps) and
    // "survivor" is a sign test; query rows past nq start at -inf and never qualify
    const bool dense = p.thr == nullptr;
    for (int i = tid; i < QT * BM; i += 256) {
        const int64_t m = (int64_t)mt0 * BM + i;
        thr_s[i] = dense ? 0.f : (m < p.nq ? -p.thr[m] : -INFINITY);
    }
    const char *qb = reinterpret_cast<const char *>(p.q), *dbb = reinterpret_cast<const char *>(p.db);
    const int64_t mq0 = (int64_t)mt0 * BM;
    const __amdgpu_buffer_rsrc_t srd_q = make_srd(qb + mq0 * row_bytes, (unsigned long long)(p.nq - mq0) * row_bytes);
    // one window per 32-row sub-tile (see scan_emit_kernel): offsets stay below 2 GB at any level stride
    __amdgpu_buffer_rsrc_t srd_db[BR];
    unsigned doff[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int64_t nj = n0 + 32 * j;
        const int64_t left = p.nrows - nj;
        srd_db[j] = make_srd(dbb + (nj < p.nrows ? nj : 0) * p.row_stride * row_bytes,
                             left > 0 ? (unsigned long long)((left - 1) * p.row_stride + 1) * row_bytes : 0ull);
        doff[j] = nj + rowq < p.nrows ? (unsigned)((unsigned long long)rowq * p.row_stride * row_bytes) : BUF_OOB;
    }
    f32x4 ra[AR], rb[BR];
    const int q_rows_left = (int)((p.nq - mq0) < (int64_t)QT * BM ? (p.nq - mq0) : (int64_t)QT * BM);
    const unsigned tile_bytes = (unsigned)BM * row_bytes;
    unsigned aoff[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) aoff[i] = (unsigned)(rowq + 32 * i) * row_bytes;
    unsigned lkb = (unsigned)col4 * 16u;     // byte offset of this thread's 16-byte piece inside the row
    int lrow = rowq;
    unsigned lbase = 0;
    auto load_tile = [&]() {
        const bool kok = lkb < row_bytes;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            ra[i] = buf_load4(srd_q, (kok && lrow + 32 * i < q_rows_left) ? lbase + aoff[i] + lkb : BUF_OOB);
#pragma unroll
        for (int j = 0; j < BR; ++j) rb[j] = buf_load4(srd_db[j], (kok && doff[j] != BUF_OOB) ? doff[j] + lkb : BUF_OOB);
        lkb += 128u;
        const bool wrap = lkb >= row_bytes;
        lkb = wrap ? (unsigned)col4 * 16u : lkb;
        lrow += wrap ? BM : 0;
        lbase += wrap ? tile_bytes : 0u;
    };
    auto store_tile = [&](float *Ad, float *Bd) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4 *>(&Ad[(rowq + 32 * i) * LDK + col4 * 4]) = ra[i];
#pragma unroll
        for (int j = 0; j < BR; ++j)
            *reinterpret_cast<f32x4 *>(&Bd[(rowq + 32 * j) * LDK + col4 * 4]) = rb[j];
    };

    const int nk = (int)((row_bytes + 127u) / 128u);
    load_tile();
    store_tile(As, Bs);
    __syncthreads();
    const int l31 = lane & 31, lhalf = lane >> 5;
    int it = 0;
#pragma unroll 1
    for (int q = 0; q < n_q; ++q) {
        const int64_t m0 = mq0 + (int64_t)q * BM;
        const float *thr_c = thr_s + q * BM;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // C rows of register 4g+e: wm*WM + i*32 + 8g + 4*lhalf + e
                const f32x4 t4 = *reinterpret_cast<const f32x4 *>(thr_c + wm * WM + i * 32 + 8 * g + 4 * lhalf);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = t4[e];
            }
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++it) {
            const float *Ac = As + (it & 1) * (BM * LDK), *Bc = Bs + (it & 1) * (BN * LDK);
            float *An = As + ((it + 1) & 1) * (BM * LDK), *Bn = Bs + ((it + 1) & 1) * (BN * LDK);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {          // 4 MFMA steps of 16 k per 128-byte K-tile
                f16x8 a8[TM], b8[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a8[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(
                                                          &Ac[(wm * WM + i * 32 + l31) * LDK + kk * 8 + lhalf * 4]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b8[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(
                                                          &Bc[(wn * WN + j * 32 + l31) * LDK + kk * 8 + lhalf * 4]));
                if (kk == 0) {
                    load_tile();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kk == 3) store_tile(An, Bn);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[i], b8[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t n = n0 + wn * WN + j * 32 + l31;
            const bool nok = n < p.nrows;
            const unsigned row = (unsigned)(n * p.row_stride);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (dense) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        if (nok && m < p.nq) p.keys[m * CAP + n] = pack_key(acc[i][j][r], row);
                    }
                } else {
                    // fast reject: the largest of this lane's 16 values is negative (a handful of
                    // v_max3 instead of 16 compares; about one survivor per 3000 values gets through)
                    float mx = fmaxf(acc[i][j][0], acc[i][j][1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, acc[i][j][r]), acc[i][j][r + 1]);
                    if (__any(nok && mx >= 0.f)) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (nok && acc[i][j][r] >= 0.f) {
                                const int ml = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                                const int64_t m = m0 + ml;
                                const int pos = atomicAdd(&p.cnt[m], 1);
                                if (pos < CAP) p.keys[m * CAP + pos] = pack_key(acc[i][j][r] - thr_c[ml], row);
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------
// Query-stationary fp16 scan (d = 16*KS <= 128, thresholds present).  A workgroup owns ONE 128-row
// query tile and one of S interleaved slices of the db tiles:
//   * the query tile lives in REGISTERS as MFMA fragments for the whole launch (64 VGPRs at
//     d = 128) and is the SECOND MFMA operand, so a lane's 16 results are 16 db rows against ONE
//     query row: threshold, survivor test (max of the 16 vs tau - eps), list reservation and the
//     sampled pass's running maximum are all lane-local (see "query fragments" in the kernel);
//   * only the 32 KB db tile goes through LDS per step (double buffered, one barrier): half the
//     LDS traffic and staging instructions per MFMA of the generic kernel above, which matters
//     because with d this small the loop is bound by LDS and issue slots, not by the fp16 MFMA rate;
//   * survivors go to lists of the LANE's own: four private lists per (query row, db slice), one per owner (the two waves
//     of the row half x the two lane halves), write position in a register, count written once at the end (round 6; up to
//     round 5 one list per (row, slice) behind an LDS counter; the 15 million device-scope atomics of a db-stationary
//     split cost 5 ms per pass on MI355X, measured).  The select kernels gather the 4 S sub-lists of a row.
// ------------------------------------------------------------------------------------
// GMAX = true (sampled pass, no thresholds): no survivor lists at all -- every (lane, register) keeps the running maximum
// of ITS row position over all db tiles of the slice and the kernel writes 64 group maxima per (query row, slice); the
// k-th best of a row's 64*S group maxima is the score of a real row, hence (minus the rounding margin) a lower bound of its
// k-th best overall.  One such pass over every 4th row + a radix select of the group maxima replace the dense and 1/16
// levels with their two survivor selects.  Groups are interleaved on purpose (see gmx in the kernel).
// DBR = db rows per tile: 128 (two workgroups per CU), or 64 for the full pass: half the LDS and fewer accumulators per
// workgroup, THREE workgroups (12 waves) per CU -- more independent barrier domains to cover a workgroup's epilogue and
// barrier waits with the others' MFMAs, at twice the barriers per MFMA.
// -DPFANN_SCAN_TRACE (tuning builds only, tools/ubench/scan_trace.py): every wave of the full pass sums the shader cycles of
// its step phases (tile request | fragment reads + MFMAs | survivor epilogue | wait for the next tile | barrier) and
// wave 0 / lane 0 of each workgroup leaves the five sums + its step count in a caller-provided buffer.
#ifdef PFANN_SCAN_TRACE
__device__ unsigned long long *g_scan_trace = nullptr;
__device__ unsigned g_scan_trace_cap = 0;
#define SCAN_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tr_sum[i] += now_ - tr_last; tr_last = now_; } while (0)
#else
#define SCAN_STAMP(i)
#endif

// NBUF = db-tile buffers in LDS: 2 (tile t + S travels during the step of tile t), or 3 for launches that leave most of
// the chip's workgroup slots empty (the middle of the batch curve: 33 .. ~1000 query rows = 1 .. 8 query tiles): with
// nobody else on the CU to cover it, a step was one full L2 / HBM round trip long (2400 cycles for 512 cycles of MFMA).
// With every slot taken (the 9728-row launch groups) the third buffer measured +1.6 % (profiles/r6/scan_v3_abc.txt): there
// the other two workgroups of the CU already cover the wait, and the pass is bound by its survivor epilogue.
// (256 query rows per workgroup -- 128 per wave, two waves per SIMD, every 16 KB db tile feeding twice the MFMAs -- measured
// the same 2.56 ms as this 128-row form in round 6, profiles/r6/scan_bm256_ab.txt: the step is not bound by its per-tile
// costs but by what it pays per MFMA: the LDS-latency-bound fragment loop and the survivor epilogue.)
template <int KS, bool GMAX = false, int DBR = 128, int NBUF = 2>
__global__ __launch_bounds__(256, DBR == 64 ? 3 : (NBUF == 3 ? 1 : 2)) void scan_f16_qres_kernel(ScanParams p) {
    constexpr int BM = 128, WM = 64, WN = DBR / 2, TM = 2, TN = WN / 32;
    constexpr bool BAL = !GMAX;                   // balanced row -> owner mapping of the survivor-emitting passes (see `frag`)
    constexpr int ROWB = KS * 32;                 // bytes of one fp16 row
    constexpr int CPR = ROWB / 16;                // 16-byte chunks per row
    constexpr int NLD = DBR * CPR / 256;          // direct-to-LDS loads per thread per db tile
    static_assert(!GMAX || DBR == 128, "the group-maximum pass assumes 64 groups per slice");
    // one LDS object per db-tile buffer: the compiler's wait tracking for LDS loads works per object (alias scopes), so a
    // tile request into ONE buffer does not put `s_waitcnt vmcnt(0)` in front of the fragment reads of the OTHER
    // (Bs0 doubles as the [128][64] group-maxima tile of the GMAX epilogue: sized for it on its own, never relying on Bs1
    // lying behind it -- at d = 64 a db tile is only half that)
    constexpr int TILE_F = DBR * ROWB / 4, GS_F = GMAX ? BM * 64 : 0;
    __shared__ __attribute__((aligned(1024))) float Bs0[TILE_F > GS_F ? TILE_F : GS_F];
    __shared__ __attribute__((aligned(1024))) float Bs1[TILE_F];
    // NBUF = 3: tile t + 2S is requested at the top of the step of tile t, so a tile has TWO steps to arrive
    static_assert(NBUF == 2 || NBUF == 3, "two or three db-tile buffers");
    __shared__ __attribute__((aligned(1024))) float Bs2[NBUF == 3 ? TILE_F : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhalf = lane >> 5;
    const int S = p.nsub, subcap = CAP / (4 * S);       // capacity of one PRIVATE list: a quarter of the (row, slice) list
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = L / p.n_tiles_m, mt = L - seg * p.n_tiles_m;     // neighbours share the db segment (L2)
    const int64_t m0 = (int64_t)mt * BM;
    // db tiles seg, seg + S, seg + 2S, ...: interleaved, so a run of similar rows (one song) is spread
    // over the sub-lists instead of overflowing one
    const int64_t t_lo = seg, t_hi = (p.nrows + DBR - 1) / DBR;
    // this lane's private lists: row ml, slot pos -> keys[(m0 + ml) * CAP + own * subcap + pos], own = 4 seg + 2 wn + lhalf
    const __amdgpu_buffer_rsrc_t srd_k = make_srd(p.keys + m0 * CAP, (unsigned long long)BM * CAP * 8ull);
    const int own = 4 * seg + 2 * wn + lhalf;
    const unsigned own_off = (unsigned)(own * subcap) * 8u;
    int posr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) posr[i] = 0;
    const char *qb = reinterpret_cast<const char *>(p.q), *dbb = reinterpret_cast<const char *>(p.db);

    // ---- query fragments: lane (l31, lhalf) holds k = 16*kk + 8*lhalf .. +7 of query row m0 + wm*64 + i*32 + l31.
    // They are the MFMA's SECOND operand (the db rows the first), so the result tile comes out transposed: a lane owns
    // ONE query row and its 16 registers are 16 db rows (8g + 4*lhalf + e for register 4g + e).  Everything per query
    // row is then lane-local: the threshold is one register instead of a 16-register C operand, "any survivor" is a
    // compare of the lane maximum, a lane's survivors take ONE list reservation (count, then consecutive slots) instead
    // of one LDS atomic each, and the running group maximum of the sampled pass is one register with no cross-lane step.
    const __amdgpu_buffer_rsrc_t srd_q = make_srd(qb + m0 * ROWB, (unsigned long long)(p.nq - m0) * ROWB);
    f16x8 afr[TM][KS];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            afr[i][kk] = __builtin_bit_cast(f16x8, buf_load4(srd_q, (unsigned)(wm * WM + i * 32 + l31) * ROWB + kk * 32 + lhalf * 16));
    float th[TM];                  // tau - eps of this lane's query rows (rows past nq: +inf, never a survivor)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + wm * WM + i * 32 + l31;
        th[i] = GMAX ? 0.f : (m < p.nq ? p.thr[m] : INFINITY);
    }
    // GMAX: one running maximum per (query row, tile-row position): register r of sub-tile (wn, j) is its own group, fed
    // by that row position of every tile of the slice (rows 128*S*stride apart; the two lane halves are merged at the
    // end): 64 groups per slice whose members are spread over the whole shard, so a song's run of similar consecutive
    // rows lands in as many different groups instead of collapsing into one or two contiguous ones (which would push
    // the k-th best group maximum, i.e. the threshold, far down on a db of real music).
    f32x16 gmx[GMAX ? TM : 1][GMAX ? TN : 1];
#pragma unroll
    for (int i = 0; i < (GMAX ? TM : 1); ++i)
#pragma unroll
        for (int j = 0; j < (GMAX ? TN : 1); ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) gmx[i][j][r] = -INFINITY;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    // ---- db-tile staging straight into LDS (global_load_lds_dwordx4: no staging registers, no
    // ds_write pass).  The LDS image is lane-linear (wave-uniform base + lane*16), so rows cannot be
    // padded; instead 16-byte chunk c of row r is FETCHED by the lane whose slot is c ^ key(r) and
    // the fragment reads apply the same XOR: 16 consecutive rows hit 16 different bank groups.
    //   instruction (wave, u) covers LDS chunks [(wave*NLD + u)*64, +64)
    constexpr int RP = 256 / ROWB;                 // rows per 256 bytes of LDS (1 at d = 128, 2 at d = 64)
    // BAL: a 16-lane read group holds fragment rows (l31 >> 2) in {0,3,5,6} or {1,2,4,7} x (l31 & 3) = 0..3, i.e. tile rows
    // 8 a + 4 wn + b: the key is built from a mod 4 and b so that the sixteen hit sixteen different bank groups again
    auto key = [](int r) { return BAL ? (((r >> 3) & 3) * (4 / RP)) | ((r & 3) / RP) : (r / RP) & (CPR - 1); };
    // the 64-row-tile instantiation is the full pass: launched with stride 1 only (launch_scan_f16), so the row step is a
    // compile-time 1 there and the sixteen (8 g + e) * stride products of the survivor path are immediates, not registers
    const int64_t rstride = DBR == 64 ? 1 : p.row_stride;
    const int64_t last_row = (p.nrows - 1) * rstride;
    // (the BUFFER form of the LDS load: one 32-bit offset per chunk against a per-tile descriptor, and -- unlike
    // global_load_lds, a FLAT instruction after which every wait becomes vmcnt(0) / lgkmcnt(0) -- it counts in order)
    unsigned goff[NLD];                            // byte offset of this lane's chunk inside a tile (stride folded in)
    int lrow[NLD];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int ci = (wave * NLD + u) * 64 + lane;
        const int r = ci / CPR, cs = ci % CPR;
        lrow[u] = r;
        goff[u] = (unsigned)((unsigned long long)r * (unsigned long long)rstride * ROWB) + (unsigned)((cs ^ key(r)) * 16);   // (launchers: qres_stride_ok)
    }
    auto load_tile = [&](int64_t t, float *Bd) {
        const int64_t r0 = t * DBR * rstride;
        const __amdgpu_buffer_rsrc_t srd_t = make_srd(dbb + r0 * ROWB, 0x7FFFFFF0ull);
        if ((t + 1) * DBR <= p.nrows) {      // whole tile in range (uniform)
#pragma unroll
            for (int u = 0; u < NLD; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_t, (__attribute__((address_space(3))) void *)&Bd[(wave_u * NLD + u) * 256], 16,
                                                         (int)goff[u], 0, 0, 0);
        } else {
            const unsigned last_off = (unsigned)((last_row - r0) * ROWB);      // (the last valid row lies in this tile)
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                // rows past the end: fetch the last row instead (their columns are masked by `nok`)
                const bool ok = t * DBR + lrow[u] < p.nrows;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_t, (__attribute__((address_space(3))) void *)&Bd[(wave_u * NLD + u) * 256], 16,
                                                         (int)(ok ? goff[u] : last_off + (goff[u] & (ROWB - 1))), 0, 0, 0);
            }
        }
    };
    if (t_lo < t_hi) load_tile(t_lo, Bs0);
    if (NBUF == 3 && t_lo + S < t_hi) {
        load_tile(t_lo + S, Bs1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" :: "n"(NLD) : "memory");      // tile t_lo has landed; t_lo + S may still travel
        __builtin_amdgcn_s_barrier();                           // (raw: see the end of tile_step)
        asm volatile("" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the barrier's fence does not wait for an LDS load)
        __syncthreads();
    }
    // one db tile: MFMAs on the tile in Bc while the next one (NBUF = 3: the one after the next) travels into Bn
#ifdef PFANN_SCAN_TRACE
    unsigned long long tr_sum[5] = {0, 0, 0, 0, 0}, tr_last = __builtin_readcyclecounter(), tr_steps = 0;
#endif
    auto tile_step = [&](int64_t t, const float *Bc, float *Bn) {
        const bool req = t + (NBUF - 1) * S < t_hi;
        if (req) load_tile(t + (NBUF - 1) * S, Bn);
        __builtin_amdgcn_sched_barrier(0);
        SCAN_STAMP(0);
        f32x16 acc[TM][TN];
        // db fragments one K step ahead of the MFMAs that use them: with four 32-cycle MFMAs per step a wave that reads
        // its fragments only after issuing the previous step's MFMAs waits out the whole LDS latency every step
        constexpr int FA = 1;                        // K steps the fragment reads run ahead (2 measured: +0.5 %, round 5)
        f16x8 b8[FA + 1][TN];
        auto frag = [&](int kk, int set) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // tile row of this lane's fragment row l31.  Full pass (BAL): groups of four tile rows go to the four owners
                // (lane half, wave) in turn, so that a run of similar consecutive db rows (one song: up to 40 survivors of a
                // query row in a 64-row tile) fills the four private survivor lists evenly -- with each wave on its own 32-row
                // half of the tile one owner took 16 of the 40 and its list overflowed at a quarter of the slice's capacity.
                const int r = BAL ? j * 64 + 8 * (l31 >> 2) + 4 * wn + (l31 & 3) : wn * WN + j * 32 + l31;
                b8[set][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(
                                                           &Bc[r * (ROWB / 4) + (((kk * 2 + lhalf) ^ key(r)) * 4)]));
            }
        };
#pragma unroll
        for (int kk = 0; kk < FA; ++kk) frag(kk, kk);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + FA < KS) frag(kk + FA, (kk + FA) % (FA + 1));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[kk % (FA + 1)][j], afr[i][kk], kk == 0 ? zero16 : acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        SCAN_STAMP(1);
        if (NBUF == 3) {
            // tile t + S (requested one step ago) must have landed before the barrier below; vmcnt is ONE in-order counter
            // of loads and stores, so the wait goes HERE, before this step's survivor stores: everything older than the
            // request of this step -- tile t + S and the previous step's stores -- is then complete
            if (req) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int nvalid = (int)(p.nrows - t * DBR < DBR ? p.nrows - t * DBR : DBR);
        if (nvalid < DBR) {              // last tile (uniform): db rows past the end never count
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((BAL ? j * 64 + 16 * (r >> 2) + 8 * lhalf + 4 * wn + (r & 3) : wn * WN + j * 32 + 8 * (r >> 2) + 4 * lhalf + (r & 3)) >= nvalid) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j][r] = -INFINITY;
                    }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // maxima of the four register quads (4 consecutive db rows each), then of the lane
                float mg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mg[g] = fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                const float mx = fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3]));
                if (GMAX) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) gmx[i][j][r] = fmaxf(gmx[i][j][r], acc[i][j][r]);
                } else if (__any(mx >= th[i])) {
                    // Survivors are rare (a 32x32 block holds one with probability ~0.3, almost never two in a lane), so
                    // the path is built from wave-uniform tests per register quad and stores whose offset is out of range
                    // for non-survivors (the buffer unit drops them): no per-register exec-mask regions -- the
                    // straightforward `if (survivor) { atomic; store }` per register compiled to 440 instructions per
                    // block and was a third of the pass.
                    // Each lane appends to a list of ITS OWN (round 6): a query row's survivors in this slice come from four
                    // owners -- the two waves of its row half x the two lane halves -- which used to share one list behind an
                    // LDS counter (count pass, ds_add_rtn round trip, and in front of it the compiler's vmcnt(0)).  Now the
                    // (row, slice) list is cut into four private quarters, the write position is a register, and the count
                    // is written once at the end of the kernel; the select kernels gather 4 S sub-lists per row.
                    const int ml = wm * WM + i * 32 + l31;
                    // (BAL: register 4 g + e of this lane is tile row j * 64 + 16 g + 8 lhalf + 4 wn + e)
                    const unsigned row0 = (unsigned)((t * DBR + j * 64 + 8 * lhalf + 4 * wn) * rstride);
                    int pos = posr[i];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (__any(mg[g] >= th[i])) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[i][j][4 * g + e];
                                const bool sv = v >= th[i];
                                const unsigned long long key = pack_key(v, row0 + (unsigned)((16 * g + e) * rstride));
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, key), srd_k,
                                                                      (sv && pos < subcap) ? (unsigned)(ml * CAP + pos) * 8u + own_off : BUF_OOB, 0, 0);
                                pos += sv ? 1 : 0;
                            }
                        }
                    }
                    posr[i] = pos;
                }
            }
        SCAN_STAMP(2);
#ifdef PFANN_SCAN_TRACE
        ++tr_steps;
        if (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SCAN_STAMP(3);
            __syncthreads();
            SCAN_STAMP(4);
        } else
#endif
        if (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile has landed (and this tile's survivor stores are out)
            __syncthreads();
        } else {
            // a RAW barrier: __syncthreads() carries a fence, and with an LDS load outstanding the fence is lowered to
            // `s_waitcnt vmcnt(0)` -- it would drain the request of tile t + 2S issued at the top of this step, i.e. undo the
            // third buffer.  What the hand-over needs is above (this wave's share of tile t + S has landed: vmcnt(NLD)) and
            // here (its fragment reads are complete: lgkmcnt(0)).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    };
    if constexpr (NBUF == 3) {
        int64_t t = t_lo;
#pragma unroll 1
        while (t < t_hi) {
            tile_step(t, Bs0, Bs2);
            t += S;
            if (t >= t_hi) break;
            __builtin_amdgcn_sched_barrier(0);
            tile_step(t, Bs1, Bs0);
            t += S;
            if (t >= t_hi) break;
            __builtin_amdgcn_sched_barrier(0);
            tile_step(t, Bs2, Bs1);
            t += S;
        }
    } else {
        int64_t t = t_lo;
#pragma unroll 1
        while (t < t_hi) {
            tile_step(t, Bs0, Bs1);
            t += S;
            if (t >= t_hi) break;
            __builtin_amdgcn_sched_barrier(0);
            tile_step(t, Bs1, Bs0);
            t += S;
        }
    }
    if constexpr (GMAX) {
        // group maxima -> LDS [128 query rows][64 slots] (the two lane halves of a register are merged: 64 groups per
        // slice; XOR-swizzled by the row: lanes write 32 different rows at the same slot) -> one coalesced 256 B row per
        // query row: gmax[m][seg * 64 + slot]
        float *Gs = Bs0;
        static_assert(sizeof(Bs0) >= BM * 64 * sizeof(float), "group-maxima tile must fit its own LDS object");
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * WM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaxf(gmx[i][j][r], __shfl_xor(gmx[i][j][r], 32, 64));
                    const int slot = (wn * TN + j) * 16 + r;
                    if (lhalf == 0) Gs[ml * 64 + (slot ^ (ml & 31))] = v;
                }
        }
        __syncthreads();
        const int64_t G = (int64_t)S * 64;
        for (int it = 0; it < 8; ++it) {
            const int idx4 = it * 256 + tid, row = idx4 >> 4, c4 = idx4 & 15;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = Gs[row * 64 + ((4 * c4 + e) ^ (row & 31))];
            if (m0 + row < p.nq) *reinterpret_cast<f32x4 *>(&p.gmax[(m0 + row) * G + seg * 64 + 4 * c4]) = v;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + wm * WM + i * 32 + l31;
        if (m < p.nq) p.cnt[m * (4 * S) + own] = posr[i];
    }
#ifdef PFANN_SCAN_TRACE
    if (!GMAX && lane == 0 && g_scan_trace != nullptr && blockIdx.x < g_scan_trace_cap) {
        unsigned long long *o = g_scan_trace + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = tr_sum[i];
        o[5] = tr_steps; o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
}

#ifdef PFANN_SCAN_TRACE
extern "C" int pfann_debug_set_scan_trace(void *buf, unsigned cap_blocks) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_trace), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_trace_cap), &cap_blocks, sizeof(cap_blocks)) != hipSuccess) return -1;
    return 0;
}
#endif

// The query-stationary kernel addresses a db tile with 32-bit chunk offsets against a per-tile descriptor of 0x7FFFFFF0 bytes:
// the last row of a strided 128-row tile must lie inside it (stride <= ~66 k, i.e. shards below ~268 M rows for the ladder's
// largest stride); beyond that the launchers take the generic kernel / the survivor ladder.
static inline bool qres_stride_ok(int64_t stride, int d) {
    return 127ll * stride * (2ll * d) + 2ll * d <= 0x7FFFFFF0ll;
}

// Fewest query rows that take the query-stationary kernels (sampled group-maximum pass + full pass with sub-lists).  Up to
// round 5 this was 1024: between the streaming small-batch kernel (<= 32 rows) and 1024 rows the survivor ladder ran on the
// generic kernel with one to eight query tiles -- two sampled levels + full pass = 1.1-1.4 ms whatever the row count
// (profiles/r6/scan_mid_before.txt), the hole in the middle of the batch curve.  PFANN_QRES_MIN_NQ=1024 restores it (A/B).
static inline int64_t qres_min_nq() {
    static const int64_t v = getenv("PFANN_QRES_MIN_NQ") ? atoll(getenv("PFANN_QRES_MIN_NQ")) : 33;
    return v;
}

// Sampled group-maximum pass (every `stride`-th row): fills gmax[nq][*n_groups_out] for group_max_select.
// Returns 1 (not applicable: use the survivor ladder) when the shapes do not give >= 4 k groups per row.
int launch_scan_f16_gmax(const void *dbh, int64_t n, int d, int64_t stride, const void *qh, int64_t nq, int k,
                         SearchWorkspace &ws, int *n_groups_out, hipStream_t s) {
    ScanParams p;
    p.q = reinterpret_cast<const float *>(qh);
    p.db = reinterpret_cast<const float *>(dbh);
    p.nq = nq; p.d = d; p.row_stride = stride;
    p.nrows = (n + stride - 1) / stride;
    p.thr = nullptr; p.cnt = ws.cnt; p.keys = nullptr;
    p.n_tiles_m = cdiv(nq, 128);
    const int64_t db_tiles = cdiv(p.nrows, 128);
    if (!(d == 128 || d == 64) || nq < qres_min_nq() || db_tiles < 16 || !qres_stride_ok(stride, d)) return 1;
    // few query tiles (the middle of the batch curve): up to 64 slices, so that one tile still becomes 64 workgroups
    const int s_max = p.n_tiles_m * 32 < 512 ? 64 : 32;
    int S = (int)(2048 / p.n_tiles_m);
    S = S < 1 ? 1 : (S > s_max ? s_max : S);
    // no more groups than the threshold needs: 5 k of them (64 per slice) give the k-th best group maximum the same quality
    // as 1000-1600 did -- the full pass that follows is not a microsecond slower -- while the group select, whose cost is
    // the number of groups, halves: 0.77 -> 0.45 ms per 77,824 rows at 8 shards, 0.48 -> 0.29 on one GPU
    // (profiles/r4/sharded_scan_model.txt; PFANN_GMAX_S overrides)
    // ... but never fewer slices than fill the chip twice, and WHOLE rounds of the 512 resident workgroups: 76 query tiles x
    // 8 slices are 1.2 rounds (the sampled pass itself slows from 0.64 to 0.78 ms), x 14 are 2.08 rounds (0.72: the last
    // 0.08 of a round costs a round), x 13 are 1.93
    const int s_cap = std::max(std::max(8, (5 * k + 63) / 64), (int)(1024 / p.n_tiles_m));
    if (S > s_cap) S = s_cap;
    static const int gs_env = getenv("PFANN_GMAX_S") ? atoi(getenv("PFANN_GMAX_S")) : 0;      // tuning aid: slices of the sampled pass
    if (gs_env > 0) S = gs_env > 64 ? 64 : gs_env;       // (G = 64 S groups: the group select holds at most 4096 of them)
    if (S > db_tiles) S = (int)db_tiles;
    const int G = S * 64;                    // one group per (slice, row position in the 128-row tile up to the lane half)
    if (G < 4 * k || db_tiles < 4 * (int64_t)S) return 1;      // >= 4 rows per group
    p.nsub = S;
    p.gmax = reinterpret_cast<float *>(ws.cl);
    ProfScope ps("scan_topk_f16_sample", s, 2.0 * (double)nq * p.nrows * d);
    const dim3 grid((unsigned)(p.n_tiles_m * S));
    // at most one workgroup per CU: nobody covers a tile's round trip -> three tile buffers (96 KB, one workgroup per CU)
    if (d == 128 && p.n_tiles_m * S <= 256) PF_LAUNCH((scan_f16_qres_kernel<8, true, 128, 3>), grid, dim3(256), 0, s, p);
    else if (d == 128) PF_LAUNCH((scan_f16_qres_kernel<8, true>), grid, dim3(256), 0, s, p);
    else PF_LAUNCH((scan_f16_qres_kernel<4, true>), grid, dim3(256), 0, s, p);
    PF_HIP(hipGetLastError());
    *n_groups_out = G;
    return 0;
}

__global__ void fill_int2_kernel(int *p, int v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

int launch_scan_f16(const void *dbh, int64_t n, int d, int64_t stride, const void *qh, int64_t nq,
                    const float *thr_adj, SearchWorkspace &ws, bool allow_sublists, int *nsub_out, hipStream_t s) {
    ScanParams p;
    p.q = reinterpret_cast<const float *>(qh);
    p.db = reinterpret_cast<const float *>(dbh);
    p.nq = nq; p.d = d;
    p.row_stride = stride;
    p.nrows = (n + stride - 1) / stride;
    p.thr = thr_adj; p.cnt = ws.cnt; p.keys = reinterpret_cast<unsigned long long *>(ws.cl);
    p.nsub = 1;
    if (thr_adj == nullptr) {
        if (p.nrows > CAP) { set_error("scan: dense level with %lld rows > %d", (long long)p.nrows, CAP); return -1; }
        PF_LAUNCH(fill_int2_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, ws.cnt, (int)p.nrows, nq);
    } else {
        PF_HIP(hipMemsetAsync(ws.cnt, 0, sizeof(int) * nq, s));
    }
    ProfScope ps(stride == 1 ? "scan_topk_f16" : "scan_topk_f16_sample", s, 2.0 * (double)nq * p.nrows * d);
    p.n_tiles_m = cdiv(nq, 128);
    const int64_t db_tiles = cdiv(p.nrows, 128);
    static const bool no_qres = getenv("PFANN_NO_QRES") != nullptr;
    bool qres_lists = false;
    if (thr_adj != nullptr && allow_sublists && !no_qres && (d == 128 || d == 64) && nq >= qres_min_nq() && db_tiles >= 16 && qres_stride_ok(stride, d)) {
        qres_lists = true;
        // S interleaved db slices: about four rounds of the 512 resident workgroups, sub-lists of >= 256; with few query
        // tiles up to 64 slices (sub-lists of 128: a row's ~330 survivors spread over them), so that one query tile still
        // becomes 64 workgroups
        const int s_max = p.n_tiles_m * 32 < 768 ? 64 : 32;
        int S = (int)(2048 / p.n_tiles_m);
        S = S < 1 ? 1 : (S > s_max ? s_max : S);
        // whole rounds of the resident workgroups: with three per CU (64-row tiles) 76 query tiles x 26 slices are 2.57
        // rounds of 768; 30 slices (2.97 rounds) run the pass in 2.77 instead of 2.91 ms (20: 2.82, 32: 2.97)
        static const bool dbr64_s = getenv("PFANN_SCAN_DBR128") == nullptr;
        if (dbr64_s && d == 128 && stride == 1) {
            const int64_t slots = 768, rounds = (p.n_tiles_m * (int64_t)S + slots - 1) / slots;
            const int64_t s2 = rounds * slots / p.n_tiles_m;
            if (s2 >= S && s2 <= s_max) S = (int)s2;
        }
        static const int s_env = getenv("PFANN_SCAN_S") ? atoi(getenv("PFANN_SCAN_S")) : 0;
        if (s_env > 0) S = s_env;
        if (S > db_tiles) S = (int)db_tiles;
        p.nsub = S;
        // (no counter reset: every (row, private list) count is written by the lane that owns it, cnt[nq][4 S])
        const dim3 grid((unsigned)(p.n_tiles_m * S));
        // 64-row db tiles, three workgroups per CU (168 VGPRs) for the full pass: 2.97 -> 2.86 ms on the bench's 9728 x 1 M
        // pass, back to back on one box (four per CU would need <= 128 VGPRs: 35 spilled); PFANN_SCAN_DBR128=1: the old tiles
        static const bool dbr64 = getenv("PFANN_SCAN_DBR128") == nullptr;
        static const bool nbuf3_always = getenv("PFANN_SCAN_NBUF3") != nullptr;       // (A/B aid)
        if (d == 128 && dbr64 && stride == 1 && (p.n_tiles_m * S < 768 || nbuf3_always))      // less than one round of the resident slots
            PF_LAUNCH((scan_f16_qres_kernel<8, false, 64, 3>), grid, dim3(256), 0, s, p);
        else if (d == 128 && dbr64 && stride == 1) PF_LAUNCH((scan_f16_qres_kernel<8, false, 64>), grid, dim3(256), 0, s, p);
        else
        if (d == 128) PF_LAUNCH((scan_f16_qres_kernel<8>), grid, dim3(256), 0, s, p);
        else PF_LAUNCH((scan_f16_qres_kernel<4>), grid, dim3(256), 0, s, p);
    } else if (db_tiles * cdiv(p.n_tiles_m, 4) >= 4096)
        PF_LAUNCH((scan_f16_kernel<4>), dim3((unsigned)(db_tiles * cdiv(p.n_tiles_m, 4))), dim3(256), 0, s, p);
    else
        PF_LAUNCH((scan_f16_kernel<1>), dim3((unsigned)(db_tiles * p.n_tiles_m)), dim3(256), 0, s, p);
    PF_HIP(hipGetLastError());
    *nsub_out = qres_lists ? 4 * p.nsub : p.nsub;       // the query-stationary kernel: four private lists per (row, slice)
    return 0;
}

// ------------------------------------------------------------------------------------
// Select with exact re-scoring for rows with at most SMALL_N survivors (the normal case: ~1600):
// 256 threads and 32 KB of LDS per query row, so four rows are resident per CU and their global
// round trips (count / key gather / candidate rows) overlap.  Same steps and results as
// select_rescore_kernel below, which keeps the rows with more survivors.
// ------------------------------------------------------------------------------------
// Offsets of a row's sub-lists (at most NSUB_MAX = 256 since round 6: the query-stationary scan keeps FOUR private lists
// per (row, db slice) -- one per owner lane -- instead of one list with an LDS counter): s_off[g] = keys of sub-lists
// 0 .. g-1 (counts clamped to the sub-list capacity), s_off[nsub] = n, s_off[NSUB_MAX + 1] = some list overflowed.
// Called by every thread of the workgroup (NT >= 256 threads); contains barriers.
constexpr int NSUB_MAX = 256;
template <int NT>
__device__ __forceinline__ void sublist_offsets(const int *__restrict__ cnt_row, int nsub, int subcap, int *s_off, int *s_wt) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = 0;
    bool ov = false;
    if (tid < NSUB_MAX) {
        int c = tid < nsub ? cnt_row[tid] : 0;
        ov = c > subcap;
        c = ov ? subcap : c;
        incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wt[wave] = incl;
        const bool any_ov = __any(ov);
        if (lane == 0) s_wt[4 + wave] = any_ov ? 1 : 0;
    }
    __syncthreads();
    if (tid < NSUB_MAX) {
        int base = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) base += w < wave ? s_wt[w] : 0;
        s_off[tid + 1] = base + incl;
        if (tid == 0) {
            s_off[0] = 0;
            s_off[NSUB_MAX + 1] = s_wt[4] | s_wt[5] | s_wt[6] | s_wt[7];
        }
    }
    __syncthreads();
}

constexpr int SMALL_N = 4096;
__device__ __forceinline__ void select_rescore_small_row(const int64_t m, const unsigned long long *__restrict__ keys,
                                                const int *__restrict__ cnt, int k, int mode,
                                                float *__restrict__ thr, float *__restrict__ thr_adj,
                                                const float *__restrict__ eps, float *__restrict__ D,
                                                int64_t *__restrict__ I, int64_t label_base,
                                                int *overflow, int *__restrict__ row_ovf,
                                                const float *__restrict__ q32,
                                                const float *__restrict__ db32, int d, int nsub,
                                                int rescore) {
    constexpr int NT = 256, KPT = SMALL_N / NT;
    __shared__ __attribute__((aligned(16))) unsigned long long skeys[SMALL_N];
    __shared__ int s_n2, s_bin, s_kk;
    __shared__ int s_off[NSUB_MAX + 2], s_wt[8];
    __shared__ int hist[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int subcap = CAP / nsub;
    sublist_offsets<NT>(cnt + m * nsub, nsub, subcap, s_off, s_wt);
    const int n = s_off[nsub];
    if (n > SMALL_N) {                                // select_rescore_kernel's job
        if (tid == 0) atomicAdd(overflow + 1, 1);
        return;
    }
    const bool over = s_off[NSUB_MAX + 1] != 0;
    if (over && mode == 1 && tid == 0) row_ovf[m] = 1;           // topk_fallback_kernel recomputes this row
    if (nsub == 1) {
        for (int i = tid; i < n; i += NT) skeys[i] = keys[m * CAP + i];
    } else {
        const int l = tid & 7;                        // 8 threads per sub-list, 32 sub-lists per pass
        for (int g = tid >> 3; g < nsub; g += NT / 8) {
            const int o = s_off[g], c = s_off[g + 1] - o;
            for (int i = l; i < c; i += 8) skeys[o + i] = keys[m * CAP + g * subcap + i];
        }
    }
    __syncthreads();
    const float e2 = rescore ? 2.0f * eps[m] : 0.f;
    if (n <= k) {
        if (tid == 0) s_n2 = n;
        __syncthreads();
    } else {
        unsigned prefix = 0;
        int kk = k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += NT) {
                const unsigned hi = (unsigned)(skeys[i] >> 32);
                if (pass == 0 || (hi >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(hi >> shift) & 255], 1);
            }
            __syncthreads();
            if (wave == 0) {
                const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
                const int sum4 = c0 + c1 + c2 + c3;
                int incl = sum4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                const int excl = incl - sum4;
                if (excl < kk && kk <= incl) {
                    int rem = kk - excl, bin = 4 * lane;
                    if (rem > c0) { rem -= c0; ++bin; if (rem > c1) { rem -= c1; ++bin; if (rem > c2) { rem -= c2; ++bin; } } }
                    s_bin = bin;
                    s_kk = rem;
                }
            }
            __syncthreads();
            prefix |= (unsigned)s_bin << shift;
            kk = s_kk;
        }
        const float cut = ord2f(~prefix) - e2;
        const unsigned cut_hi = ~f2ord(cut);
        // in-place compaction: every thread takes its keys into registers first
        unsigned long long mine[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) mine[j] = tid + NT * j < n ? skeys[tid + NT * j] : ~0ull;
        if (tid == 0) s_n2 = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (tid + NT * j < n && (unsigned)(mine[j] >> 32) <= cut_hi) skeys[atomicAdd(&s_n2, 1)] = mine[j];
        __syncthreads();
    }
    const int n2 = s_n2;
    // exact fp32 scores: 4 threads per candidate (fixed reduction order), 4 x 64 candidates in flight per pass.  A
    // candidate's four lanes sit in one wave, so its key is read before it is replaced in program order: no barrier
    const float *qv = q32 + m * d;
    if (rescore)
    for (int c0 = 0; c0 < n2; c0 += NT) {
        float part[4];
        unsigned row[4];
        const int sub = tid & 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * (NT / 4) + (tid >> 2);
            row[u] = c < n2 ? (unsigned)(skeys[c] & 0xFFFFFFFFull) : 0u;
            part[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * (NT / 4) + (tid >> 2);
            if (c < n2) {
                const float *xv = db32 + (int64_t)row[u] * d;
                for (int e = sub * 4; e < d; e += 16) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(xv + e);
                    const float4 q4 = *reinterpret_cast<const float4 *>(qv + e);
                    part[u] = fmaf(x4.x, q4.x, part[u]); part[u] = fmaf(x4.y, q4.y, part[u]);
                    part[u] = fmaf(x4.z, q4.z, part[u]); part[u] = fmaf(x4.w, q4.w, part[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * (NT / 4) + (tid >> 2);
            part[u] += __shfl_xor(part[u], 1, 64);
            part[u] += __shfl_xor(part[u], 2, 64);
            if (c < n2 && sub == 0) skeys[c] = pack_key(part[u], row[u]);
        }
    }
    __syncthreads();
    // rank sort (keys are unique: they contain the row)
    {
        unsigned long long mine[KPT];
        int rank[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = tid + NT * j;
            mine[j] = ~0ull; rank[j] = 0;
            if (i < n2) {
                mine[j] = skeys[i];
                for (int t = 0; t < n2; ++t) rank[j] += skeys[t] < mine[j] ? 1 : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (tid + NT * j < n2) skeys[rank[j]] = mine[j];
        __syncthreads();
    }
    if (mode == 0) {
        if (tid == 0) {
            const float t = n2 >= k ? ord2f(~(unsigned)(skeys[k - 1] >> 32)) : -INFINITY;
            thr[m] = t;
            // finite (the scan's accumulators start at -thr_adj): below every possible score
            thr_adj[m] = fmaxf(rescore ? t - eps[m] : t, -1000.f * eps[m]);
        }
    } else {
        for (int i = tid; i < k; i += NT) {
            if (i < n2) {
                const unsigned long long key = skeys[i];
                D[m * k + i] = ord2f(~(unsigned)(key >> 32));
                I[m * k + i] = (int64_t)(unsigned)(key & 0xFFFFFFFFu) + label_base;
            } else {
                D[m * k + i] = -3.4028234663852886e38f;
                I[m * k + i] = -1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void select_rescore_small_kernel(const unsigned long long *__restrict__ keys,
                                                                   const int *__restrict__ cnt, int k, int mode,
                                                                   float *__restrict__ thr, float *__restrict__ thr_adj,
                                                                   const float *__restrict__ eps, float *__restrict__ D,
                                                                   int64_t *__restrict__ I, int64_t label_base,
                                                                   int *overflow, int *__restrict__ row_ovf,
                                                                   const float *__restrict__ q32,
                                                                   const float *__restrict__ db32, int d, int nsub,
                                                                   int rescore) {
    select_rescore_small_row(blockIdx.x, keys, cnt, k, mode, thr, thr_adj, eps, D, I, label_base, overflow, row_ovf, q32, db32, d, nsub,
                             rescore);
}
// the rows select_rescore_wave_kernel left (left[0 .. overflow[2])): a small grid walks the list -- normally empty, and an
// empty launch of one workgroup per query row cost 13 us per 9728 rows
__global__ __launch_bounds__(256) void select_rescore_small_list_kernel(const unsigned long long *__restrict__ keys,
                                                                        const int *__restrict__ cnt, int k, int mode,
                                                                        float *__restrict__ thr, float *__restrict__ thr_adj,
                                                                        const float *__restrict__ eps, float *__restrict__ D,
                                                                        int64_t *__restrict__ I, int64_t label_base,
                                                                        int *overflow, int *__restrict__ row_ovf,
                                                                        const float *__restrict__ q32,
                                                                        const float *__restrict__ db32, int d, int nsub,
                                                                        int rescore, const int *__restrict__ left) {
    const int n_left = overflow[2];
    for (int it = blockIdx.x; it < n_left; it += gridDim.x) {
        select_rescore_small_row(left[it], keys, cnt, k, mode, thr, thr_adj, eps, D, I, label_base, overflow, row_ovf, q32, db32, d, nsub,
                                 rescore);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// The same select for rows with at most WAVE_N survivors, ONE WAVEFRONT per row, four rows per workgroup (round 5): the
// second phase of a sharded search -- every shard emits only rows that can reach the GLOBAL k-th bound, a few dozen per
// query row at 8 shards -- left select_rescore_small_kernel with 256 threads, 32 KB of LDS and a dozen workgroup
// barriers around ~30 keys: 0.70 of a rank's 4.8 ms of scan kernels per 77,824-row step, a cost per query row that does not
// shrink with the shard (tools/ubench/sharded_scan_model.py).  Same steps, same summation order of the exact scores (four
// lanes per candidate), same outputs; rows with more survivors or an overflowed sub-list are counted in overflow[2] and
// left to select_rescore_small_kernel / select_rescore_kernel / the fallback.  mode 1 only (D, I out).
// ------------------------------------------------------------------------------------
constexpr int WAVE_N = 256;
__global__ __launch_bounds__(256) void select_rescore_wave_kernel(const unsigned long long *__restrict__ keys,
                                                                  const int *__restrict__ cnt, int k,
                                                                  const float *__restrict__ eps, float *__restrict__ D,
                                                                  int64_t *__restrict__ I, int64_t label_base, int *overflow,
                                                                  const float *__restrict__ q32, const float *__restrict__ db32,
                                                                  int d, int nsub, int rescore, int64_t nq, int *__restrict__ left) {
    constexpr int KPL = WAVE_N / 64;
    __shared__ __attribute__((aligned(16))) unsigned long long wk[4][WAVE_N];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + w;
    if (m >= nq) return;
    unsigned long long *sk = wk[w];
    const int subcap = CAP / nsub;
    // up to 256 sub-lists, four per lane (sub-list lane + 64 j in chunk j); offsets by a wave scan per chunk + carry
    int c[4], og[4], n = 0;
    bool ov = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = lane + 64 * j < nsub ? cnt[m * nsub + lane + 64 * j] : 0;
        ov = ov || c[j] > subcap;
        int incl = c[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        og[j] = n + incl - c[j];
        n += __shfl(incl, 63, 64);
    }
    if (__any(ov) || n > WAVE_N) {
        if (lane == 0) left[atomicAdd(overflow + 2, 1)] = (int)m;
        return;
    }
    // gather: every lane copies its own (short: a handful of keys) sub-lists, sub-list g at keys[m*CAP + g*subcap]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (64 * j >= nsub) break;
        int cmax = c[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
        const unsigned long long *src = keys + m * CAP + (int64_t)(lane + 64 * j) * subcap;
        for (int i = 0; i < cmax; ++i)
            if (i < c[j]) sk[og[j] + i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float e2 = rescore ? 2.0f * eps[m] : 0.f;
    int n2 = n;
    if (n > k) {
        // k-th best approximate score: the k-th smallest high word, by MSB-first bisection (largest P with fewer than k
        // high words below it), then everything within 2 eps of it is a candidate
        unsigned hi[KPL];
        unsigned long long mine[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int i = lane + 64 * j;
            mine[j] = i < n ? sk[i] : ~0ull;
            hi[j] = (unsigned)(mine[j] >> 32);
        }
        unsigned prefix = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = prefix | (1u << bit);
            int cl = 0;
#pragma unroll
            for (int j = 0; j < KPL; ++j) cl += __popcll(__ballot(lane + 64 * j < n && hi[j] < cand));
            if (cl < k) prefix = cand;
        }
        const float cut = ord2f(~prefix) - e2;
        const unsigned cut_hi = ~f2ord(cut);
        int base = 0;                                       // (every key is in registers: the list may be overwritten)
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const bool take = lane + 64 * j < n && hi[j] <= cut_hi;
            const unsigned long long mask = __ballot(take);
            if (take) sk[base + __popcll(mask & ((1ull << lane) - 1ull))] = mine[j];
            base += __popcll(mask);
        }
        n2 = base;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // exact fp32 scores: 4 lanes per candidate, 16 candidates per pass, select_rescore_small_kernel's summation order
    const float *qv = q32 + m * d;
    if (rescore)
        for (int c0 = 0; c0 < n2; c0 += 16) {
            const int ci = c0 + (lane >> 2), sub = lane & 3;
            float part = 0.f;
            unsigned row = 0u;
            if (ci < n2) {
                row = (unsigned)(sk[ci] & 0xFFFFFFFFull);
                const float *xv = db32 + (int64_t)row * d;
                for (int e = sub * 4; e < d; e += 16) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(xv + e);
                    const float4 q4 = *reinterpret_cast<const float4 *>(qv + e);
                    part = fmaf(x4.x, q4.x, part); part = fmaf(x4.y, q4.y, part);
                    part = fmaf(x4.z, q4.z, part); part = fmaf(x4.w, q4.w, part);
                }
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            if (ci < n2 && sub == 0) sk[ci] = pack_key(part, row);
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // rank sort (keys are unique: they contain the row)
    {
        unsigned long long mine[KPL];
        int rank[KPL];
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int i = lane + 64 * j;
            mine[j] = ~0ull; rank[j] = 0;
            if (i < n2) {
                mine[j] = sk[i];
                for (int t = 0; t < n2; ++t) rank[j] += sk[t] < mine[j] ? 1 : 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int j = 0; j < KPL; ++j)
            if (lane + 64 * j < n2) sk[rank[j]] = mine[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (int i = lane; i < k; i += 64) {
        if (i < n2) {
            const unsigned long long key = sk[i];
            D[m * k + i] = ord2f(~(unsigned)(key >> 32));
            I[m * k + i] = (int64_t)(unsigned)(key & 0xFFFFFFFFu) + label_base;
        } else {
            D[m * k + i] = -3.4028234663852886e38f;
            I[m * k + i] = -1;
        }
    }
}

// ------------------------------------------------------------------------------------
// Select with exact re-scoring (step 4 above).  One 1024-thread workgroup per query row.
//   mode 0: thr[m] = exact k-th best (or -inf), thr_adj[m] = thr[m] - eps[m]
//   mode 1: D, I = exact top-k;  list overflow -> overflow flag (+ raised thresholds)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void select_rescore_body(const unsigned long long *__restrict__ keys,
                                           const int *__restrict__ cnt, int k, int mode,
                                           float *__restrict__ thr, float *__restrict__ thr_adj,
                                           const float *__restrict__ eps, float *__restrict__ D,
                                           int64_t *__restrict__ I, int64_t label_base,
                                           int *overflow, int *row_ovf,
                                           const float *__restrict__ q32,
                                           const float *__restrict__ db32, int d, int nsub,
                                           int skip_small, int rescore, int64_t m = -1) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    __shared__ int s_n2;
    __shared__ int s_off[NSUB_MAX + 2], s_wt[8];
    if (m < 0) m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (skip_small && overflow[1] == 0) return;       // every row was handled by select_rescore_small_kernel
    // gather the row's nsub sub-lists (sub-list g holds cnt[m*nsub+g] keys at keys[m*CAP + g*subcap])
    const int subcap = CAP / nsub;
    sublist_offsets<1024>(cnt + m * nsub, nsub, subcap, s_off, s_wt);
    const int n = s_off[nsub];
    if (skip_small && n <= SMALL_N) return;          // done by select_rescore_small_kernel
    const bool over = s_off[NSUB_MAX + 1] != 0;
    if (over && mode == 1 && tid == 0) row_ovf[m] = 1;           // topk_fallback_kernel recomputes this row
    int P = 1;
    while (P < n) P <<= 1;
    if (nsub == 1) {
        for (int i = tid; i < n; i += 1024) skeys[i] = keys[m * CAP + i];
    } else {
        const int l = tid & 15;                    // 16 threads per sub-list, 64 sub-lists per pass
        for (int g = tid >> 4; g < nsub; g += 64) {
            const int o = s_off[g], c = s_off[g + 1] - o;
            for (int i = l; i < c; i += 16) skeys[o + i] = keys[m * CAP + g * subcap + i];
        }
    }
    __syncthreads();
    // candidates that can still belong to the exact top-k: s16 >= (k-th best s16) - 2 eps
    const float e2 = rescore ? 2.0f * eps[m] : 0.f;
    unsigned long long *ck = skeys;               // the candidates end up in ck[0 .. n2)
    if (n <= k) {
        if (tid == 0) s_n2 = n;
    } else if (n <= CAP / 2) {
        // k-th smallest key by MSB radix select on the 32 score bits (4 passes of 8 bits, LDS
        // histogram), then compaction of everything above the cut: no sort of the ~16k survivors
        __shared__ int hist[256];
        __shared__ int s_bin, s_kk;
        unsigned prefix = 0;
        int kk = k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += 1024) {
                const unsigned hi = (unsigned)(skeys[i] >> 32);
                if (pass == 0 || (hi >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(hi >> shift) & 255], 1);
            }
            __syncthreads();
            if (wave == 0) {
                const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
                const int sum4 = c0 + c1 + c2 + c3;
                int incl = sum4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += v;
                }
                const int excl = incl - sum4;
                if (excl < kk && kk <= incl) {          // exactly one lane
                    int rem = kk - excl, bin = 4 * lane;
                    if (rem > c0) { rem -= c0; ++bin; if (rem > c1) { rem -= c1; ++bin; if (rem > c2) { rem -= c2; ++bin; } } }
                    s_bin = bin;
                    s_kk = rem;
                }
            }
            __syncthreads();
            prefix |= (unsigned)s_bin << shift;
            kk = s_kk;
        }
        const float cut = ord2f(~prefix) - e2;
        const unsigned cut_hi = ~f2ord(cut);         // score >= cut  <=>  key's high word <= cut_hi
        ck = skeys + CAP / 2;
        if (tid == 0) s_n2 = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = skeys[i];
            if ((unsigned)(key >> 32) <= cut_hi) ck[atomicAdd(&s_n2, 1)] = key;
        }
    } else {
        for (int i = n + tid; i < P; i += 1024) skeys[i] = ~0ull;
        __syncthreads();
        bitonic_sort_u64(skeys, P, tid, 1024);      // by approximate score, descending
        if (tid == 0) {
            const float cut = ord2f(~(unsigned)(skeys[k - 1] >> 32)) - e2;
            int lo = k, hi = n;                   // first index whose score < cut
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (ord2f(~(unsigned)(skeys[mid] >> 32)) >= cut) lo = mid + 1; else hi = mid;
            }
            s_n2 = lo;
        }
    }
    __syncthreads();
    const int n2 = s_n2;
    // exact fp32 scores, all candidates of a pass in flight together: 8 threads per candidate, each a
    // strided set of float4 chunks (a row is read as whole 128-byte lines), fixed reduction order
    const float *qv = q32 + m * d;
    if (rescore)
    for (int c0 = 0; c0 < n2; c0 += 128) {
        const int c = c0 + (tid >> 3), sub = tid & 7;
        float part = 0.f;
        unsigned row = 0;
        if (c < n2) {
            row = (unsigned)(ck[c] & 0xFFFFFFFFull);
            const float *xv = db32 + (int64_t)row * d;
            for (int e = sub * 4; e < d; e += 32) {
                const float4 x4 = *reinterpret_cast<const float4 *>(xv + e);
                const float4 q4 = *reinterpret_cast<const float4 *>(qv + e);
                part = fmaf(x4.x, q4.x, part); part = fmaf(x4.y, q4.y, part);
                part = fmaf(x4.z, q4.z, part); part = fmaf(x4.w, q4.w, part);
            }
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (c < n2 && sub == 0) ck[c] = pack_key(part, row);
    }
    __syncthreads();
    if (n2 <= 1024) {
        // rank sort (keys are unique: they contain the row): one pass of broadcast LDS reads, no
        // log^2 barrier ladder for ~100-150 candidates
        unsigned long long mine = ~0ull;
        int rank = 0;
        if (tid < n2) {
            mine = ck[tid];
            for (int j = 0; j < n2; ++j) rank += ck[j] < mine ? 1 : 0;
        }
        __syncthreads();
        if (tid < n2) ck[rank] = mine;
        __syncthreads();
    } else {
        int P2 = 1;
        while (P2 < n2) P2 <<= 1;
        for (int i = n2 + tid; i < P2; i += 1024) ck[i] = ~0ull;     // entries beyond n2 can no longer matter
        __syncthreads();
        bitonic_sort_u64(ck, P2, tid, 1024);         // by exact score, descending
    }
    if (mode == 0) {
        if (tid == 0) {
            const float t = n2 >= k ? ord2f(~(unsigned)(ck[k - 1] >> 32)) : -INFINITY;
            thr[m] = t;
            thr_adj[m] = fmaxf(rescore ? t - eps[m] : t, -1000.f * eps[m]);
        }
    } else {
        for (int i = tid; i < k; i += 1024) {
            if (i < n2) {
                const unsigned long long key = ck[i];
                D[m * k + i] = ord2f(~(unsigned)(key >> 32));
                I[m * k + i] = (int64_t)(unsigned)(key & 0xFFFFFFFFu) + label_base;
            } else {
                D[m * k + i] = -3.4028234663852886e38f;
                I[m * k + i] = -1;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void select_rescore_kernel(const unsigned long long *__restrict__ keys,
                                                              const int *__restrict__ cnt, int k, int mode,
                                                              float *__restrict__ thr, float *__restrict__ thr_adj,
                                                              const float *__restrict__ eps, float *__restrict__ D,
                                                              int64_t *__restrict__ I, int64_t label_base,
                                                              int *overflow, int *__restrict__ row_ovf,
                                                              const float *__restrict__ q32,
                                                              const float *__restrict__ db32, int d, int nsub,
                                                              int skip_small, int rescore) {
    select_rescore_body(keys, cnt, k, mode, thr, thr_adj, eps, D, I, label_base, overflow, row_ovf, q32, db32, d, nsub, skip_small, rescore);
}

__global__ __launch_bounds__(1024) void select_rescore_list_kernel(const unsigned long long *__restrict__ keys,
                                                                   const int *__restrict__ cnt, int k, int mode,
                                                                   float *__restrict__ thr, float *__restrict__ thr_adj,
                                                                   const float *__restrict__ eps, float *__restrict__ D,
                                                                   int64_t *__restrict__ I, int64_t label_base,
                                                                   int *overflow, int *__restrict__ row_ovf,
                                                                   const float *__restrict__ q32,
                                                                   const float *__restrict__ db32, int d, int nsub,
                                                                   int rescore, const int *__restrict__ left) {
    if (overflow[1] == 0) return;                     // no row of the list had more than SMALL_N survivors
    const int n_left = overflow[2];
    for (int it = blockIdx.x; it < n_left; it += gridDim.x) {
        select_rescore_body(keys, cnt, k, mode, thr, thr_adj, eps, D, I, label_base, overflow, row_ovf, q32, db32, d, nsub, 1, rescore, left[it]);
        __syncthreads();
    }
}

// Last launch of the small-batch search (search.hip, search_small): the rows select_rescore_small_kernel left (more than
// SMALL_N survivors; normally none: returns after one 4-byte read) and, behind it, the exact fallback for flagged rows
// (normally none either) -- two launches' worth of "nothing to do" in one.  FB_ELT: element size of the fallback's rows.
template <int FB_ELT>
__global__ __launch_bounds__(1024) void select_tail_kernel(const unsigned long long *__restrict__ keys, const int *__restrict__ cnt, int k,
                                                           float *__restrict__ thr, float *__restrict__ thr_adj,
                                                           const float *__restrict__ eps, float *D, int64_t *I, int64_t label_base,
                                                           int *overflow, int *row_ovf, const float *__restrict__ q32,
                                                           const float *__restrict__ db32, int d, int nsub, int rescore,
                                                           const void *__restrict__ fb_rows, int64_t n) {
    select_rescore_body(keys, cnt, k, 1, thr, thr_adj, eps, D, I, label_base, overflow, row_ovf, q32, db32, d, nsub, 1, rescore);
    __syncthreads();
    topk_fallback_body<FB_ELT, 1024>(blockIdx.x, row_ovf, q32, fb_rows, n, d, k, D, I, label_base);
}

int launch_select_tail(SearchWorkspace &ws, int64_t nq, int k, float *D, int64_t *I, int64_t label_base, const float *q32,
                       const float *db32, const void *fb_rows, int fb_elt, int64_t n, int d, int nsub, int rescore, hipStream_t s) {
    if (d > 1024) { set_error("search_topk: d=%d > 1024", d); return -1; }
    const void *fn = fb_elt == 4 ? (const void *)select_tail_kernel<4> : (const void *)select_tail_kernel<2>;
    if (ensure_dyn_lds(fn, CAP * 8)) return -1;
    ProfScope ps("topk_select_tail", s);
    if (fb_elt == 4)
        PF_LAUNCH(select_tail_kernel<4>, dim3((unsigned)nq), dim3(1024), CAP * 8, s, reinterpret_cast<const unsigned long long *>(ws.cl),
                  ws.cnt, k, ws.thr, ws.thr_adj, ws.eps, D, I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, rescore, fb_rows, n);
    else
        PF_LAUNCH(select_tail_kernel<2>, dim3((unsigned)nq), dim3(1024), CAP * 8, s, reinterpret_cast<const unsigned long long *>(ws.cl),
                  ws.cnt, k, ws.thr, ws.thr_adj, ws.eps, D, I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, rescore, fb_rows, n);
    PF_HIP(hipGetLastError());
    return 0;
}

// small-batch path: ws.overflow[1] was zeroed by the group select; the rows this kernel leaves go to launch_select_tail
int launch_select_rescore_small(SearchWorkspace &ws, int64_t nq, int k, float *D, int64_t *I, int64_t label_base,
                                const float *q32, const float *db32, int d, int nsub, int rescore, hipStream_t s) {
    ProfScope ps(rescore ? "topk_select_rescore" : "topk_select_radix", s);
    PF_LAUNCH(select_rescore_small_kernel, dim3((unsigned)nq), dim3(256), 0, s,
              reinterpret_cast<const unsigned long long *>(ws.cl), ws.cnt, k, 1, ws.thr, ws.thr_adj, ws.eps, D,
              I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, rescore);
    PF_HIP(hipGetLastError());
    return 0;
}

// few_survivors: the caller expects a few dozen survivors per row (second phase of a sharded search): the wave-per-row
// tier runs first and the workgroup kernels only see the rows it left
int launch_select_rescore(SearchWorkspace &ws, int64_t nq, int k, int mode, float *D, int64_t *I, int64_t label_base,
                          const float *q32, const float *db32, int d, int nsub, int rescore, hipStream_t s, bool few_survivors) {
    if (ensure_dyn_lds((const void *)select_rescore_kernel, CAP * 8)) return -1;
    ProfScope ps(rescore ? "topk_select_rescore" : "topk_select_radix", s);
    PF_HIP(hipMemsetAsync(ws.overflow + 1, 0, 2 * sizeof(int), s));
    static const bool no_wave = getenv("PFANN_NO_WAVE_SELECT") != nullptr;       // A/B aid
    const bool wave_tier = few_survivors && mode == 1 && nsub <= NSUB_MAX && k <= WAVE_N && !no_wave;
    const unsigned long long *keys = reinterpret_cast<const unsigned long long *>(ws.cl);
    if (wave_tier) {
        if (ensure_dyn_lds((const void *)select_rescore_list_kernel, CAP * 8)) return -1;
        PF_LAUNCH(select_rescore_wave_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, s, keys, ws.cnt, k, ws.eps, D, I, label_base,
                  ws.overflow, q32, db32, d, nsub, rescore, nq, ws.left);
        PF_LAUNCH(select_rescore_small_list_kernel, dim3((unsigned)std::min<int64_t>(nq, 2048)), dim3(256), 0, s, keys, ws.cnt, k, mode,
                  ws.thr, ws.thr_adj, ws.eps, D, I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, rescore, ws.left);
        PF_LAUNCH(select_rescore_list_kernel, dim3((unsigned)std::min<int64_t>(nq, 512)), dim3(1024), CAP * 8, s, keys, ws.cnt, k, mode,
                  ws.thr, ws.thr_adj, ws.eps, D, I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, rescore, ws.left);
        PF_HIP(hipGetLastError());
        return 0;
    }
    PF_LAUNCH(select_rescore_small_kernel, dim3((unsigned)nq), dim3(256), 0, s, keys, ws.cnt, k, mode, ws.thr, ws.thr_adj, ws.eps, D,
              I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, rescore);
    PF_LAUNCH(select_rescore_kernel, dim3((unsigned)nq), dim3(1024), CAP * 8, s, keys, ws.cnt, k, mode, ws.thr, ws.thr_adj, ws.eps, D,
              I, label_base, ws.overflow, ws.row_ovf, q32, db32, d, nsub, 1, rescore);
    PF_HIP(hipGetLastError());
    return 0;
}

// pfann_prewarm: one empty launch per translation unit makes the runtime load this unit's code object now
__global__ void noop_search_f16_kernel() {}
int prewarm_search_f16() {
    hipLaunchKernelGGL(noop_search_f16_kernel, dim3(1), dim3(1), 0, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace pfann
