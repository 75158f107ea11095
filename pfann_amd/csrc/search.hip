// Exact brute-force inner-product top-k for gfx950 (reference database.py:121
// `index.search(query, top_k)` with an IndexFlatIP, i.e. the k largest q.x per query row).
//
// Design (MI355X-first, not faiss' GPU design):
//   * scores are produced by an LDS-tiled fp32-MFMA GEMM tile  q[BM] x db[BN]  (K = d);
//     every db row is streamed from HBM once per pass (query panels come from L2);
//   * NO score matrix is ever written.  Each score is compared in the MFMA epilogue with a
//     per-query threshold tau_q and only survivors are appended (one 8-byte packed key per
//     survivor, one atomic) to a small per-query list;
//   * tau_q is a provable lower bound of the query's k-th best score, obtained from the
//     exact top-k of a strided 1/R, 1/R^2 .. sample of the shard scanned first (the k-th
//     best of a subset can only be <= the k-th best of the whole).  Each level emits
//     ~R*k survivors per query, so the lists stay tiny and the final bitonic select in
//     LDS is exact;
//   * a list overflow on the final level (pathological ties / duplicates) raises tau from
//     the captured survivors and rescans; it never degrades to an approximate answer.
#include <algorithm>

#include "search_common.h"

namespace pfann {

// One workgroup owns one db tile (BN rows) and a run of QT consecutive query tiles, walked with a
// single software pipeline: tile (q, kt+1) -- or (q+1, 0) -- is prefetched while (q, kt) is on the
// matrix cores, so with K = d = 128 (4 K-tiles) the exposed first load and the pipeline drain are
// paid once per QT*4 K-tiles instead of once per 4.  The db tile is re-read from L1/L2 (64 KB).
template <int BM, int BN, int WM, int WN, int QT>
__global__ __launch_bounds__(256, 2) void scan_emit_kernel(ScanParams p) {
    constexpr int BK = 32, LDK = BK + 4;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(BM % 32 == 0 && BN % 32 == 0, "tiles are whole 32-row MFMA blocks");
    constexpr int AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BN * LDK];
    __shared__ __attribute__((aligned(16))) float thr_s[QT * BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int n_mg = (p.n_tiles_m + QT - 1) / QT;                // query-tile groups
    const int nt = L / n_mg, mg = L - nt * n_mg;                 // groups fastest: db tile stays in L2
    const int mt0 = mg * QT;
    const int n_q = min(QT, p.n_tiles_m - mt0);                  // query tiles this block really has
    const int64_t n0 = (int64_t)nt * BN;
    const int col4 = tid & 7, rowq = tid >> 3;

    for (int i = tid; i < QT * BM; i += 256) {
        const int64_t m = (int64_t)mt0 * BM + i;
        thr_s[i] = (p.thr != nullptr && m < p.nq) ? p.thr[m] : -INFINITY;
    }

    // bounds-checked buffer loads (OOB lanes read 0); windows start at this block's first rows
    const int64_t mq0 = (int64_t)mt0 * BM;
    const __amdgpu_buffer_rsrc_t srd_q = make_srd(p.q + mq0 * p.d, (unsigned long long)(p.nq - mq0) * p.d * 4ull);
    const int64_t row0 = n0 * p.row_stride;
    const int64_t rows_left = (p.nrows - n0 - 1) * p.row_stride + 1;
    const __amdgpu_buffer_rsrc_t srd_db = make_srd(p.db + row0 * p.d, (unsigned long long)rows_left * p.d * 4ull);
    unsigned doff[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int64_t rl = rowq + 32 * j;
        const unsigned long long off = (unsigned long long)rl * p.row_stride * p.d * 4ull;
        doff[j] = (n0 + rl < p.nrows && off < 0x7FFF0000ull) ? (unsigned)off : BUF_OOB;
    }
    f32x4 ra[AR], rb[BR];
    // load cursor of the NEXT tile to fetch, all in 32-bit byte offsets inside the two windows
    const int q_rows_left = (int)((p.nq - mq0) < (int64_t)QT * BM ? (p.nq - mq0) : (int64_t)QT * BM);
    const unsigned tile_bytes = (unsigned)BM * (unsigned)p.d * 4u;
    unsigned aoff[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) aoff[i] = (unsigned)(rowq + 32 * i) * (unsigned)p.d * 4u;
    int lkap = col4 * 4, lrow = rowq;        // lrow: row (inside the run) of this thread's first A row
    unsigned lbase = 0;                      // byte offset of the cursor's query tile
    auto load_tile = [&]() {
        const bool kok = lkap < p.d;
#pragma unroll
        for (int i = 0; i < AR; ++i)
            ra[i] = buf_load4(srd_q, (kok && lrow + 32 * i < q_rows_left) ? lbase + aoff[i] + (unsigned)lkap * 4u : BUF_OOB);
#pragma unroll
        for (int j = 0; j < BR; ++j) rb[j] = buf_load4(srd_db, kok ? doff[j] + (unsigned)lkap * 4u : BUF_OOB);
        lkap += BK;
        const bool wrap = lkap >= p.d;
        lkap = wrap ? col4 * 4 : lkap;
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

    const int nk = (p.d + BK - 1) / BK;
    load_tile();
    store_tile(As, Bs);
    __syncthreads();
    const int l31 = lane & 31, lhalf = lane >> 5;
    int it = 0;                               // running K-tile counter: LDS buffer = it & 1
#pragma unroll 1
    for (int q = 0; q < n_q; ++q) {
        const int64_t m0 = mq0 + (int64_t)q * BM;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++it) {
            const float *Ac = As + (it & 1) * (BM * LDK), *Bc = Bs + (it & 1) * (BN * LDK);
            float *An = As + ((it + 1) & 1) * (BM * LDK), *Bn = Bs + ((it + 1) & 1) * (BN * LDK);
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 a4[TM], b4[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a4[i] = *reinterpret_cast<const f32x4 *>(&Ac[(wm * WM + i * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b4[j] = *reinterpret_cast<const f32x4 *>(&Bc[(wn * WN + j * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
                if (kk == 0) {
                    load_tile();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kk == BK / 8 - 1) store_tile(An, Bn);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][s], b4[j][s], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        // epilogue: threshold filter + append.  acc[i][j][r]: query row = (r&3)+8*(r>>2)+4*lhalf,
        // db row = lane&31 of the 32x32 tile.  A wave-level branch costs ~50 cycles and survivors
        // are rare (~1.6e-3 per score), so survival is tested per GROUP of 4 slots (one branch),
        // and only groups with a survivor fall into the per-slot append code.
        const float *thr_c = thr_s + q * BM;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t n = n0 + wn * WN + j * 32 + l31;
            const bool nok = n < p.nrows;
            const unsigned row = (unsigned)(n * p.row_stride);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (p.thr == nullptr) {          // top sampling level: nrows <= CAP, no filter, no atomics
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                        if (nok && m < p.nq) p.keys[m * CAP + n] = pack_key(acc[i][j][r], row);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // rows (r&3)+8g+4*lhalf for r&3 = 0..3: four consecutive query rows
                        const int mlg = wm * WM + i * 32 + 8 * g + 4 * lhalf;
                        const f32x4 th = *reinterpret_cast<const f32x4 *>(thr_c + mlg);
                        bool sv[4];
                        bool any = false;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            sv[e] = nok && (m0 + mlg + e) < p.nq && acc[i][j][4 * g + e] >= th[e];
                            any |= sv[e];
                        }
                        if (__any(any)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (sv[e]) {
                                    const int64_t m = m0 + mlg + e;
                                    const int pos = atomicAdd(&p.cnt[m], 1);
                                    if (pos < CAP) p.keys[m * CAP + pos] = pack_key(acc[i][j][4 * g + e], row);
                                }
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
// Small-batch scan (nq <= 32, d = 128 or 64): the HBM-bound regime (one 10 s query = 19 rows;
// intensity Q/2 flop/byte).  Streaming design:
//   * the query block lives in registers as MFMA A-fragments for the whole kernel;
//   * every wave owns whole 32-row db tiles (16 KB contiguous at d = 128) and reads them with
//     fully contiguous 1 KB-per-instruction buffer loads (HBM-friendly bursts, each byte once);
//   * the tile is transposed to the MFMA B-fragment layout through a WAVE-PRIVATE padded LDS tile
//     (pitch d+4 dwords: conflict-free ds_write_b128 / ds_read_b128), so there are no workgroup
//     barriers at all; the next tile is already in flight in registers while the current one is
//     on the matrix cores;
//   * a persistent grid (2 workgroups per CU) walks the tiles: no per-tile pipeline fill.
// K order: MFMA step s of half h uses k = 8*(s>>2) + 4*h + (s&3)  (same map for both operands).
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void scan_small_kernel(ScanParams p) {
    constexpr int KS = D / 8;                 // float4 fragment pieces per lane
    constexpr int LD = D + 4;                 // LDS row pitch (dwords)
    constexpr int RPI = 256 / D;              // db rows covered by one 1 KB wave instruction (2 or 4)
    constexpr int NI = 32 / RPI;              // load instructions per 32-row tile (16 or 8)
    __shared__ __attribute__((aligned(16))) float tile_s[4 * 32 * LD];
    __shared__ __attribute__((aligned(16))) float thr_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhalf = lane >> 5;
    const int64_t n_tiles = (p.nrows + 31) / 32;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
    if (tid < 32) thr_s[tid] = tid < p.nq ? (p.thr != nullptr ? p.thr[tid] : -INFINITY) : INFINITY;
    __syncthreads();
    float *ts = tile_s + wave * 32 * LD;
    constexpr int NSUB = 32;
    const int sub = blockIdx.x & (NSUB - 1);

    f32x4 qa[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j)
        qa[j] = l31 < p.nq ? *reinterpret_cast<const f32x4 *>(p.q + (int64_t)l31 * p.d + 8 * j + 4 * lhalf)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t srd_db =
        make_srd(p.db, (unsigned long long)((p.nrows - 1) * p.row_stride + 1) * p.d * 4ull);
    const unsigned row_bytes = (unsigned)(p.row_stride * p.d * 4);
    // lane -> (row inside the instruction's RPI rows, float4 column)
    const int lrow = lane / (D / 4), lcol = lane % (D / 4);

    f32x4 st[NI];                              // next tile, in flight
    auto load_tile = [&](int64_t t) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int64_t n = t * 32 + j * RPI + lrow;
            const bool ok = t < n_tiles && n < p.nrows;
            st[j] = buf_load4(srd_db, ok ? (unsigned)n * row_bytes + (unsigned)lcol * 16u : BUF_OOB);
        }
    };
    load_tile(gw);
    for (int64_t t = gw; t < n_tiles; t += nw) {
        // registers -> wave-private LDS tile (previous tile's fragment reads are complete:
        // their results fed MFMAs already issued), then put the next tile in flight
#pragma unroll
        for (int j = 0; j < NI; ++j)
            *reinterpret_cast<f32x4 *>(&ts[(j * RPI + lrow) * LD + lcol * 4]) = st[j];
        load_tile(t + nw);                     // past the end: all lanes out of range, reads zeros
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const f32x4 xb = *reinterpret_cast<const f32x4 *>(&ts[l31 * LD + 8 * j + 4 * lhalf]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[j][e], xb[e], acc, 0, 0, 0);
        }
        // acc[r]: query row (r&3)+8*(r>>2)+4*lhalf, db row t*32 + l31
        const int64_t n = t * 32 + l31;
        const bool nok = n < p.nrows;
        const unsigned row = (unsigned)(n * p.row_stride);
        if (p.thr == nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                if (nok && m < p.nq) p.keys[(int64_t)m * CAP + n] = pack_key(acc[r], row);
            }
        } else {
            bool any = false;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 th = *reinterpret_cast<const f32x4 *>(thr_s + 8 * g + 4 * lhalf);
#pragma unroll
                for (int e = 0; e < 4; ++e) any |= acc[4 * g + e] >= th[e];
            }
            if (__any(any && nok)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    if (nok && acc[r] >= thr_s[m]) {
                        // same-address returning atomics serialise (~0.2 us each): with <= 32 query
                        // rows one counter per row would take ~1600 hits, so each row has NSUB
                        // sub-lists (CAP/NSUB slots each) picked by workgroup id
                        const int pos = atomicAdd(&p.cnt[m * NSUB + sub], 1);
                        if (pos < CAP / NSUB) p.keys[(int64_t)m * CAP + sub * (CAP / NSUB) + pos] = pack_key(acc[r], row);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Per-query exact select: bitonic sort of the (<= CAP) packed survivors in LDS.
//   mode 0: write thr[m] = k-th best score (or -inf when fewer than k survivors)
//   mode 1: write D[m][k], I[m][k] (+label_base); pad with -FLT_MAX / -1
// overflow[0] is set when a final-level list overflowed.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_kernel(const unsigned long long *__restrict__ keys,
                                                      const int *__restrict__ cnt, int k, int mode,
                                                      float *__restrict__ thr, float *__restrict__ D,
                                                      int64_t *__restrict__ I, int64_t label_base,
                                                      int *overflow, int nsub) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    __shared__ int s_off[130];
    const int64_t m = blockIdx.x;
    const int tid = threadIdx.x;
    const int subcap = CAP / nsub;
    bool over = false;
    if (tid == 0) {
        int run = 0;
        for (int g = 0; g < nsub; ++g) {
            int c = cnt[m * nsub + g];
            if (c > subcap) { c = subcap; over = true; }
            s_off[g] = run;
            run += c;
        }
        s_off[nsub] = run;
        s_off[129] = over ? 1 : 0;
        if (over && mode == 1) atomicExch(overflow, 1);
    }
    __syncthreads();
    const int n = s_off[nsub];
    over = s_off[129] != 0;
    int P = 1;
    while (P < n) P <<= 1;
    for (int g = 0; g < nsub; ++g) {
        const int o = s_off[g], c = s_off[g + 1] - o;
        for (int i = tid; i < c; i += 1024) skeys[o + i] = keys[m * CAP + g * subcap + i];
    }
    for (int i = n + tid; i < P; i += 1024) skeys[i] = ~0ull;
    __syncthreads();
    bitonic_sort_u64(skeys, P, tid, 1024);
    if (mode == 0) {
        if (tid == 0) thr[m] = n >= k ? ord2f(~(unsigned)(skeys[k - 1] >> 32)) : -INFINITY;
    } else {
        for (int i = tid; i < k; i += 1024) {
            if (i < n) {
                const unsigned long long key = skeys[i];
                D[m * k + i] = ord2f(~(unsigned)(key >> 32));
                I[m * k + i] = (int64_t)(unsigned)(key & 0xFFFFFFFFu) + label_base;
            } else {
                D[m * k + i] = -3.4028234663852886e38f;
                I[m * k + i] = -1;
            }
        }
        // on overflow, publish the raised threshold for the rescan
        if (over && tid == 0 && thr != nullptr && n >= k) thr[m] = ord2f(~(unsigned)(skeys[k - 1] >> 32));
    }
}

__global__ void fill_int_kernel(int *p, int v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static int launch_scan(const float *db, int64_t n, int d, int64_t stride, const float *q, int64_t nq,
                       const float *thr, SearchWorkspace &ws, int *nsub_out, hipStream_t s) {
    ScanParams p;
    p.q = q; p.db = db; p.nq = nq; p.d = d;
    p.row_stride = stride;
    p.nrows = (n + stride - 1) / stride;
    p.thr = thr; p.cnt = ws.cnt; p.keys = reinterpret_cast<unsigned long long *>(ws.cl);
    const bool small = nq <= 32 && (d == 128 || d == 64) && (uint64_t)p.nrows * stride * d * 4 < 0x7FFF0000ull;
    p.nsub = (small && thr != nullptr) ? 32 : 1;
    *nsub_out = p.nsub;
    if (thr == nullptr) {
        if (p.nrows > CAP) { set_error("scan: dense level with %lld rows > %d", (long long)p.nrows, CAP); return -1; }
        PF_LAUNCH(fill_int_kernel, dim3((unsigned)cdiv(nq, 256)), dim3(256), 0, s, ws.cnt, (int)p.nrows, nq);
    } else {
        PF_HIP(hipMemsetAsync(ws.cnt, 0, sizeof(int) * nq * p.nsub, s));
    }
    ProfScope ps(stride == 1 ? "scan_topk" : "scan_topk_sample", s, 2.0 * (double)nq * p.nrows * d);
    if (small) {
        // HBM-bound regime: persistent streaming kernel, 2 blocks per CU
        const int64_t tiles = (p.nrows + 31) / 32;
        const unsigned grid = (unsigned)std::min<int64_t>(512, (tiles + 3) / 4);
        if (d == 128) PF_LAUNCH((scan_small_kernel<128>), dim3(grid), dim3(256), 0, s, p);
        else PF_LAUNCH((scan_small_kernel<64>), dim3(grid), dim3(256), 0, s, p);
    } else if (nq <= 32) {
        p.n_tiles_m = 1;
        PF_LAUNCH((scan_emit_kernel<32, 128, 32, 32, 1>), dim3((unsigned)cdiv(p.nrows, 128)), dim3(256), 0, s, p);
    } else if (nq <= 64) {
        p.n_tiles_m = 1;
        PF_LAUNCH((scan_emit_kernel<64, 64, 32, 32, 1>), dim3((unsigned)cdiv(p.nrows, 64)), dim3(256), 0, s, p);
    } else {
        p.n_tiles_m = cdiv(nq, 128);
        const int64_t db_tiles = cdiv(p.nrows, 128);
        // long runs of query tiles per block only when the grid still fills the chip many times over
        if (db_tiles * cdiv(p.n_tiles_m, 4) >= 4096)
            PF_LAUNCH((scan_emit_kernel<128, 128, 64, 64, 4>), dim3((unsigned)(db_tiles * cdiv(p.n_tiles_m, 4))),
                               dim3(256), 0, s, p);
        else
            PF_LAUNCH((scan_emit_kernel<128, 128, 64, 64, 1>), dim3((unsigned)(db_tiles * p.n_tiles_m)),
                               dim3(256), 0, s, p);
    }
    PF_HIP(hipGetLastError());
    return 0;
}

static int launch_select(SearchWorkspace &ws, int64_t nq, int k, int mode, float *D, int64_t *I,
                         int64_t label_base, int nsub, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP(hipFuncSetAttribute((const void *)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   CAP * 8));
        attr_set = true;
    }
    ProfScope ps("topk_select", s);
    PF_LAUNCH(select_kernel, dim3((unsigned)nq), dim3(1024), CAP * 8, s,
                       reinterpret_cast<const unsigned long long *>(ws.cl), ws.cnt, k, mode, ws.thr, D, I,
                       label_base, ws.overflow, nsub);
    PF_HIP(hipGetLastError());
    return 0;
}

static int ensure_ws(SearchWorkspace &ws, int64_t nq) {
    if (ws.cap_q >= nq) return 0;
    if (ws.thr) { (void)hipFree(ws.thr); (void)hipFree(ws.cnt); (void)hipFree(ws.cl); }
    if (!ws.overflow) PF_HIP(hipMalloc(&ws.overflow, 4 * sizeof(int)));    // [0] overflow flag, [1] rows left to the big select kernel
    const int64_t cap = nq < 64 ? 64 : nq;
    if (ws.thr_adj) { (void)hipFree(ws.thr_adj); (void)hipFree(ws.eps); }
    PF_HIP(hipMalloc(&ws.thr_adj, sizeof(float) * cap));
    PF_HIP(hipMalloc(&ws.eps, sizeof(float) * cap));
    PF_HIP(hipMalloc(&ws.thr, sizeof(float) * cap));
    PF_HIP(hipMalloc(&ws.cnt, sizeof(int) * (cap < 32 ? 32 : cap) * 64));      // up to 64 sub-lists per row
    PF_HIP(hipMalloc(&ws.cl, sizeof(unsigned long long) * cap * CAP));
    ws.cap_q = cap;
    ws.cap_c = CAP;
    return 0;
}

__global__ void fill_empty_kernel(float *D, int64_t *I, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { D[i] = -3.4028234663852886e38f; I[i] = -1; }
}

int search_topk(const float *db, const void *dbh, float xnorm_max, int64_t n, int d, int64_t label_base,
                const float *q, int64_t nq, int k, float *D, int64_t *I, SearchWorkspace &ws, hipStream_t s) {
    if (nq <= 0) return 0;
    if (k < 1 || k > 1024) { set_error("search_topk: k=%d outside 1..1024", k); return -1; }
    if (d % 4 != 0) { set_error("search_topk: d=%d must be a multiple of 4", d); return -1; }
    if (n >= (1ll << 32)) { set_error("search_topk: shard rows %lld >= 2^32", (long long)n); return -1; }
    if (n == 0) {
        PF_LAUNCH(fill_empty_kernel, dim3((unsigned)cdiv(nq * k, 256)), dim3(256), 0, s, D, I, nq * k);
        PF_HIP(hipGetLastError());
        return 0;
    }
    if (ensure_ws(ws, nq)) return -1;
    // sampling ratio per level: expected survivors ~ R*k per query, kept <= CAP/4
    // (R = 8 and 4 were measured for the batched path too: more passes and selects cost more than the
    // shorter survivor lists save)
    const bool batched = dbh != nullptr && nq > 64;
    const int R = k <= 128 ? 16 : (k <= 512 ? 4 : 2);
    int levels = 0;
    int64_t stride = 1;
    // the coarsest level is scanned densely and fully sorted per query row: keep it <= 4096 rows (a
    // 1/8 shard of 1 M rows would otherwise sort 8192 keys per row); it still holds > 4096/R >= k rows
    const int64_t DENSE_CAP = 4096;
    while ((n + stride - 1) / stride > DENSE_CAP) { stride *= R; ++levels; }
    PF_HIP(hipMemsetAsync(ws.overflow, 0, sizeof(int), s));
    if (batched) {
        // ---- fp16 pre-filter + exact fp32 re-scoring (search_f16.hip): same levels, same exact result
        if (ws.qh_elems < nq * d) {
            if (ws.qh) (void)hipFree(ws.qh);
            ws.qh = nullptr; ws.qh_elems = 0;
            PF_HIP(hipMalloc(&ws.qh, (size_t)nq * d * 2));
            ws.qh_elems = nq * d;
        }
        if (launch_q_prep(q, nq, d, xnorm_max, ws.qh, ws.eps, s)) return -1;
        const float *ta = nullptr;
        for (int lev = levels; lev >= 1; --lev) {
            int nsub = 1;
            if (launch_scan_f16(dbh, n, d, stride, ws.qh, nq, ta, ws, true, &nsub, s)) return -1;
            if (launch_select_rescore(ws, nq, k, 0, nullptr, nullptr, 0, q, db, d, nsub, s)) return -1;
            ta = ws.thr_adj;
            stride /= R;
        }
        for (int attempt = 0; attempt < 4; ++attempt) {
            int nsub = 1;
            // a retry means some sub-list overflowed (e.g. hundreds of near-duplicate rows in one place):
            // fall back to one list of CAP entries per row
            if (launch_scan_f16(dbh, n, d, 1, ws.qh, nq, ta, ws, attempt == 0, &nsub, s)) return -1;
            if (launch_select_rescore(ws, nq, k, 1, D, I, label_base, q, db, d, nsub, s)) return -1;
            int ovf = 0;
            PF_HIP(hipMemcpyAsync(&ovf, ws.overflow, sizeof(int), hipMemcpyDeviceToHost, s));
            PF_HIP(hipStreamSynchronize(s));
            if (!ovf) return 0;
            if (ta == nullptr) { set_error("search_topk: survivor list overflow without threshold"); return -1; }
            PF_HIP(hipMemsetAsync(ws.overflow, 0, sizeof(int), s));
        }
        set_error("search_topk: survivor lists keep overflowing (more than %d rows tie at the k-th score?)", CAP);
        return -4;
    }
    PF_HIP(hipMemsetAsync(ws.overflow, 0, sizeof(int), s));
    const float *thr = nullptr;
    for (int lev = levels; lev >= 1; --lev) {
        int nsub = 1;
        if (launch_scan(db, n, d, stride, q, nq, thr, ws, &nsub, s)) return -1;
        if (launch_select(ws, nq, k, 0, nullptr, nullptr, 0, nsub, s)) return -1;
        thr = ws.thr;
        stride /= R;
    }
    for (int attempt = 0; attempt < 4; ++attempt) {
        int nsub = 1;
        if (launch_scan(db, n, d, 1, q, nq, thr, ws, &nsub, s)) return -1;
        if (launch_select(ws, nq, k, 1, D, I, label_base, nsub, s)) return -1;
        int ovf = 0;
        PF_HIP(hipMemcpyAsync(&ovf, ws.overflow, sizeof(int), hipMemcpyDeviceToHost, s));
        PF_HIP(hipStreamSynchronize(s));
        if (!ovf) return 0;
        // survivors overflowed a list: thresholds were raised by select_kernel; rescan
        if (thr == nullptr) { set_error("search_topk: survivor list overflow without threshold"); return -1; }
        PF_HIP(hipMemsetAsync(ws.overflow, 0, sizeof(int), s));
    }
    set_error("search_topk: survivor lists keep overflowing (more than %d rows tie at the k-th score?)", CAP);
    return -4;
}

// ------------------------------------------------------------------------------------
// Merge of per-shard top-k lists (after the RCCL all-gather): exact top-k of S/L[nq][m].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void merge_kernel(const float *__restrict__ S, const int64_t *__restrict__ Lb,
                                                     int m, int k, float *__restrict__ D,
                                                     int64_t *__restrict__ I) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const int64_t qi = blockIdx.x;
    const int tid = threadIdx.x;
    int P = 1;
    while (P < m) P <<= 1;
    for (int i = tid; i < P; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < m && Lb[qi * m + i] >= 0) key = pack_key(S[qi * m + i], (unsigned)i);
        skeys[i] = key;
    }
    __syncthreads();
    bitonic_sort_u64(skeys, P, tid, 1024);
    for (int i = tid; i < k; i += 1024) {
        const unsigned long long key = i < P ? skeys[i] : ~0ull;
        if (key != ~0ull) {
            D[qi * k + i] = ord2f(~(unsigned)(key >> 32));
            I[qi * k + i] = Lb[qi * m + (unsigned)(key & 0xFFFFFFFFu)];
        } else {
            D[qi * k + i] = -3.4028234663852886e38f;
            I[qi * k + i] = -1;
        }
    }
}

int topk_merge(const float *S, const int64_t *L, int64_t nq, int m, int k, float *D, int64_t *I,
               hipStream_t s) {
    if (nq <= 0) return 0;
    if (m > 16384) { set_error("topk_merge: m=%d > 16384", m); return -1; }
    int P = 1;
    while (P < m) P <<= 1;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP(hipFuncSetAttribute((const void *)merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   16384 * 8));
        attr_set = true;
    }
    ProfScope ps("topk_merge", s);
    PF_LAUNCH(merge_kernel, dim3((unsigned)nq), dim3(1024), (size_t)P * 8, s, S, L, m, k, D, I);
    PF_HIP(hipGetLastError());
    return 0;
}

}  // namespace pfann
