// Stand-alone reproducer: an fp16-MFMA kernel on one stream perturbs FFT kernels (rocFFT under torch.fft, and any kernel
// with packed-fp32 VALU arithmetic around LDS traffic) running on ANOTHER stream of the same process -- see README.md.
// This file is the aggressor alone: the query-stationary inner-product scan of libpfann_amd.so (csrc/search_f16.hip:
// scan_f16_qres_kernel<8, false, 64>), with the device helpers it uses, and a C entry point that launches it on a stream.
// Nothing else of the library is needed.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans -shared scan_alone.hip -o libscan_alone.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
static constexpr int CAP = 8192;                 // survivor slots per query row
static constexpr unsigned BUF_OOB = 0x80000000u; // any offset >= num_records: loads return 0, stores are dropped

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ unsigned long long pack_key(float score, unsigned row) {
    return ((unsigned long long)(~f2ord(score)) << 32) | row;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *base, unsigned long long bytes) {
    const unsigned n = bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff = 0) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, soff, 0);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}
struct ScanParams {
    const float *q, *db;     // fp16 rows (typed float* for historical reasons): q [nq][d], db [nrows][d]
    int64_t nq, nrows;
    int64_t row_stride;      // 1
    int d;
    const float *thr;        // [nq] survivor thresholds
    int *cnt;                // [nq][nsub] survivors per sub-list
    unsigned long long *keys;  // [nq][CAP]
    int n_tiles_m;
    int nsub;
    float *gmax;
};

template <int KS, bool GMAX = false, int DBR = 128>
__global__ __launch_bounds__(256, DBR == 64 ? 3 : 2) void scan_f16_qres_kernel(ScanParams p) {
    constexpr int BM = 128, WM = 64, WN = DBR / 2, TM = 2, TN = WN / 32;
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
    __shared__ int s_cnt[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhalf = lane >> 5;
    const int S = p.nsub, subcap = CAP / S;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = L / p.n_tiles_m, mt = L - seg * p.n_tiles_m;     // neighbours share the db segment (L2)
    const int64_t m0 = (int64_t)mt * BM;
    // db tiles seg, seg + S, seg + 2S, ...: interleaved, so a run of similar rows (one song) is spread
    // over the sub-lists instead of overflowing one
    const int64_t t_lo = seg, t_hi = (p.nrows + DBR - 1) / DBR;
    if (tid < BM) s_cnt[tid] = 0;
    // this workgroup's sub-lists: row ml, slot pos -> keys[(m0 + ml) * CAP + seg * subcap + pos]
    const __amdgpu_buffer_rsrc_t srd_k = make_srd(p.keys + m0 * CAP + (int64_t)seg * subcap, (unsigned long long)BM * CAP * 8ull);
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
    auto key = [](int r) { return (r / RP) & (CPR - 1); };
    const int64_t last_row = (p.nrows - 1) * p.row_stride;
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
        goff[u] = (unsigned)((unsigned long long)r * (unsigned long long)p.row_stride * ROWB) + (unsigned)((cs ^ key(r)) * 16);   // (launchers: qres_stride_ok)
    }
    auto load_tile = [&](int64_t t, float *Bd) {
        const int64_t r0 = t * DBR * p.row_stride;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the barrier's fence does not wait for an LDS load)
    __syncthreads();
    // one db tile: MFMAs on the tile in Bc while the next one travels into Bn
    auto tile_step = [&](int64_t t, const float *Bc, float *Bn) {
        if (t + S < t_hi) load_tile(t + S, Bn);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[TM][TN];
        // db fragments one K step ahead of the MFMAs that use them: with four 32-cycle MFMAs per step a wave that reads
        // its fragments only after issuing the previous step's MFMAs waits out the whole LDS latency every step
        constexpr int FA = 1;                        // K steps the fragment reads run ahead (2 measured: +0.5 %, round 5)
        f16x8 b8[FA + 1][TN];
        auto frag = [&](int kk, int set) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WN + j * 32 + l31;
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
        const int nvalid = (int)(p.nrows - t * DBR < DBR ? p.nrows - t * DBR : DBR);
        if (nvalid < DBR) {              // last tile (uniform): db rows past the end never count
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (wn * WN + j * 32 + 8 * (r >> 2) + 4 * lhalf + (r & 3) >= nvalid) {
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
                    const int ml = wm * WM + i * 32 + l31;
                    bool anyg[4];
                    int c = 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        anyg[g] = __any(mg[g] >= th[i]);
                        if (anyg[g]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) c += acc[i][j][4 * g + e] >= th[i] ? 1 : 0;
                        }
                    }
                    int pos = 0;
                    if (c > 0) pos = atomicAdd(&s_cnt[ml], c);       // one reservation for all of the lane's survivors
                    const unsigned row0 = (unsigned)((t * DBR + wn * WN + j * 32 + 4 * lhalf) * p.row_stride);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (anyg[g]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[i][j][4 * g + e];
                                const bool sv = v >= th[i];
                                const unsigned long long key = pack_key(v, row0 + (unsigned)((8 * g + e) * p.row_stride));
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, key), srd_k,
                                                                      (sv && pos < subcap) ? (unsigned)(ml * CAP + pos) * 8u : BUF_OOB, 0, 0);
                                pos += sv ? 1 : 0;
                            }
                        }
                    }
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next tile has landed (and this tile's survivor stores are out)
        __syncthreads();
    };
    {
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
    if (tid < BM && m0 + tid < p.nq) p.cnt[(m0 + tid) * S + seg] = s_cnt[tid];
}

// q, db: fp16 rows on the device; thr [nq] floats; cnt [nq * S] ints (zeroed by the caller); keys [nq * 8192] u64.
// d = 128.  S = number of interleaved db slices (30 in the library for 9728 query rows).
extern "C" int scan_alone_launch(const void *db, int64_t nrows, const void *q, int64_t nq, const float *thr, int *cnt,
                                 unsigned long long *keys, int S, void *stream) {
    ScanParams p;
    p.q = reinterpret_cast<const float *>(q);
    p.db = reinterpret_cast<const float *>(db);
    p.nq = nq; p.nrows = nrows; p.row_stride = 1; p.d = 128;
    p.thr = thr; p.cnt = cnt; p.keys = keys; p.gmax = nullptr;
    p.n_tiles_m = (int)((nq + 127) / 128);
    p.nsub = S;
    hipLaunchKernelGGL((scan_f16_qres_kernel<8, false, 64>), dim3((unsigned)(p.n_tiles_m * S)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
