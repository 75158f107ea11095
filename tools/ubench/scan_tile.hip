// Clone experiment (tuning aid, not product code; results are not checked): the batched fp16 scan's full pass
// (search_f16.hip, scan_f16_qres_kernel<8, false, 64>: 9728 query rows x 1 M db rows x 128, query fragments in registers,
// db tiles streamed through LDS by direct-to-LDS loads, lane-local survivor test) with switches that take one ingredient
// out at a time, and with more query rows per workgroup, to see what the pass is made of.
//   TM   : 32-row query blocks per wave (2 = product: 128 query rows per workgroup; 4: 256)
//   DBR  : db rows per tile (64 = product)
//   WGS  : workgroups per CU asked for by __launch_bounds__
//   ABL  : 0 everything; 1 no MFMAs (fragment reads kept); 2 no epilogue; 3 no db loads (tiles stay what they are);
//          4 no fragment reads (MFMAs on stale registers)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/scan_tile.hip -o tools/ubench/scan_tile && tools/ubench/scan_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static constexpr int CAP = 8192;
static constexpr unsigned BUF_OOB = 0x80000000u;

__device__ inline __amdgpu_buffer_rsrc_t make_srd(const void *p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (unsigned)bytes, 0x00020000);
}
__device__ inline unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ inline unsigned long long pack_key(float s, unsigned row) { return ((unsigned long long)(~f2ord(s)) << 32) | row; }
__device__ inline int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ void fill_rows(_Float16 *p, int64_t n, unsigned seed) {       // ~unit-norm rows of 128
    const int64_t row = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float v[2], ss = 0.f;
    for (int e = 0; e < 2; ++e) {
        unsigned h = (unsigned)(row * 128 + lane * 2 + e) * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        unsigned g = h * 747796405u + 2891336453u; g ^= g >> 17;
        // sum of two uniforms: close enough to a bell for a scan-rate test
        v[e] = ((float)(h & 0xFFFF) + (float)(g & 0xFFFF)) / 65536.f - 1.f;
        ss += v[e] * v[e];
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = 1.f / sqrtf(ss);
    for (int e = 0; e < 2; ++e) p[row * 128 + lane * 2 + e] = (_Float16)(v[e] * inv);
}

struct P {
    const char *q, *db;
    int64_t nq, nrows;
    const float *thr;
    int *cnt;
    unsigned long long *keys;
    int n_tiles_m, nsub;
};

template <int TM, int DBR, int WGS, int ABL, bool SPLITB = false>
__global__ __launch_bounds__(256, WGS) void scan(P p) {
    constexpr int KS = 8, BM = 64 * TM, WM = 32 * TM, WN = DBR / 2, TN = WN / 32;
    constexpr int ROWB = 256, CPR = 16, NLD = DBR * CPR / 256;
    __shared__ __attribute__((aligned(1024))) float Bs0[SPLITB ? DBR * ROWB / 4 : 256];
    __shared__ __attribute__((aligned(1024))) float Bs1[SPLITB ? DBR * ROWB / 4 : 256];
    __shared__ __attribute__((aligned(1024))) float Bsd[SPLITB ? 1 : 2][SPLITB ? 256 : DBR * ROWB / 4];   // !SPLITB: ONE array, runtime buffer index (the product's form)
    int bi = 0;
    __shared__ int s_cnt[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhalf = lane >> 5;
    const int S = p.nsub, subcap = CAP / S;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = L / p.n_tiles_m, mt = L - seg * p.n_tiles_m;
    const int64_t m0 = (int64_t)mt * BM;
    const int64_t t_lo = seg, t_hi = p.nrows / DBR;
    for (int i = tid; i < BM; i += 256) s_cnt[i] = 0;
    const __amdgpu_buffer_rsrc_t srd_k = make_srd(p.keys + m0 * CAP + (int64_t)seg * subcap, (uint64_t)BM * CAP * 8ull);
    const __amdgpu_buffer_rsrc_t srd_q = make_srd(p.q + m0 * ROWB, (uint64_t)(p.nq - m0) * ROWB);
    f16x8 afr[TM][KS];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            afr[i][kk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_q, (unsigned)(wm * WM + i * 32 + l31) * ROWB + kk * 32 + lhalf * 16, 0, 0));
    float th[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + wm * WM + i * 32 + l31;
        th[i] = m < p.nq ? p.thr[m] : INFINITY;
    }
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    auto key = [](int r) { return r & (CPR - 1); };
    unsigned long long goff[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int ci = (wave * NLD + u) * 64 + lane;
        const int r = ci / CPR, cs = ci % CPR;
        goff[u] = (unsigned long long)r * ROWB + (unsigned)((cs ^ key(r)) * 16);
    }
    auto load_tile = [&](int64_t t, auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        const char *base = p.db + t * DBR * ROWB;
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + goff[u]),
                                             (__attribute__((address_space(3))) void *)(SPLITB ? (bb == 0 ? &Bs0[(wave * NLD + u) * 256] : &Bs1[(wave * NLD + u) * 256]) : &Bsd[bb ^ bi][(wave * NLD + u) * 256]), 16, 0, 0);
    };
    using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
    if (t_lo < t_hi) load_tile(t_lo, B0{});
    __syncthreads();
    f16x8 b8[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) b8[s][j] = afr[0][s];
    auto tile = [&](int64_t t, auto bc, auto bnc) {
        constexpr int b = decltype(bc)::value;
        const float *Bc = SPLITB ? (b == 0 ? Bs0 : Bs1) : &Bsd[b ^ bi][0];
        if (ABL != 3 && t + S < t_hi) load_tile(t + S, bnc);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[TM][TN];
        auto frag = [&](int kk, int set) {
            if (ABL == 4) return;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WN + j * 32 + l31;
                b8[set][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(&Bc[r * (ROWB / 4) + (((kk * 2 + lhalf) ^ key(r)) * 4)]));
            }
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) frag(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ABL == 1) {
                        if (kk == 0) acc[i][j] = zero16;
                        acc[i][j][kk] += (float)b8[kk & 1][j][0] * (float)afr[i][kk][0];     // keeps the fragment reads alive
                    } else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[kk & 1][j], afr[i][kk], kk == 0 ? zero16 : acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL == 2) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) mx = fmaxf(mx, acc[i][j][0] + acc[i][j][15]);
            if (mx > 1e30f) s_cnt[tid & (BM - 1)] = 1;
        } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float mg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mg[g] = fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                const float mx = fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3]));
                if (__any(mx >= th[i])) {
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
                    if (c > 0) pos = atomicAdd(&s_cnt[ml], c);
                    const unsigned row0 = (unsigned)(t * DBR + wn * WN + j * 32 + 4 * lhalf);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (anyg[g]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[i][j][4 * g + e];
                                const bool sv = v >= th[i];
                                const unsigned long long key = pack_key(v, row0 + (unsigned)(8 * g + e));
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, key), srd_k,
                                                                      (sv && pos < subcap) ? (unsigned)(ml * CAP + pos) * 8u : BUF_OOB, 0, 0);
                                pos += sv ? 1 : 0;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int64_t t = t_lo; t < t_hi; t += 2 * S) {
        if (SPLITB) {
            tile(t, B0{}, B1{});
            if (t + S < t_hi) tile(t + S, B1{}, B0{});
        } else {
            tile(t, B0{}, B1{});
            bi ^= 1;
            asm volatile("" : "+s"(bi));
            if (t + S < t_hi) { tile(t + S, B0{}, B1{}); bi ^= 1; asm volatile("" : "+s"(bi)); }
        }
    }
    for (int i = tid; i < BM; i += 256)
        if (m0 + i < p.nq) p.cnt[(m0 + i) * S + seg] = s_cnt[i];
}

template <int TM, int DBR, int WGS, int ABL, bool SPLITB = true>
static void run(const char *what, P p, int slots_per_cu) {
    p.n_tiles_m = (int)((p.nq + 64 * TM - 1) / (64 * TM));
    // slices: whole rounds of the resident workgroups, about 2048 workgroups, <= 32 sub-lists
    int S = 2048 / p.n_tiles_m; S = S < 1 ? 1 : (S > 32 ? 32 : S);
    const int64_t slots = 256 * slots_per_cu, rounds = (p.n_tiles_m * (int64_t)S + slots - 1) / slots;
    const int64_t s2 = rounds * slots / p.n_tiles_m;
    if (s2 >= S && s2 <= 32) S = (int)s2;
    p.nsub = S;
    const dim3 grid((unsigned)(p.n_tiles_m * S));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(p.cnt, 0, sizeof(int) * p.nq * 64);
    hipLaunchKernelGGL((scan<TM, DBR, WGS, ABL, SPLITB>), grid, dim3(256), 0, 0, p); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan<TM, DBR, WGS, ABL, SPLITB>), grid, dim3(256), 0, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    int *hc = (int *)malloc(sizeof(int) * p.nq * S);
    (void)hipMemcpy(hc, p.cnt, sizeof(int) * p.nq * S, hipMemcpyDeviceToHost);
    double sv = 0; for (int64_t i = 0; i < p.nq * S; ++i) sv += hc[i];
    free(hc);
    const double flop = 2.0 * p.nq * p.nrows * 128;
    printf("%-72s grid %5u S %2d  %7.3f ms  %6.0f TF = %.3f of 2500  survivors/row %.0f  %s\n", what, grid.x, S, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 2500.0, sv / p.nq, hipGetErrorString(hipGetLastError()));
}

// Second layout: a wave owns its query rows exclusively (QB blocks of 32) and covers ALL DBR rows of the db tile, so a
// query row's list position lives in a register (the two lane halves exchange their counts by one shuffle): no LDS atomic
// and, with the two tile buffers as two distinct LDS arrays, no wait for the tile in flight anywhere in the loop body.
template <int QB, int DBR, int WGS, int ABL>
__global__ __launch_bounds__(256, WGS) void scan2(P p) {
    constexpr int KS = 8, BM = 128 * QB, TN = DBR / 32;
    constexpr int ROWB = 256, CPR = 16, NLD = DBR * CPR / 256;
    __shared__ __attribute__((aligned(1024))) float Bs0[DBR * ROWB / 4];
    __shared__ __attribute__((aligned(1024))) float Bs1[DBR * ROWB / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhalf = lane >> 5;
    const int S = p.nsub, subcap = CAP / S;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = L / p.n_tiles_m, mt = L - seg * p.n_tiles_m;
    const int64_t m0 = (int64_t)mt * BM;
    const int64_t t_lo = seg, t_hi = p.nrows / DBR;
    const __amdgpu_buffer_rsrc_t srd_k = make_srd(p.keys + m0 * CAP + (int64_t)seg * subcap, (uint64_t)BM * CAP * 8ull);
    const __amdgpu_buffer_rsrc_t srd_q = make_srd(p.q + m0 * ROWB, (uint64_t)(p.nq - m0) * ROWB);
    f16x8 afr[QB][KS];
#pragma unroll
    for (int i = 0; i < QB; ++i)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            afr[i][kk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_q, (unsigned)(wave * 32 * QB + i * 32 + l31) * ROWB + kk * 32 + lhalf * 16, 0, 0));
    float th[QB];
    int cnt_reg[QB];
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int64_t m = m0 + wave * 32 * QB + i * 32 + l31;
        th[i] = m < p.nq ? p.thr[m] : INFINITY;
        cnt_reg[i] = 0;
    }
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    auto key = [](int r) { return r & (CPR - 1); };
    unsigned long long goff[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int ci = (wave * NLD + u) * 64 + lane;
        const int r = ci / CPR, cs = ci % CPR;
        goff[u] = (unsigned long long)r * ROWB + (unsigned)((cs ^ key(r)) * 16);
    }
    auto load_tile = [&](int64_t t, auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        const char *base = p.db + t * DBR * ROWB;
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + goff[u]),
                                             (__attribute__((address_space(3))) void *)(bb == 0 ? &Bs0[(wave * NLD + u) * 256] : &Bs1[(wave * NLD + u) * 256]), 16, 0, 0);
    };
    using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
    if (t_lo < t_hi) load_tile(t_lo, B0{});
    __syncthreads();
    auto tile = [&](int64_t t, auto bc, auto bnc) {
        constexpr int b = decltype(bc)::value;
        const float *Bc = b == 0 ? Bs0 : Bs1;
        if (ABL != 3 && t + S < t_hi) load_tile(t + S, bnc);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[QB][TN];
        f16x8 b8[2][TN];
        auto frag = [&](int kk, int set) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = j * 32 + l31;
                b8[set][j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(&Bc[r * (ROWB / 4) + (((kk * 2 + lhalf) ^ key(r)) * 4)]));
            }
        };
        frag(0, 0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) frag(kk + 1, (kk + 1) & 1);
#pragma unroll
            for (int i = 0; i < QB; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[kk & 1][j], afr[i][kk], kk == 0 ? zero16 : acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL == 2) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < QB; ++i) mx = fmaxf(mx, acc[i][j][0] + acc[i][j][15]);
            if (mx > 1e30f) cnt_reg[0] = 1;
        } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                float mg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    mg[g] = fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]), fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                const float mx = fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3]));
                if (__any(mx >= th[i])) {
                    const int ml = wave * 32 * QB + i * 32 + l31;
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
                    const int c_o = __shfl_xor(c, 32, 64);
                    int pos = cnt_reg[i] + (lhalf ? c_o : 0);
                    cnt_reg[i] += c + c_o;
                    const unsigned row0 = (unsigned)(t * DBR + j * 32 + 4 * lhalf);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (anyg[g]) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = acc[i][j][4 * g + e];
                                const bool sv = v >= th[i];
                                const unsigned long long key = pack_key(v, row0 + (unsigned)(8 * g + e));
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, key), srd_k,
                                                                      (sv && pos < subcap) ? (unsigned)(ml * CAP + pos) * 8u : BUF_OOB, 0, 0);
                                pos += sv ? 1 : 0;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int64_t t = t_lo; t < t_hi; t += 2 * S) {
        tile(t, B0{}, B1{});
        if (t + S < t_hi) tile(t + S, B1{}, B0{});
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int64_t m = m0 + wave * 32 * QB + i * 32 + l31;
        if (lhalf == 0 && m < p.nq) p.cnt[m * S + seg] = cnt_reg[i];
    }
}

template <int QB, int DBR, int WGS, int ABL>
static void run2(const char *what, P p, int slots_per_cu) {
    p.n_tiles_m = (int)((p.nq + 128 * QB - 1) / (128 * QB));
    int S = 2048 / p.n_tiles_m; S = S < 1 ? 1 : (S > 32 ? 32 : S);
    const int64_t slots = 256 * slots_per_cu, rounds = (p.n_tiles_m * (int64_t)S + slots - 1) / slots;
    const int64_t s2 = rounds * slots / p.n_tiles_m;
    if (s2 >= S && s2 <= 32) S = (int)s2;
    p.nsub = S;
    const dim3 grid((unsigned)(p.n_tiles_m * S));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(p.cnt, 0, sizeof(int) * p.nq * 64);
    hipLaunchKernelGGL((scan2<QB, DBR, WGS, ABL>), grid, dim3(256), 0, 0, p); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan2<QB, DBR, WGS, ABL>), grid, dim3(256), 0, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    int *hc = (int *)malloc(sizeof(int) * p.nq * S);
    (void)hipMemcpy(hc, p.cnt, sizeof(int) * p.nq * S, hipMemcpyDeviceToHost);
    double sv = 0; for (int64_t i = 0; i < p.nq * S; ++i) sv += hc[i];
    free(hc);
    const double flop = 2.0 * p.nq * p.nrows * 128;
    printf("%-72s grid %5u S %2d  %7.3f ms  %6.0f TF = %.3f of 2500  survivors/row %.0f  %s\n", what, grid.x, S, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 2500.0, sv / p.nq, hipGetErrorString(hipGetLastError()));
}

// Third variant (product shape: wave = 64 query rows x 32 db rows): TWO accumulator sets -- the survivor test of tile
// t - 1 is spread between the MFMAs of tile t (one register quad's maximum per K step), so a wave's matrix pipe does
// not idle through its own epilogue; fragments PF K-steps ahead.
template <int WGS, int PF, int ABL>
__global__ __launch_bounds__(256, WGS) void scan3(P p) {
    constexpr int KS = 8, TM = 2, DBR = 64, BM = 128, WM = 64, WN = 32;
    constexpr int ROWB = 256, CPR = 16, NLD = DBR * CPR / 256;
    __shared__ __attribute__((aligned(1024))) float Bs0[DBR * ROWB / 4];
    __shared__ __attribute__((aligned(1024))) float Bs1[DBR * ROWB / 4];
    __shared__ int s_cnt[BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lhalf = lane >> 5;
    const int S = p.nsub, subcap = CAP / S;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = L / p.n_tiles_m, mt = L - seg * p.n_tiles_m;
    const int64_t m0 = (int64_t)mt * BM;
    const int64_t t_lo = seg, t_hi = p.nrows / DBR;
    if (tid < BM) s_cnt[tid] = 0;
    const __amdgpu_buffer_rsrc_t srd_k = make_srd(p.keys + m0 * CAP + (int64_t)seg * subcap, (uint64_t)BM * CAP * 8ull);
    const __amdgpu_buffer_rsrc_t srd_q = make_srd(p.q + m0 * ROWB, (uint64_t)(p.nq - m0) * ROWB);
    f16x8 afr[TM][KS];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            afr[i][kk] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(srd_q, (unsigned)(wm * WM + i * 32 + l31) * ROWB + kk * 32 + lhalf * 16, 0, 0));
    float th[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + wm * WM + i * 32 + l31;
        th[i] = m < p.nq ? p.thr[m] : INFINITY;
    }
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    auto key = [](int r) { return r & (CPR - 1); };
    unsigned long long goff[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int ci = (wave * NLD + u) * 64 + lane;
        const int r = ci / CPR, cs = ci % CPR;
        goff[u] = (unsigned long long)r * ROWB + (unsigned)((cs ^ key(r)) * 16);
    }
    auto load_tile = [&](int64_t t, auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        const char *base = p.db + t * DBR * ROWB;
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + goff[u]),
                                             (__attribute__((address_space(3))) void *)(bb == 0 ? &Bs0[(wave * NLD + u) * 256] : &Bs1[(wave * NLD + u) * 256]), 16, 0, 0);
    };
    using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
    if (t_lo < t_hi) load_tile(t_lo, B0{});
    __syncthreads();
    f32x16 accA[TM], accB[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { accA[i] = zero16; accB[i] = zero16; }
    // survivors of a finished block: everything after the wave-uniform test
    auto emit = [&](const f32x16 &a, const float (&mg)[4], int i, int64_t t) {
        const int ml = wm * WM + i * 32 + l31;
        bool anyg[4];
        int c = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            anyg[g] = __any(mg[g] >= th[i]);
            if (anyg[g]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) c += a[4 * g + e] >= th[i] ? 1 : 0;
            }
        }
        int pos = 0;
        if (c > 0) pos = atomicAdd(&s_cnt[ml], c);
        const unsigned row0 = (unsigned)(t * DBR + wn * WN + 4 * lhalf);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (anyg[g]) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = a[4 * g + e];
                    const bool sv = v >= th[i];
                    const unsigned long long key = pack_key(v, row0 + (unsigned)(8 * g + e));
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, key), srd_k,
                                                          (sv && pos < subcap) ? (unsigned)(ml * CAP + pos) * 8u : BUF_OOB, 0, 0);
                    pos += sv ? 1 : 0;
                }
            }
        }
    };
    // MFMAs of tile t into `cur`, the quad maxima of the previous tile's `prev` between them, then prev's survivors
    auto tile = [&](int64_t t, auto bc, auto bnc, f32x16 (&cur)[TM], f32x16 (&prev)[TM], bool have_prev) {
        constexpr int b = decltype(bc)::value;
        const float *Bc = b == 0 ? Bs0 : Bs1;
        if (ABL != 3 && ABL < 5 && t + S < t_hi) load_tile(t + S, bnc);
        __builtin_amdgcn_sched_barrier(0);
        f16x8 b8[PF];
        auto frag = [&](int kk) {
            const int r = wn * WN + l31;
            if (ABL == 6) { b8[kk % PF] = afr[0][kk]; return; }
            b8[kk % PF] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4 *>(&Bc[r * (ROWB / 4) + (((kk * 2 + lhalf) ^ key(r)) * 4)]));
        };
#pragma unroll
        for (int kk = 0; kk < PF; ++kk) frag(kk);
        float mg[TM][4];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                cur[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b8[kk % PF], afr[i][kk], kk == 0 ? zero16 : cur[i], 0, 0, 0);
            if (kk + PF < KS) frag(kk + PF);
            if (ABL != 2 && ABL < 5) {
                const int i = kk >> 2, g = kk & 3;           // one register quad of the previous tile per K step
                mg[i][g] = fmaxf(fmaxf(prev[i][4 * g], prev[i][4 * g + 1]), fmaxf(prev[i][4 * g + 2], prev[i][4 * g + 3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ABL != 2 && ABL < 5 && have_prev) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float mx = fmaxf(fmaxf(mg[i][0], mg[i][1]), fmaxf(mg[i][2], mg[i][3]));
                if (__any(mx >= th[i])) emit(prev[i], mg[i], i, t - S);
            }
        }
        if ((ABL == 2 || ABL >= 5) && prev[0][0] + prev[1][15] > 1e30f) s_cnt[tid & (BM - 1)] = 1;      // keeps the MFMAs alive
        if (ABL != 5 && ABL != 6) __syncthreads();
    };
    int64_t t = t_lo;
    bool have = false;
#pragma unroll 1
    for (; t < t_hi; t += 2 * S) {
        tile(t, B0{}, B1{}, accA, accB, have);
        have = true;
        if (t + S < t_hi) tile(t + S, B1{}, B0{}, accB, accA, true);
    }
    // the last tile's survivors
    {
        const int64_t n_t = (t_hi - t_lo + S - 1) / S;               // tiles of this workgroup
        const int64_t t_last = t_lo + (n_t - 1) * S;
        if (ABL != 2 && ABL < 5 && n_t > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x16 &a = (n_t & 1) ? accA[i] : accB[i];
                float mg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) mg[g] = fmaxf(fmaxf(a[4 * g], a[4 * g + 1]), fmaxf(a[4 * g + 2], a[4 * g + 3]));
                const float mx = fmaxf(fmaxf(mg[0], mg[1]), fmaxf(mg[2], mg[3]));
                if (__any(mx >= th[i])) emit(a, mg, i, t_last);
            }
        }
    }
    __syncthreads();
    if (tid < BM && m0 + tid < p.nq) p.cnt[(m0 + tid) * S + seg] = s_cnt[tid];
}

template <int WGS, int PF, int ABL>
static void run3(const char *what, P p, int slots_per_cu) {
    p.n_tiles_m = (int)((p.nq + 127) / 128);
    int S = 2048 / p.n_tiles_m; S = S < 1 ? 1 : (S > 32 ? 32 : S);
    const int64_t slots = 256 * slots_per_cu, rounds = (p.n_tiles_m * (int64_t)S + slots - 1) / slots;
    const int64_t s2 = rounds * slots / p.n_tiles_m;
    if (s2 >= S && s2 <= 32) S = (int)s2;
    p.nsub = S;
    const dim3 grid((unsigned)(p.n_tiles_m * S));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemset(p.cnt, 0, sizeof(int) * p.nq * 64);
    hipLaunchKernelGGL((scan3<WGS, PF, ABL>), grid, dim3(256), 0, 0, p); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan3<WGS, PF, ABL>), grid, dim3(256), 0, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    int *hc = (int *)malloc(sizeof(int) * p.nq * S);
    (void)hipMemcpy(hc, p.cnt, sizeof(int) * p.nq * S, hipMemcpyDeviceToHost);
    double sv = 0; for (int64_t i = 0; i < p.nq * S; ++i) sv += hc[i];
    free(hc);
    const double flop = 2.0 * p.nq * p.nrows * 128;
    printf("%-72s grid %5u S %2d  %7.3f ms  %6.0f TF = %.3f of 2500  survivors/row %.0f  %s\n", what, grid.x, S, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 2500.0, sv / p.nq, hipGetErrorString(hipGetLastError()));
}

// The fp16 MFMA stream by itself: NACC independent 32x32x16 accumulator chains per wave, nothing else in the loop.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak(float *out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x ^ e)); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float t = 0.f;
    for (int j = 0; j < NACC; ++j) t += acc[j][0] + acc[j][15];
    if (t == 12345.f) out[threadIdx.x] = t;
}
template <int NACC>
static void run_peak(int wgs_per_cu) {
    float *out; (void)hipMalloc(&out, 4096);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_peak<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_peak<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * wgs_per_cu * 4 * (double)iters * NACC * 32768.0;
    printf("fp16 MFMA 32x32x16 alone: %d chains per wave, %d waves per SIMD: %7.3f ms  %6.0f TF = %.3f of 2500\n", NACC, wgs_per_cu, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
    (void)hipFree(out);
}

// ... and the fp32 MFMA stream (32x32x2, the conv GEMMs' instruction) for comparison
template <int NACC>
__global__ __launch_bounds__(256) void mfma32_peak(float *out, int iters) {
    float a = 0.001f * threadIdx.x, b = 0.002f * (threadIdx.x ^ 5);
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float t = 0.f;
    for (int j = 0; j < NACC; ++j) t += acc[j][0] + acc[j][15];
    if (t == 12345.f) out[threadIdx.x] = t;
}
template <int NACC>
static void run_peak32(int wgs_per_cu, int iters) {
    float *out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma32_peak<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((mfma32_peak<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * wgs_per_cu * 4 * (double)iters * NACC * 4096.0;
    printf("fp32 MFMA 32x32x2 alone: %d chains per wave, %d waves per SIMD: %7.3f ms  %6.1f TF = %.3f of 157.3\n", NACC, wgs_per_cu, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    (void)hipFree(out);
}

int main(int argc, char **argv) {
    const int64_t nq = 9728, n = 1000064;
    const float thr = argc > 1 ? (float)atof(argv[1]) : 0.27f;
    _Float16 *q, *db; float *th; int *cnt; unsigned long long *keys;
    (void)hipMalloc(&q, nq * 256); (void)hipMalloc(&db, n * 256); (void)hipMalloc(&th, nq * 4); (void)hipMalloc(&cnt, nq * 64 * 4);
    (void)hipMalloc(&keys, (size_t)(nq + 256) * CAP * 8);
    fill_rows<<<(unsigned)((nq + 3) / 4), 256>>>(q, nq, 17u);
    fill_rows<<<(unsigned)((n + 3) / 4), 256>>>(db, n, 99u);
    float *hth = (float *)malloc(nq * 4); for (int64_t i = 0; i < nq; ++i) hth[i] = thr;
    (void)hipMemcpy(th, hth, nq * 4, hipMemcpyHostToDevice);
    P p; p.q = (const char *)q; p.db = (const char *)db; p.nq = nq; p.nrows = n; p.thr = th; p.cnt = cnt; p.keys = keys;
    run_peak32<3>(2, 5000); run_peak32<3>(4, 5000); run_peak32<3>(4, 40000); run_peak32<2>(4, 40000);
    run_peak<1>(1); run_peak<2>(1); run_peak<4>(1); run_peak<2>(2); run_peak<2>(3); run_peak<4>(2); run_peak<4>(4);
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 64, 3, 0, false>("product form: ONE LDS array indexed [b] (s_waitcnt vmcnt(0) before the first fragment read)", p, 3);
        run<2, 64, 3, 2, false>("  ... no epilogue", p, 3);
        run<2, 64, 3, 0>("product shape: 128 q rows x 64 db rows, 3 wg/CU, two distinct LDS arrays", p, 3);
        run<2, 64, 3, 1>("  ... no MFMAs", p, 3);
        run<2, 64, 3, 2>("  ... no epilogue", p, 3);
        run<2, 64, 3, 3>("  ... no db loads", p, 3);
        run<2, 64, 3, 4>("  ... no fragment reads", p, 3);
        run<2, 128, 2, 0>("128 q rows x 128 db rows, 2 wg/CU", p, 2);
        run3<3, 2, 0>("two accumulator sets, test of tile t-1 between the MFMAs of t, 3 wg/CU, frag 2 ahead", p, 3);
        run3<3, 2, 2>("  ... no epilogue", p, 3);
        run3<3, 2, 7>("  ... no loads, no epilogue (barrier kept)", p, 3);
        run3<3, 2, 5>("  ... no loads, no epilogue, no barrier", p, 3);
        run3<3, 2, 6>("  ... MFMAs alone (no fragment reads either)", p, 3);
        run3<2, 8, 7>("  2 wg/CU, 8 fragments up front: no loads, no epilogue (barrier kept)", p, 2);
        run3<2, 8, 5>("  ... no barrier", p, 2);
        run3<2, 8, 6>("  ... MFMAs alone", p, 2);
        run3<2, 4, 0>("two accumulator sets, 2 wg/CU, frag 4 ahead", p, 2);
        run3<2, 8, 0>("two accumulator sets, 2 wg/CU, all 8 fragments up front", p, 2);
        run3<2, 8, 2>("  ... no epilogue", p, 2);
        run2<1, 64, 3, 0>("wave-owned rows, 128 q x 64 db, register counters, 3 wg/CU", p, 3);
        run2<1, 64, 3, 2>("  ... no epilogue", p, 3);
        run2<1, 64, 3, 3>("  ... no db loads", p, 3);
        run2<1, 64, 4, 0>("wave-owned rows, 128 q x 64 db, 4 wg/CU", p, 4);
        run2<1, 128, 2, 0>("wave-owned rows, 128 q x 128 db, 2 wg/CU", p, 2);
        run2<2, 64, 3, 0>("wave-owned rows, 256 q x 64 db, 3 wg/CU", p, 3);
        run2<2, 64, 2, 0>("wave-owned rows, 256 q x 64 db, 2 wg/CU", p, 2);
    }
    return 0;
}
