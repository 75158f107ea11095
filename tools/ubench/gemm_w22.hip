// Clone experiment (tuning aid, not product code; results are not checked): the K = 384 conv GEMM with FEWER MULTIPLIES.
// A 3-tap stride-2 convolution of two neighbouring outputs
//     y0 = W0 e0 + W1 o0 + W2 e1          y1 = W0 e1 + W1 o1 + W2 e2        (e = even, o = odd input positions)
// is a 2-tap stride-1 convolution of the even samples plus a 1-tap one of the odd samples; Winograd's F(2,2) does the
// even part with three products instead of four:
//     A0 = (e0 - e1) W0 + o0 W1       A1 = e1 (W0 + W2)       A2 = (e1 - e2) (-W2) + o1 W1       y0 = A0 + A1,  y1 = A1 + A2
// i.e. 5 C_in x C_out blocks per output pair instead of 6 (-16.7 % MFMA work) for 1.5x the accumulators.  The tile life is
// the product kernel's (gemm_tile.hip's `k<2, 15, true>`): 128 output rows (= 64 pairs) x 128 channels per workgroup of 8
// waves, LayerNorm transform on the way into LDS, double-buffered LDS with one barrier per sub-step, output rows staged
// through LDS, statistics.  A channel chunk of 32 takes four sub-steps: [o0; o1] x W1 (full 128-row step), (e0-e1) x W0,
// (e1-e2) x -W2, e1 x (W0+W2) (64-row half steps) -- 80 MFMAs per wave and chunk instead of 96, 4 barriers instead of 3.
// TFLOP/s are ALGORITHMIC (2 M N 384), so the fractions compare directly with gemm_tile's table.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_w22.hip -o tools/ubench/gemm_w22 && tools/ubench/gemm_w22
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline __amdgpu_buffer_rsrc_t srd(const void *p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000);
}
__device__ inline f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned vo, int so) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
}
__device__ inline void st4(__amdgpu_buffer_rsrc_t r, unsigned vo, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), r, vo, 0, 0);
}

__global__ void fill(float *p, size_t n, float scale, float bias) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = bias + scale * ((float)(h & 0xFFFFFF) / 8388608.f - 1.f);
    }
}

// MODE 0: the F(2,2) scheme.  MODE 1: the same kernel skeleton doing the plain 6-block convolution (3 full 128-row
// sub-steps per chunk, two accumulator blocks) -- the like-for-like baseline inside this file.
// WHATIF bits (MODE 0 only; results meaningless, timing only; 1 = fewer MFMAs, 2 = + transform arithmetic, 4 = + 8 live registers): an optimistic bound for Winograd F(4,2) on the even phase -- 9
// channel blocks per FOUR outputs instead of F(2,2)'s 10 -- inside F(2,2)'s own tile life: every second channel chunk
// leaves out the 16 MFMAs of its e1 x (W0+W2) half step (80 + 64 = 144 per two chunks = 72 per chunk = 9/10), the loader
// pays the extra transform arithmetic F(4,2) needs (5 transformed even samples from 5 inputs with coefficients 2, 3:
// ~10 more FMAs per 16 loaded bytes) and a fourth accumulator block is kept live (F(4,2): 7 blocks per quad = 56
// registers per 64 x 32 wave tile instead of 48).  What it does NOT pay: F(4,2)'s 9 sub-steps (and barriers) per chunk
// instead of 4, its wider epilogue (coefficients up to 8) and its larger rounding error.
template <int MODE, int EPI = 0, int WHATIF = 0>       // EPI 1: output rows stored straight from the accumulators (16 B per lane), no LDS staging
__global__ __launch_bounds__(512, 4) void kw(const float *__restrict__ x, const float *__restrict__ lw, const float *__restrict__ lb,
                                             const float *__restrict__ w, const float *__restrict__ stats, float *__restrict__ y,
                                             float *__restrict__ part, int rps_out, unsigned long long *ts, int hot) {
    constexpr int LDK = 36, BM = 128, BN = 128, C = 128;
    constexpr int NSUB = MODE == 0 ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem, *Bs = smem + 2 * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhalf = lane >> 5, col4 = tid & 7, rowq = tid >> 3;
    const unsigned long long t_start = __builtin_readcyclecounter();
    const int m0 = blockIdx.x * BM;
    const int b = m0 / rps_out, r0 = m0 % rps_out;
    const int rps_in = 2 * rps_out;
    const float mu = stats[2 * b], rs = stats[2 * b + 1];
    // hot != 0: every tile reads the activations of sample (b % hot) -- an L2-resident operand, to see how much of the
    // tile time is HBM latency of the activation loads
    const __amdgpu_buffer_rsrc_t sx = srd(x + (int64_t)(hot ? b % hot : b) * rps_in * C, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t slw = srd(lw, (uint64_t)rps_in * C * 4), slb = srd(lb, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t sw = srd(w, (uint64_t)BN * 4 * C * 4);
    auto off = [&](int inrow) { return (unsigned)inrow < (unsigned)rps_in ? (unsigned)((inrow * C + col4 * 4) * 4) : 0x80000000u; };
    // MODE 0: loader thread = pair rowq of the tile (input rows 4p-1 .. 4p+3 = e0 o0 e1 o1 e2)
    // MODE 1: loader thread = output rows rowq and rowq + 64 (input rows 2r-1+tap)
    unsigned vo[5], vb[2];
    if (MODE == 0) { const int base = 4 * (r0 / 2 + rowq) - 1; for (int j = 0; j < 5; ++j) vo[j] = off(base + j); }
    for (int i = 0; i < 2; ++i) vb[i] = (unsigned)(rowq + 64 * i) * (unsigned)(4 * C) * 4u + col4 * 16u;
    f32x4 px[2], pw[2], pb[2], rb[2], fe1;
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 4; ++e) { px[i][e] = 0.1f * e; pw[i][e] = 1.f; pb[i][e] = 0.2f; rb[i][e] = 0.01f; }
    for (int e = 0; e < 4; ++e) fe1[e] = 0.3f;
    auto fx = [&](f32x4 xv, f32x4 wv, f32x4 bv) {
        f32x4 v;
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf((xv[e] - mu) * rs, wv[e], bv[e]), 0.f);
        return v;
    };
    f32x4 carry;                          // WHATIF: the running neighbours an F(4,2) input transform combines
    for (int e = 0; e < 4; ++e) carry[e] = 0.25f * e;
    auto extra = [&](f32x4 v) {           // ~10 dependent-free FMAs per vector, the result feeds what is stored
        if (!(WHATIF & 2)) return v;
        f32x4 t = v;
        for (int e = 0; e < 4; ++e) {
            const float a = fmaf(2.f, v[e], -carry[e]), b2 = fmaf(-3.f, carry[e], v[e]), c = fmaf(2.f, carry[e], v[e]);
            const float d2 = fmaf(-2.f, v[e], carry[e]), g = fmaf(a, 1e-30f, b2), h2 = fmaf(c, 1e-30f, d2);
            t[e] = fmaf(g, 1e-30f, fmaf(h2, 1e-30f, v[e]));
            carry[e] = fmaf(v[e], 1e-30f, carry[e]);
        }
        return t;
    };
    auto issue = [&](int kind, int cc) {                      // global loads of sub-step (kind, channel chunk cc)
        if (MODE == 0) {
            if (kind == 0) {
                px[0] = ld4(sx, vo[1], cc * 4); pw[0] = ld4(slw, vo[1], cc * 4); pb[0] = ld4(slb, vo[1], cc * 4);
                px[1] = ld4(sx, vo[3], cc * 4); pw[1] = ld4(slw, vo[3], cc * 4); pb[1] = ld4(slb, vo[3], cc * 4);
            } else if (kind == 1) {
                px[0] = ld4(sx, vo[0], cc * 4); pw[0] = ld4(slw, vo[0], cc * 4); pb[0] = ld4(slb, vo[0], cc * 4);
                px[1] = ld4(sx, vo[2], cc * 4); pw[1] = ld4(slw, vo[2], cc * 4); pb[1] = ld4(slb, vo[2], cc * 4);
            } else if (kind == 2) {
                px[0] = ld4(sx, vo[4], cc * 4); pw[0] = ld4(slw, vo[4], cc * 4); pb[0] = ld4(slb, vo[4], cc * 4);
            }
        } else {
            for (int i = 0; i < 2; ++i) {
                const unsigned o = off(2 * (r0 + rowq + 64 * i) - 1 + kind);
                px[i] = ld4(sx, o, cc * 4); pw[i] = ld4(slw, o, cc * 4); pb[i] = ld4(slb, o, cc * 4);
            }
        }
        for (int i = 0; i < 2; ++i) rb[i] = ld4(sw, vb[i], (kind * C + cc) * 4);
    };
    auto stash = [&](int kind, float *Ad, float *Bd) {        // transform + LDS refill of that sub-step
        if (MODE == 0) {
            if (kind == 0) {
                *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = fx(px[0], pw[0], pb[0]);
                *reinterpret_cast<f32x4 *>(&Ad[(64 + rowq) * LDK + col4 * 4]) = fx(px[1], pw[1], pb[1]);
            } else if (kind == 1) {
                const f32x4 f0 = fx(px[0], pw[0], pb[0]);
                fe1 = fx(px[1], pw[1], pb[1]);
                *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = extra(f0 - fe1);
            } else if (kind == 2) {
                *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = extra(fe1 - fx(px[0], pw[0], pb[0]));
            } else {
                *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = fe1;
            }
        } else {
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(&Ad[(rowq + 64 * i) * LDK + col4 * 4]) = fx(px[i], pw[i], pb[i]);
        }
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(&Bd[(rowq + 64 * i) * LDK + col4 * 4]) = rb[i];
    };
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 acc_extra[2];                   // WHATIF: 8 more live accumulator registers (56 instead of 48)
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 4; ++r) acc_extra[i][r] = 0.f;
    constexpr int NST = NSUB * (C / 32);
    issue(0, 0);
    const unsigned long long t_p1 = __builtin_readcyclecounter();
    stash(0, As, Bs);
    const unsigned long long t_p2 = __builtin_readcyclecounter();
    __syncthreads();
    // one sub-step: MFMAs on LDS buffer PB while sub-step st+1 goes global -> registers -> buffer PB^1
    auto substep = [&](int st, auto kindc, auto pbc) {
        constexpr int KIND = decltype(kindc)::value, PB = decltype(pbc)::value;
        constexpr int NEXT = (KIND + 1) % NSUB;
        const float *Ac = As + PB * (BM * LDK), *Bc = Bs + PB * (BN * LDK);
        float *An = As + (PB ^ 1) * (BM * LDK), *Bn = Bs + (PB ^ 1) * (BN * LDK);
        const bool more = st + 1 < NST;
        const int cc_next = ((st + 1) / NSUB) * 32;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a0, a1;
            if (MODE == 0) {
                a0 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
                if (KIND == 0) a1 = *reinterpret_cast<const f32x4 *>(&Ac[(64 + wm * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            } else {
                a0 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + l31) * LDK + kk * 8 + lhalf * 4]);
                a1 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            }
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&Bc[(wn * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            if (kk == 0) { if (more) issue(NEXT, cc_next); __builtin_amdgcn_sched_barrier(0); }
            if (kk == 3) { __builtin_amdgcn_sched_barrier(0); if (more) stash(NEXT, An, Bn); }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (MODE == 0) {
                    if (KIND == 0) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[0], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a1[s], acc[2], 0, 0, 0);
                    } else if (KIND == 1) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[0], 0, 0, 0);
                    else if (KIND == 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[2], 0, 0, 0);
                    else if (!(WHATIF & 1) || ((st / NSUB) & 1) == 0) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[1], 0, 0, 0);
                    else if (WHATIF & 4) { acc_extra[s & 1][s >> 1] += b0[s] * a0[s]; }
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a1[s], acc[1], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    };
    const unsigned long long t_loop0 = __builtin_readcyclecounter();
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    if (MODE == 0) {
        for (int ch = 0; ch < C / 32; ++ch) {
            substep(ch * 4 + 0, I0{}, I0{}); substep(ch * 4 + 1, I1{}, I1{});
            substep(ch * 4 + 2, I2{}, I0{}); substep(ch * 4 + 3, I3{}, I1{});
        }
    } else {
        for (int ch = 0; ch < C / 32; ch += 2) {          // 3 sub-steps per chunk: buffer parity repeats every two chunks
            substep(ch * 3 + 0, I0{}, I0{}); substep(ch * 3 + 1, I1{}, I1{}); substep(ch * 3 + 2, I2{}, I0{});
            substep(ch * 3 + 3, I0{}, I1{}); substep(ch * 3 + 4, I1{}, I0{}); substep(ch * 3 + 5, I2{}, I1{});
        }
    }
    const unsigned long long t_loop1 = __builtin_readcyclecounter();
    // ---- epilogue: rows through LDS as whole 512 B rows + statistics (as gemm_tile's EPI = 2, STATS)
    const __amdgpu_buffer_rsrc_t sy = srd(y + (int64_t)m0 * BN, (uint64_t)BM * BN * 4);
    constexpr int LDC = 132;
    float *Cs = smem;
    float a1s[2] = {0.f, 0.f}, a2s[2] = {0.f, 0.f};
    for (int i = 0; i < 2; ++i) {
        // MODE 0: pair p = wm*32 + l31 -> output rows 2p (i = 0: A0 + A1) and 2p + 1 (i = 1: A1 + A2)
        const int row = MODE == 0 ? 2 * (wm * 32 + l31) + i : wm * 64 + i * 32 + l31;
        for (int g = 0; g < 4; ++g) {
            f32x4 z4;
            for (int e = 0; e < 4; ++e) {
                const float z = (MODE == 0 ? acc[i][4 * g + e] + acc[i + 1][4 * g + e] : acc[i][4 * g + e]) + 0.5f + ((WHATIF & 4) ? acc_extra[i][g & 3] * 1e-30f : 0.f);
                a1s[i] += z; a2s[i] = fmaf(z, z, a2s[i]); z4[e] = z;
            }
            if (EPI == 1) st4(sy, (unsigned)row * (BN * 4u) + (unsigned)(wn * 32 + 8 * g + 4 * lhalf) * 4u, z4);
            else *reinterpret_cast<f32x4 *>(&Cs[row * LDC + wn * 32 + 8 * g + 4 * lhalf]) = z4;
        }
    }
    if (EPI == 0) __syncthreads();
    const unsigned long long t_e1 = __builtin_readcyclecounter();
    if (EPI == 0)
    for (int it = 0; it < 8; ++it) {
        const int row = it * 16 + wave * 2 + lhalf;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&Cs[row * LDC + l31 * 4]);
        st4(sy, (unsigned)row * (BN * 4u) + l31 * 16u, v);
    }
    const unsigned long long t_e2 = __builtin_readcyclecounter();
    if (EPI == 0) __syncthreads();
    float *red = smem + 128 * 132;
    for (int i = 0; i < 2; ++i) {
        const int row = MODE == 0 ? 2 * (wm * 32 + l31) + i : wm * 64 + i * 32 + l31;
        float a1 = a1s[i] + __shfl_xor(a1s[i], 32, 64), a2 = a2s[i] + __shfl_xor(a2s[i], 32, 64);
        if (lhalf == 0) { red[row * 4 + wn] = a1; red[512 + row * 4 + wn] = a2; }
    }
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
    if (tid < 128) {
        for (int q = 0; q < 4; ++q) { t1 += red[tid * 4 + q]; t2 += red[512 + tid * 4 + q]; }
        for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (lane == 0) { part[(blockIdx.x * 2 + wave) * 2] = t1; part[(blockIdx.x * 2 + wave) * 2 + 1] = t2; }
    }
    if (ts != nullptr && tid == 0) {
        const unsigned long long t_e3 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long *o = ts + (size_t)blockIdx.x * 10;
        o[0] = t_start; o[1] = t_loop0; o[2] = t_loop1; o[3] = __builtin_readcyclecounter();
        o[4] = t_p1; o[5] = t_p2; o[6] = t_e1; o[7] = t_e2; o[8] = t_e3;
    }
}

int main(int argc, char **argv) {
    const int rps_out = 1024, B = argc > 1 ? atoi(argv[1]) : 2048, hot = argc > 2 ? atoi(argv[2]) : 0;
    const int64_t M = (int64_t)B * rps_out;
    const int ntiles = (int)(M / 128);
    float *x, *lw, *lb, *w, *stats, *y, *part;
    (void)hipMalloc(&x, M * 2 * 128 * 4); (void)hipMalloc(&y, M * 128 * 4);
    (void)hipMalloc(&lw, 2 * rps_out * 128 * 4); (void)hipMalloc(&lb, 2 * rps_out * 128 * 4);
    (void)hipMalloc(&w, 128 * 4 * 128 * 4); (void)hipMalloc(&stats, B * 8); (void)hipMalloc(&part, (size_t)ntiles * 128 + 64);
    fill<<<4096, 256>>>(x, (size_t)M * 2 * 128, 1.f, 0.f); fill<<<256, 256>>>(lw, (size_t)2 * rps_out * 128, 0.2f, 1.f);
    fill<<<256, 256>>>(lb, (size_t)2 * rps_out * 128, 0.5f, 0.3f); fill<<<64, 256>>>(w, (size_t)128 * 4 * 128, 0.05f, 0.f);
    fill<<<16, 256>>>(stats, (size_t)B * 2, 0.01f, 1.f);
    const size_t lds = (128 * 132 + 1024 + 64) * 4;
    const double flop = 2.0 * M * 128 * 384;                  // ALGORITHMIC: the convolution's 3 taps
    printf("M = %lld rows, N = 128, K = 384 (algorithmic), %d tiles, LDS %zu B\n", (long long)M, ntiles, lds);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 6; mode >= 0; --mode) {
            auto kp = mode == 0 ? kw<0> : (mode == 1 ? kw<1> : (mode == 2 ? kw<0, 1> : (mode == 3 ? kw<0, 0, 1> : (mode == 4 ? kw<0, 0, 3> :
                      (mode == 5 ? kw<0, 0, 5> : kw<0, 0, 7>)))));
            (void)hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(kp, dim3(ntiles), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, rps_out, nullptr, hot); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kp, dim3(ntiles), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, rps_out, nullptr, hot);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            printf("%-64s %8.3f ms  %6.1f TFLOP/s (algorithmic)  %.3f  %s\n",
                   mode == 0 ? "F(2,2): 5 blocks per output pair, 4 sub-steps per chunk" : (mode == 1 ? "plain: 6 blocks per output pair, 3 sub-steps per chunk" : (mode == 2 ? "F(2,2), rows stored straight from the accumulators (no LDS staging)" :
                    (mode == 3 ? "WHAT-IF F(4,2): 9/10 of F(2,2)'s MFMAs, nothing else" : (mode == 4 ? "WHAT-IF F(4,2): 9/10 MFMAs + transform arithmetic" :
                     (mode == 5 ? "WHAT-IF F(4,2): 9/10 MFMAs + 8 more live registers" : "WHAT-IF F(4,2): 9/10 MFMAs + transform + 8 registers"))))),
                   ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
        }
    // per-tile phase stamps (shader clock cycles; two workgroups resident per CU)
    unsigned long long *ts; (void)hipMalloc(&ts, (size_t)ntiles * 80);
    unsigned long long *h = (unsigned long long *)malloc((size_t)ntiles * 80);
    for (int mode = 1; mode >= 0; --mode) {
        auto kp = mode == 0 ? kw<0> : kw<1>;
        hipLaunchKernelGGL(kp, dim3(ntiles), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, rps_out, ts, hot); (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, ts, (size_t)ntiles * 80, hipMemcpyDeviceToHost);
        double a = 0, b2 = 0, c2 = 0, d[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
        for (int i = ntiles / 4; i < ntiles * 3 / 4; ++i) {
            const unsigned long long *o = h + 10 * (size_t)i;
            a += o[1] - o[0]; b2 += o[2] - o[1]; c2 += o[3] - o[2]; ++n;
            d[0] += o[4] - o[0]; d[1] += o[5] - o[4]; d[2] += o[1] - o[5]; d[3] += o[6] - o[2]; d[4] += o[7] - o[6]; d[5] += o[8] - o[7];
        }
        printf("%s per tile (cycles): prologue %.0f, loop %.0f, epilogue %.0f, total %.0f\n", mode == 0 ? "F(2,2)" : "plain ", a / n, b2 / n, c2 / n, (a + b2 + c2) / n);
        printf("   prologue: set-up + first loads issued %.0f, loads arrive + transform + LDS refill %.0f, barrier %.0f\n", d[0] / n, d[1] / n, d[2] / n);
        printf("   epilogue: C tile to LDS + barrier %.0f, row stores issued %.0f, barrier + statistics %.0f, wait for the stores %.0f\n",
               d[3] / n, d[4] / n, d[5] / n, c2 / n - (d[3] + d[4] + d[5]) / n);
    }
    return 0;
}
