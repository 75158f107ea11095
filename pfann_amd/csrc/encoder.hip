// Fingerprint encoder kernels for gfx950 (CDNA4): the 8 separable-conv blocks and the
// grouped projection head of the reference's FpNetwork (model.py:14-153), re-designed as
//   * channels-last activations  x[b][f][t][c]  (K = taps x channels is contiguous),
//   * conv-as-implicit-GEMM on the exact-fp32 MFMA  v_mfma_f32_32x32x2_f32
//     (bitwise an fmaf chain, so embeddings stay inside the 1e-4 parity budget),
//   * LDS-staged 128x128x32 / 64x64x32 tiles read with ds_read_b128 (row pitch 36 dwords:
//     conflict-free for the b128 lane groups), register-prefetched global loads,
//   * XCD-aware block->tile mapping (blocks sharing an activation panel share an L2).
#include "kernels.h"

namespace pfann {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));



// ------------------------------------------------------------------------------------
// Implicit-GEMM convolution:  y[m][n] = bias[n] + sum_{tap,c} x[row(m) + tap][c] * w[n][tap][c]
// ------------------------------------------------------------------------------------
struct GemmConvParams {
    const float *x, *w, *bias;
    float *y;
    int64_t M;
    int N, K, Ci;
    int rows_per_sample, To, F, T;
    int axis, stride, pad_lo, in_len;
    int64_t tap_stride;
    int n_tiles_n;
    int k_begin, k_end;   // live part of K = taps x Ci (taps that only ever see padding are skipped)
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmConvParams p) {
    constexpr int BK = 32, LDK = BK + 4;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[2 * BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L / p.n_tiles_n, nt = L - mt * p.n_tiles_n;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;

    const int col4 = tid & 7, rowq = tid >> 3;
    // Tile operands come through bounds-checked buffer loads (OOB lanes read 0): the A window
    // starts at the first sample this tile touches, so 32-bit byte offsets suffice.
    const int64_t b_first = m0 / p.rows_per_sample;
    const int64_t in_elems = (int64_t)p.F * p.T * p.Ci;
    const int64_t total_in = (p.M / p.rows_per_sample) * in_elems;
    const __amdgpu_buffer_rsrc_t srd_a = make_srd(p.x + b_first * in_elems, (total_in - b_first * in_elems) * 4ull);
    const __amdgpu_buffer_rsrc_t srd_b = make_srd(p.w, (unsigned long long)p.N * p.K * 4ull);
    int aoff[AR], ap0[AR];      // element offset of tap 0 inside the window; coordinate of tap 0
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int64_t m = m0 + rowq + 32 * i;
        if (m < p.M) {
            const int64_t b = m / p.rows_per_sample;
            const int r = (int)(m - b * p.rows_per_sample);
            const int fo = r / p.To, to = r - fo * p.To;
            int rel;
            if (p.axis == 0) { ap0[i] = to * p.stride - p.pad_lo; rel = (fo * p.T + ap0[i]) * p.Ci; }
            else { ap0[i] = fo * p.stride - p.pad_lo; rel = (ap0[i] * p.T + to) * p.Ci; }
            aoff[i] = (int)((b - b_first) * in_elems) + rel;
        } else {
            ap0[i] = -(1 << 20);  // never valid
            aoff[i] = 0;
        }
    }
    unsigned boff[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int n = n0 + rowq + 32 * j;
        boff[j] = n < p.N ? (unsigned)n * (unsigned)p.K * 4u : BUF_OOB;
    }
    // running (tap, c) of this thread's float4 inside the K dimension
    int kap = p.k_begin + col4 * 4;
    int tap = kap / p.Ci, c = kap - tap * p.Ci;
    const int tap_stride = (int)p.tap_stride;

    f32x4 ra[AR], rb[BR];
    auto load_tile = [&]() {
        const bool kok = kap < p.k_end;
        const int toff = tap * tap_stride + c;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const bool ok = kok && (unsigned)(ap0[i] + tap) < (unsigned)p.in_len;
            ra[i] = buf_load4(srd_a, ok ? (unsigned)(aoff[i] + toff) * 4u : BUF_OOB);
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) rb[j] = buf_load4(srd_b, kok ? boff[j] + (unsigned)kap * 4u : BUF_OOB);
        kap += BK;
        c += BK;
#pragma unroll
        for (int q = 0; q < 2; ++q) { const bool w = c >= p.Ci; c -= w ? p.Ci : 0; tap += w ? 1 : 0; }
    };
    auto store_tile = [&](float *Ad, float *Bd) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            *reinterpret_cast<f32x4 *>(&Ad[(rowq + 32 * i) * LDK + col4 * 4]) = ra[i];
#pragma unroll
        for (int j = 0; j < BR; ++j)
            *reinterpret_cast<f32x4 *>(&Bd[(rowq + 32 * j) * LDK + col4 * 4]) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Main loop: LDS double buffer, ONE barrier per K-tile.  The global loads of tile kt+1 are
    // issued right after the first fragment reads of tile kt and their LDS stores (into the
    // other buffer) right before the last MFMA group, so both sit in MFMA shadows.
    const int nk = (p.k_end - p.k_begin + BK - 1) / BK;
    load_tile();
    store_tile(As, Bs);
    __syncthreads();
    const int l31 = lane & 31, lhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const float *Ac = As + (kt & 1) * (BM * LDK), *Bc = Bs + (kt & 1) * (BN * LDK);
        float *An = As + ((kt + 1) & 1) * (BM * LDK), *Bn = Bs + ((kt + 1) & 1) * (BN * LDK);
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a4[TM], b4[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a4[i] = *reinterpret_cast<const f32x4 *>(&Ac[(wm * WM + i * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b4[j] = *reinterpret_cast<const f32x4 *>(&Bc[(wn * WN + j * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            // unconditional (past the last tile every lane is out of range and reads zeros):
            // keeps the iteration one basic block so loads/stores interleave with the MFMAs
            if (kk == 0) {
                load_tile();
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (never sink it)
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
    // epilogue: + bias, bounds-checked stores (rows >= M and cols >= N fall outside the SRD).
    // C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const __amdgpu_buffer_rsrc_t srd_y = make_srd(p.y + m0 * p.N, (unsigned long long)(p.M - m0) * p.N * 4ull);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        const float bv = n < p.N ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                buf_store1(srd_y, n < p.N ? (unsigned)(ml * p.N + n) * 4u : BUF_OOB, acc[i][j][r] + bv);
            }
        }
    }
}

int launch_conv_gemm(const SubLayer &L, const float *x, float *y, int64_t B, hipStream_t s) {
    GemmConvParams p;
    p.x = x; p.w = L.w; p.bias = L.bias; p.y = y;
    p.rows_per_sample = L.Fo * L.To;
    p.M = B * p.rows_per_sample;
    p.N = L.co; p.Ci = L.ci; p.K = 3 * L.ci;
    p.To = L.To; p.F = L.F; p.T = L.T;
    p.axis = L.axis; p.stride = L.stride; p.pad_lo = L.pad_lo;
    p.in_len = L.axis == 0 ? L.T : L.F;
    p.tap_stride = L.axis == 0 ? (int64_t)L.ci : (int64_t)L.T * L.ci;
    if (L.ci % 4 != 0) { set_error("conv_gemm needs ci %% 4 == 0 (ci=%d)", L.ci); return -1; }
    // taps that fall into the zero padding for EVERY output position contribute nothing
    {
        const int out_len = L.axis == 0 ? L.To : L.Fo;
        int lo = 3, hi = -1;
        for (int tap = 0; tap < 3; ++tap)
            for (int o = 0; o < out_len; ++o) {
                const int pos = o * L.stride - L.pad_lo + tap;
                if (pos >= 0 && pos < p.in_len) { lo = tap < lo ? tap : lo; hi = tap > hi ? tap : hi; break; }
            }
        p.k_begin = lo * L.ci;
        p.k_end = (hi + 1) * L.ci;
    }
    const double flops = 2.0 * (double)p.M * p.N * (p.k_end - p.k_begin);
    // tile choice: big tiles when the grid still fills 256 CUs a few times over
    const int64_t blocks128 = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 128);
    if (p.N >= 128 && blocks128 >= 512) {
        p.n_tiles_n = cdiv(p.N, 128);
        ProfScope ps("conv_gemm_128", s, flops);
        PF_LAUNCH((conv_gemm_kernel<128, 128, 64, 64>), dim3((unsigned)blocks128), dim3(256),
                           0, s, p);
    } else {
        p.n_tiles_n = cdiv(p.N, 64);
        const int64_t blocks = (int64_t)cdiv(p.M, 64) * p.n_tiles_n;
        ProfScope ps("conv_gemm_64", s, flops);
        PF_LAUNCH((conv_gemm_kernel<64, 64, 32, 32>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    }
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// First conv (ci == 1): x[b][f][t] -> y[b][f][to][co]; pure bandwidth (2 MiB written/sample).
// w is stored [3][co].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ x,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ bias,
                                                         float *__restrict__ y, int64_t M, int co,
                                                         int To, int T, int stride, int pad_lo) {
    const int co4 = co >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * co4) return;
    const int64_t m = gid / co4;
    const int c = (int)(gid - m * co4) * 4;
    const int64_t bf = m / To;           // (b*F + f)
    const int to = (int)(m - bf * To);
    const int p0 = to * stride - pad_lo;
    f32x4 o = *reinterpret_cast<const f32x4 *>(bias + c);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        const int t = p0 + tap;
        if ((unsigned)t < (unsigned)T) {
            const float xv = x[bf * T + t];
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + tap * co + c);
            o += xv * wv;
        }
    }
    *reinterpret_cast<f32x4 *>(y + m * co + c) = o;
}

int launch_conv_first(const SubLayer &L, const float *x, float *y, int64_t B, hipStream_t s) {
    const int64_t M = B * L.Fo * L.To;
    const int64_t n = M * (L.co / 4);
    ProfScope ps("conv_first", s, 4.0 * ((double)M * L.co + (double)B * L.F * L.T));
    PF_LAUNCH(conv_first_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, x, L.w, L.bias,
                       y, M, L.co, L.To, L.T, L.stride, L.pad_lo);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Depthwise 3x1 conv along F (conv2 of non-"fuller" models): w stored [3][co].
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_dw_kernel(const float *__restrict__ x,
                                                      const float *__restrict__ w,
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ y, int64_t M, int C, int Fo,
                                                      int To, int F, int T, int axis, int stride,
                                                      int pad_lo) {
    const int c4n = C >> 2;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * c4n) return;
    const int64_t m = gid / c4n;
    const int c = (int)(gid - m * c4n) * 4;
    const int rps = Fo * To;
    const int64_t b = m / rps;
    const int r = (int)(m - b * rps);
    const int fo = r / To, to = r - fo * To;
    f32x4 o = *reinterpret_cast<const f32x4 *>(bias + c);
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
        int f = fo, t = to;
        bool ok;
        if (axis == 0) { t = to * stride - pad_lo + tap; ok = (unsigned)t < (unsigned)T; }
        else { f = fo * stride - pad_lo + tap; ok = (unsigned)f < (unsigned)F; }
        if (ok) {
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + ((b * F + f) * (int64_t)T + t) * C + c);
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + tap * C + c);
            o += xv * wv;
        }
    }
    *reinterpret_cast<f32x4 *>(y + m * C + c) = o;
}

int launch_conv_depthwise(const SubLayer &L, const float *x, float *y, int64_t B, hipStream_t s) {
    const int64_t M = B * L.Fo * L.To;
    const int64_t n = M * (L.co / 4);
    ProfScope ps("conv_depthwise", s);
    PF_LAUNCH(conv_dw_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, x, L.w, L.bias, y, M,
                       L.co, L.Fo, L.To, L.F, L.T, L.axis, L.stride, L.pad_lo);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// LayerNorm over the whole (C,F,T) sample + activation, in place, one block per sample.
// model.py:58-72: relu_after_bn ? act(LN(x)) : LN(act(x)).  Statistics: per-thread fp32
// partials, combined in fp64 (biased variance, eps 1e-5).
// ------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void ln_act_kernel(float *__restrict__ xy, const float *__restrict__ w,
                                                    const float *__restrict__ b, int n, int act,
                                                    int after_bn) {
    __shared__ double red[2][NT / 64];
    __shared__ float stat[2];
    float *x = xy + (int64_t)blockIdx.x * n;
    const int tid = threadIdx.x, n4 = n >> 2;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < n4; i += NT) {
        f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        if (!after_bn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn(v[e], act);
        }
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    double d1 = wave_sum_d((double)s1), d2 = wave_sum_d((double)s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = d1; red[1][tid >> 6] = d2; }
    __syncthreads();
    if (tid == 0) {
        double t1 = 0, t2 = 0;
        for (int i = 0; i < NT / 64; ++i) { t1 += red[0][i]; t2 += red[1][i]; }
        const double mean = t1 / n;
        double var = t2 / n - mean * mean;
        if (var < 0) var = 0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    for (int i = tid; i < n4; i += NT) {
        f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        const f32x4 wv = reinterpret_cast<const f32x4 *>(w)[i];
        const f32x4 bv = reinterpret_cast<const f32x4 *>(b)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = after_bn ? v[e] : act_fn(v[e], act);
            t = (t - mean) * rstd * wv[e] + bv[e];
            v[e] = after_bn ? act_fn(t, act) : t;
        }
        reinterpret_cast<f32x4 *>(x)[i] = v;
    }
}

int launch_ln_act(const SubLayer &L, float *xy, int64_t B, int activation, int relu_after_bn,
                  hipStream_t s) {
    const int n = L.co * L.Fo * L.To;
    ProfScope ps("ln_act", s, 4.0 * (double)B * n * 3.0);
    if (n >= 65536)
        PF_LAUNCH((ln_act_kernel<1024>), dim3((unsigned)B), dim3(1024), 0, s, xy, L.ln_w, L.ln_b, n,
                           activation, relu_after_bn);
    else
        PF_LAUNCH((ln_act_kernel<256>), dim3((unsigned)B), dim3(256), 0, s, xy, L.ln_w, L.ln_b, n,
                           activation, relu_after_bn);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Projection head MyG (model.py:122-130): d groups x (v -> u, ELU, u -> 1), then optional
// L2 normalise  y / max(||y||, 1e-12).  One block per sample, one thread per group.
// ------------------------------------------------------------------------------------
__global__ void myg_kernel(const float *__restrict__ x, const float *__restrict__ w1,
                           const float *__restrict__ b1, const float *__restrict__ w2,
                           const float *__restrict__ b2, int d, int u, int v, float *__restrict__ emb,
                           int normalize) {
    __shared__ float red[16];
    const int g = threadIdx.x;
    const float *xs = x + (int64_t)blockIdx.x * d * v;
    float y = 0.f;
    if (g < d) {
        float xin[32];
        for (int j = 0; j < v; ++j) xin[j] = xs[g * v + j];
        for (int k = 0; k < u; ++k) {
            const float *wr = w1 + ((int64_t)g * u + k) * v;
            float hsum = 0.f;
            for (int j = 0; j < v; ++j) hsum = fmaf(wr[j], xin[j], hsum);
            hsum += b1[g * u + k];
            hsum = hsum > 0.f ? hsum : expm1f(hsum);
            y = fmaf(w2[g * u + k], hsum, y);
        }
        y += b2[g];
    }
    if (normalize) {
        float ss = wave_sum(g < d ? y * y : 0.f);
        if ((g & 63) == 0) red[g >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
        y = y / fmaxf(sqrtf(tot), 1e-12f);
    }
    if (g < d) emb[(int64_t)blockIdx.x * d + g] = y;
}

int launch_myg(const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
               int d, int u, int v, int64_t B, float *emb, int normalize, hipStream_t s) {
    if (v > 32) { set_error("MyG: h/d = %d > 32 unsupported", v); return -1; }
    const int nt = ((d + 63) / 64) * 64;
    if (nt > 1024) { set_error("MyG: d = %d > 1024 unsupported", d); return -1; }
    ProfScope ps("myg", s);
    PF_LAUNCH(myg_kernel, dim3((unsigned)B), dim3(nt), 0, s, x, w1, b1, w2, b2, d, u, v, emb,
                       normalize);
    PF_HIP(hipGetLastError());
    return 0;
}

// channels-last [B][HW][C] -> NCHW [B][C][HW] (verification taps only)
__global__ void cl_to_nchw_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t total, int C,
                                  int HW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / ((int64_t)C * HW);
    const int r = (int)(i - b * C * HW);
    const int c = r / HW, hw = r - c * HW;
    y[i] = x[(b * HW + hw) * C + c];
}
int launch_cl_to_nchw(const float *x, float *y, int64_t B, int C, int HW, hipStream_t s) {
    const int64_t total = B * C * HW;
    PF_LAUNCH(cl_to_nchw_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, x, y, total, C, HW);
    PF_HIP(hipGetLastError());
    return 0;
}

// pfann_prewarm: one empty launch per translation unit makes the runtime load this unit's code object now
__global__ void noop_encoder_kernel() {}
int prewarm_encoder() {
    hipLaunchKernelGGL(noop_encoder_kernel, dim3(1), dim3(1), 0, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace pfann
