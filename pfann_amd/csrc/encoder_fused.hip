// LayerNorm-fused encoder path for gfx950 ("fuller" models: every conv2 is a full conv).
//
// The reference normalises every conv output over the whole (C,F,T) sample before the next
// conv (model.py:58-72), i.e. 16 grid-wide reductions per forward.  Unfused that is two extra
// HBM round trips per sub-layer (csrc/encoder.hip: ln_act_kernel, 20 % of the step).  Here:
//
//   producer epilogue : z = PRE(acc + bias) is stored once, and the block reduces its tile to
//                       per-sample partial (sum, sum of squares) in a FIXED order (registers ->
//                       wave shuffle -> LDS -> one thread), written to part[b][slot][2];
//   consumer prologue : reduces the <= 64 partials of the samples its tile touches (fp64) to
//                       (mean, rstd) in LDS;
//   consumer A-loader : while staging the activation tile global -> registers -> LDS it applies
//                       v = POST((z - mean) * rstd * W + B), W/B being the full-size LayerNorm
//                       affine tensors in the same channels-last order (L2/Infinity-Cache
//                       resident, shared by the whole batch); padding stays exactly 0.
//   (PRE, POST) = (identity, act) for relu_after_bn, (act, identity) otherwise.
//
// No atomics: results are bit-reproducible run to run.  The normalised tensor never exists in HBM.
#include <stdlib.h>

#include "kernels.h"
#include <set>
#include <string>
#include <type_traits>

namespace pfann {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// -DPFANN_TILE_TRACE (tuning builds only, tools/ubench/tile_trace.py; never the shipped library): every workgroup of
// conv_gemm_ln_w22_kernel leaves four 100 MHz timestamps (start, first MFMA sub-step, end of the K loop, end) and its
// HW_ID / XCC_ID registers in a caller-provided buffer, from which the tool rebuilds each CU's timeline.
#ifdef PFANN_TILE_TRACE
__device__ unsigned long long *g_tile_trace = nullptr;
__device__ unsigned g_tile_trace_cap = 0;
__device__ int g_tile_trace_rps = 0;          // only launches with this many rows per sample leave stamps (0: all)
#define TILE_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define TILE_STAMP(var)
#endif

struct FusedGemmParams {
    const float *x, *w, *bias;
    const void *w_hi, *w_lo;     // SPLIT kernels: fp16 halves of w * 2^e, [N][K]
    float w_inv_scale;           // 2^-e
    float *y;
    int M;                       // B * rows_per_sample (< 2^31, checked by the launcher)
    int N, K, Ci;
    int rows_per_sample, To, F, T;
    int rps_shift, To_shift;     // rows_per_sample and To are powers of two (fused_supported)
    int axis, stride, pad_lo, in_len;
    int64_t tap_stride;
    int n_tiles_n;
    FastDiv dv_group, dv_tile, dv_n;   // divisors 64 * (rows_per_sample / BM) * n_tiles_n, 64 * n_tiles_n, n_tiles_n
    int k_begin, k_end;
    // input normalisation
    const float *in_stats;       // [n_samples][2] = (mean, rstd) of the input, from ln_finalize_kernel
    const float *ln_w, *ln_b;
    int64_t in_elems;            // F*T*Ci
    // output statistics
    float *out_part; int out_P;
    int act, after_bn;
    int n_samples;
    // FIRST variant: the input is the C_in = 1 first conv computed on the fly from the log-mel
    const float *w1, *b1;        // first conv weights [3][Ci] and bias [Ci]
    int T0, s1, pad1;            // mel frames, first conv stride and left pad along T
    // SPLITK variant: the grid is blocks_mn x n_splits; split ks covers k in [k_begin + ks*k_chunk, ... + k_chunk)
    int blocks_mn, k_chunk;
};

// RELU_BN = true: the default model (ReLU applied after LayerNorm) with the activation folded
// into straight-line code; false: generic (ELU and/or activation before LayerNorm).
// FIRST = true: sub-layer 1 with sub-layer 0 (the C_in = 1 conv, model.py:20 with i = 1) folded into
// the A-loader: z1 = b1 + sum_tap1 w1[tap1] * mel[f][2t + tap1] is recomputed per element (3 FMAs, same
// order as conv_first_stats_kernel, so bit-identical) instead of streaming the 2 MiB/segment
// tensor through HBM; p.x is then the log-mel batch [B][F][T0].
// UNI = true: Ci % 32 == 0, see "Operand addressing" below.
// BK = K-tile depth: 32, or 16 for the 4-wave 128x128 variant that fits three workgroups per CU.
// SPLIT: operands as two fp16 terms each (x = hi + lo to 2^-22 relative), three fp16 MFMAs per product
// term (hi*hi + lo*hi + hi*lo, fp32 accumulate) instead of one fp32 MFMA: ~fp32 accuracy at 3/16 of the
// MFMA cycles.  Opt-in (pfann_set_encoder_precision); the default path is the exact fp32 one.
// SPLITK: small batches (one query = 19 segments): the deep layers have a handful of output tiles and K loops of up to
// 96 K-tiles, a serial MFMA chain of 70 us on 16 of 256 CUs.  The K range is cut into chunks of <= 8 K-tiles, one
// workgroup each; a workgroup stores its raw partial tile (no bias, no statistics) to p.y[ks][M][N] and
// splitk_reduce_ln_kernel adds the chunks in a FIXED order (deterministic), the bias, and takes the statistics.
template <int BM, int BN, int WM, int WN, bool RELU_BN, bool FIRST, bool UNI, int BK = 32, bool SPLIT = false, bool SPLITK = false>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), (BM / WM) * (BN / WN) == 8 ? 4 : (BK == 16 ? 3 : 1))
void conv_gemm_ln_kernel(FusedGemmParams p) {
    constexpr int LDK = BK + 4, TPR = BK / 4;         // TPR loader threads per tile row (4 floats each)
    constexpr int WAVES_N = BN / WN;
    constexpr int NWAVES = (BM / WM) * WAVES_N, NT = 64 * NWAVES;
    static_assert(NWAVES == 4 || NWAVES == 8, "4 or 8 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RPT = NT / TPR;                      // tile rows covered per loader pass
    constexpr int AR = BM / RPT, BR = BN / RPT;
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK + 2 * BN * LDK];     // one array: the epilogue reuses it whole
    float *const As = smem, *const Bs = smem + 2 * BM * LDK;
    __shared__ __attribute__((aligned(16))) float s_w1[FIRST ? 4 * 256 : 4];   // FIRST: w1[3][Ci], b1[Ci]; Ci <= 256

    // Every launch constant the prologue touches, requested in ONE batch of scalar loads: left to itself the compiler
    // loads them where first used, and each of those five `s_waitcnt lgkmcnt(0)` cost a scalar-cache round trip of
    // 1-2 k cycles on a chip busy streaming operands (in-kernel timestamps: 11 k cycles before the first operand load).
    asm volatile("" :: "s"(p.x), "s"(p.w), "s"(p.in_stats), "s"(p.ln_w), "s"(p.ln_b), "s"(p.in_elems), "s"(p.tap_stride),
                 "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.Ci), "s"(p.rows_per_sample), "s"(p.To), "s"(p.F), "s"(p.T));
    asm volatile("" :: "s"(p.rps_shift), "s"(p.To_shift), "s"(p.axis), "s"(p.stride), "s"(p.pad_lo), "s"(p.in_len),
                 "s"(p.n_tiles_n), "s"(p.k_begin), "s"(p.k_end), "s"(p.n_samples), "s"(p.dv_group.mul), "s"(p.dv_group.shift),
                 "s"(p.dv_tile.mul), "s"(p.dv_tile.shift), "s"(p.dv_n.mul), "s"(p.dv_n.shift));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int L = xcd_remap(blockIdx.x, gridDim.x);
    int ks = 0;
    if (SPLITK) { ks = L / p.blocks_mn; L -= ks * p.blocks_mn; }
    const int k_begin = SPLITK ? p.k_begin + ks * p.k_chunk : p.k_begin;
    const int k_end = SPLITK ? (p.k_end < k_begin + p.k_chunk ? p.k_end : k_begin + p.k_chunk) : p.k_end;
    const int rps = p.rows_per_sample;
    int mt, nt;
    if (rps >= 2 * BM) {
        // several tiles per sample: run tile `ti` of GS neighbouring samples back to back, so the
        // LayerNorm affine slice they all stream (identical for every sample) stays in this XCD's L2
        constexpr int GS = 64;
        const int tps = rps / BM;
        const int per_group = GS * tps * p.n_tiles_n;
        const int g = fastdiv(L, p.dv_group);
        const int left = p.n_samples - g * GS;
        const int gs = left < GS ? left : GS;
        const int r = L - g * per_group;
        const int ti = gs == GS ? fastdiv(r, p.dv_tile) : r / (gs * p.n_tiles_n);     // the last group may be short
        const int r2 = r - ti * (gs * p.n_tiles_n);
        const int bi = fastdiv(r2, p.dv_n);
        nt = r2 - bi * p.n_tiles_n;
        mt = (g * GS + bi) * tps + ti;
    } else {
        mt = fastdiv(L, p.dv_n);
        nt = L - mt * p.n_tiles_n;
    }
    const int m0 = mt * BM;
    const int n0 = nt * BN;
    const int b_first = m0 >> p.rps_shift;
    if (FIRST) {     // first-conv weights and bias -> LDS (visible after the barrier that publishes tile 0)
        for (int i = tid; i < 4 * p.Ci; i += NT) s_w1[i] = i < 3 * p.Ci ? p.w1[i] : p.b1[i - 3 * p.Ci];
        __syncthreads();
    }

    const int col4 = tid & (TPR - 1), rowq = tid / TPR;
    // bounds-checked buffer loads (OOB lanes read 0): activation window from the first sample of
    // the tile, LayerNorm affine tensors addressed sample-relative, weights by output channel
    const int64_t x_elems = FIRST ? (int64_t)p.F * p.T0 : p.in_elems;       // per-sample size of p.x
    const __amdgpu_buffer_rsrc_t srd_a =
        make_srd(p.x + (int64_t)b_first * x_elems, (unsigned long long)(p.n_samples - b_first) * x_elems * 4ull);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.ln_w, (unsigned long long)p.in_elems * 4ull);
    const __amdgpu_buffer_rsrc_t srd_lb = make_srd(p.ln_b, (unsigned long long)p.in_elems * 4ull);
    const __amdgpu_buffer_rsrc_t srd_b = make_srd(p.w, (unsigned long long)p.N * p.K * 4ull);
    const __amdgpu_buffer_rsrc_t srd_bh = make_srd(p.w_hi, SPLIT ? (unsigned long long)p.N * p.K * 2ull : 0ull);
    const __amdgpu_buffer_rsrc_t srd_bl = make_srd(p.w_lo, SPLIT ? (unsigned long long)p.N * p.K * 2ull : 0ull);
    int aoff[AR], arel[AR], ap0[AR], atq[AR];
    float amu[AR], ars[AR];
    // Branch-free on purpose: with the row set-up under `if (m < M)` the statistics loads sat inside a divergent
    // region whose join needs the loaded value at once -- two serialised memory round trips (`global_load; s_waitcnt
    // vmcnt(0)` twice in the ISA) before the first operand load was even issued, 12 k cycles of prologue per tile
    // (in-kernel timestamps).  Now: two bounds-checked loads (rows m >= M read 0) that are first needed by the
    // LayerNorm transform of K-tile 0.
    const __amdgpu_buffer_rsrc_t srd_st = make_srd(p.in_stats, (unsigned long long)p.n_samples * 8ull);
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + rowq + RPT * i;
        const bool mok = m < p.M;
        const int b = m >> p.rps_shift;
        amu[i] = buf_load1(srd_st, mok ? (unsigned)b * 8u : BUF_OOB);
        ars[i] = buf_load1(srd_st, mok ? (unsigned)b * 8u + 4u : BUF_OOB);
        const int r = m & (rps - 1);
        const int fo = r >> p.To_shift, to = r & (p.To - 1);
        int rel, a0;
        if (p.axis == 0) { a0 = to * p.stride - p.pad_lo; rel = (fo * p.T + a0) * p.Ci; }
        else { a0 = fo * p.stride - p.pad_lo; rel = (a0 * p.T + to) * p.Ci; }
        const int sl = b - b_first;
        arel[i] = rel;
        aoff[i] = sl * (int)p.in_elems + rel;
        atq[i] = 0;
        if (FIRST) {   // mel element of tap2 = 0, tap1 = 0:  ((b*F + f_in) * T0 + to*s1 - pad1)
            atq[i] = to * p.s1 - p.pad1;
            aoff[i] = (sl * p.F + a0) * p.T0 + atq[i];
        }
        ap0[i] = mok ? a0 : -(1 << 20);          // rows past M: every tap out of range -> all offsets out of range
    }
    // Operand addressing.  UNI (Ci % BK == 0): every thread of the block is in the same filter tap
    // for a whole K-tile, so the K position lives in a scalar register (the buffer instruction's
    // soffset) and the per-thread byte offsets (with the row-validity test folded in as an
    // out-of-range offset) only change when the tap does: no vector ALU work per K-tile, which
    // matters because every VALU instruction issued here displaces MFMA issue on the same SIMD
    // (measured: the loop without staging runs at 97 % of the fp32 MFMA peak).
    // !UNI: generic per-thread (tap, channel) cursor, offsets recomputed for every tile.
    constexpr int NA = FIRST ? 3 : 1;
    unsigned va[AR][NA], vr[AR], vb[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        const int n = n0 + rowq + RPT * j;
        vb[j] = n < p.N ? (unsigned)n * (unsigned)p.K * 4u + (UNI ? (unsigned)col4 * 16u : 0u) : BUF_OOB;
    }
    const int tap_stride = (int)p.tap_stride;
    int kap = k_begin + (UNI ? 0 : col4 * 4);            // UNI: uniform (scalar) cursor
    int tap = kap / p.Ci, c = kap - tap * p.Ci;
    // byte offsets of this thread's operands for filter tap `tp`, channel offset `cc`
    auto set_offsets = [&](int tp, int cc, bool kok) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const bool ok = kok && (unsigned)(ap0[i] + tp) < (unsigned)p.in_len;
            const int toff = tp * tap_stride + cc;
            vr[i] = ok ? (unsigned)(arel[i] + toff) * 4u : BUF_OOB;
            if (FIRST) {
                // three neighbouring log-mel frames of input row f_in = ap0 + tp (tp = tap2)
#pragma unroll
                for (int t1 = 0; t1 < NA; ++t1)
                    va[i][t1] = (ok && (unsigned)(atq[i] + t1) < (unsigned)p.T0)
                                    ? (unsigned)(aoff[i] + tp * p.T0 + t1) * 4u : BUF_OOB;
            } else if (!(UNI && !FIRST && !SPLIT)) {
                va[i][0] = ok ? (unsigned)(aoff[i] + toff) * 4u : BUF_OOB;
            }
        }
    };
    if (UNI) set_offsets(tap, col4 * 4, true);
    // EA ("early A"): the activation loads -- the only operand that comes from HBM rather than L2 -- run one
    // K-tile ahead of the others: tile kt+2's are issued right after tile kt+1 has gone to LDS, a full K-tile
    // (instead of three quarters of one) before they are needed, at no register cost (their destination
    // registers have just been consumed).  Measured by ablation: the activation loads alone cost 8 % on the
    // K = 384 layers, the LayerNorm-affine loads 5 %, the weight loads 2 %.
    constexpr bool EA = UNI && !FIRST && !SPLIT;
    int tapA = tap, cA = c;
    auto set_offsets_A = [&](int tp) {
#pragma unroll
        for (int i = 0; i < AR; ++i)
            va[i][0] = (unsigned)(ap0[i] + tp) < (unsigned)p.in_len ? (unsigned)(aoff[i] + tp * tap_stride + col4 * 4) * 4u : BUF_OOB;
    };

    struct Stage {                       // one K-tile of prefetched operands, in registers
        f32x4 ra[2][AR], rw[AR], rbb[AR], rb[BR];   // FIRST: ra[0][i][0..2] = the three log-mel taps; EA: two sets,
                                                  // tile t's activations in ra[t & 1] (two K-tiles of prefetch)
                                                  // SPLIT: rb[j] = (hi k0..3 as 2 dwords, lo k0..3 as 2 dwords)
        int cc;                                   // channel of element 0 (first-conv weights come from LDS)
    };
    auto load_A = [&](Stage &S, int par) {        // EA only; par is a compile-time constant at every call site
#pragma unroll
        for (int i = 0; i < AR; ++i) S.ra[par][i] = buf_load4(srd_a, va[i][0], cA * 4);
        cA += BK;
        if (cA >= p.Ci) {                // uniform branch: next filter tap
            cA = 0;
            ++tapA;
            set_offsets_A(tapA);
        }
    };
    auto load_tile = [&](Stage &S) {
        int so_c = 0, so_k = 0;          // scalar byte offsets (UNI)
        bool kok = true;
        if (UNI) {
            so_c = c * 4;
            so_k = kap * 4;
        } else {
            kok = kap < k_end;
            set_offsets(tap, c, kok);
        }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            if (FIRST) {
                // the three log-mel taps of a row do not depend on the channel: with a uniform cursor they
                // are fetched on the first K-tile of each filter tap only and stay in S.ra for the others
                if (!UNI || c == 0 || kap == k_begin) {
#pragma unroll
                    for (int t1 = 0; t1 < NA; ++t1) S.ra[0][i][t1] = buf_load1(srd_a, va[i][t1]);
                }
            } else if (!EA) {
                S.ra[0][i] = buf_load4(srd_a, va[i][0], so_c);
            }
            S.rw[i] = buf_load4(srd_w, vr[i], so_c);
            S.rbb[i] = buf_load4(srd_lb, vr[i], so_c);
        }
        S.cc = kok ? c + (UNI ? col4 * 4 : 0) : 0;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            if (SPLIT) {     // UNI only: 4 k as halves = 8 bytes from each of the hi / lo arrays
                const unsigned o2 = vb[j] == BUF_OOB ? BUF_OOB : vb[j] >> 1;
                const u32x2_t hh = buf_load2u(srd_bh, o2, so_k >> 1), ll = buf_load2u(srd_bl, o2, so_k >> 1);
                S.rb[j] = __builtin_bit_cast(f32x4, u32x4_t{hh[0], hh[1], ll[0], ll[1]});
            } else {
                S.rb[j] = buf_load4(srd_b, (UNI || vb[j] == BUF_OOB) ? vb[j] : (kok ? vb[j] + (unsigned)kap * 4u : BUF_OOB), so_k);
            }
        }
        kap += BK;
        c += BK;
        if (UNI) {
            if (c >= p.Ci) {             // uniform branch: next filter tap
                c = 0;
                ++tap;
                set_offsets(tap, col4 * 4, true);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) { const bool w = c >= p.Ci; c -= w ? p.Ci : 0; tap += w ? 1 : 0; }
        }
    };
    // v = POST((z - mean) * rstd * W + B), two elements per packed-fp32 instruction.  Rows outside the
    // input (conv padding, m >= M, k >= k_end) need no select: their offsets were out of range, so
    // W = B = 0 arrive from the buffer unit and (z - mean) * rstd * 0 + 0 = +-0, which every
    // activation maps to 0.
    auto store_tile = [&](const Stage &S, float *Ad, float *Bd, int par) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const f32x2 mu2 = {amu[i], amu[i]}, rs2 = {ars[i], ars[i]};
            f32x4 v, w1v[3], b1v;
            if (FIRST) {
#pragma unroll
                for (int t1 = 0; t1 < 3; ++t1) w1v[t1] = *reinterpret_cast<const f32x4 *>(&s_w1[t1 * p.Ci + S.cc]);
                b1v = *reinterpret_cast<const f32x4 *>(&s_w1[3 * p.Ci + S.cc]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 z = {S.ra[par][i][2 * h], S.ra[par][i][2 * h + 1]};
                if (FIRST) {   // same FMA order as conv_first_stats_kernel: bias, then taps 0, 1, 2
                    z = f32x2{b1v[2 * h], b1v[2 * h + 1]};
#pragma unroll
                    for (int t1 = 0; t1 < 3; ++t1)
                        z = __builtin_elementwise_fma(f32x2{S.ra[0][i][t1], S.ra[0][i][t1]},
                                                      f32x2{w1v[t1][2 * h], w1v[t1][2 * h + 1]}, z);
                    if (!RELU_BN && !p.after_bn) { z[0] = act_fn(z[0], p.act); z[1] = act_fn(z[1], p.act); }   // PRE of sub-layer 0
                }
                const f32x2 w2 = {S.rw[i][2 * h], S.rw[i][2 * h + 1]}, b2 = {S.rbb[i][2 * h], S.rbb[i][2 * h + 1]};
                f32x2 t = __builtin_elementwise_fma((z - mu2) * rs2, w2, b2);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (RELU_BN) t[e] = fmaxf(t[e], 0.f);
                    else t[e] = p.after_bn ? act_fn(t[e], p.act) : t[e];
                    v[2 * h + e] = t[e];
                }
            }
            if (SPLIT) {
                // row = [hi: BK halves][lo: BK halves][pad]; this thread's 4 k go to byte 8*col4 of each half-row
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                f16x4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = RELU_BN ? fminf(v[e], 65000.f) : fminf(fmaxf(v[e], -65000.f), 65000.f);
                    hi[e] = (_Float16)c;
                    lo[e] = (_Float16)(c - (float)hi[e]);
                }
                char *row = reinterpret_cast<char *>(&Ad[(rowq + RPT * i) * LDK]);
                *reinterpret_cast<f16x4 *>(row + col4 * 8) = hi;
                *reinterpret_cast<f16x4 *>(row + BK * 2 + col4 * 8) = lo;
            } else {
                *reinterpret_cast<f32x4 *>(&Ad[(rowq + RPT * i) * LDK + col4 * 4]) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            if (SPLIT) {
                const u32x4_t r = __builtin_bit_cast(u32x4_t, S.rb[j]);
                char *row = reinterpret_cast<char *>(&Bd[(rowq + RPT * j) * LDK]);
                *reinterpret_cast<u32x2_t *>(row + col4 * 8) = u32x2_t{r[0], r[1]};
                *reinterpret_cast<u32x2_t *>(row + BK * 2 + col4 * 8) = u32x2_t{r[2], r[3]};
            } else {
                *reinterpret_cast<f32x4 *>(&Bd[(rowq + RPT * j) * LDK + col4 * 4]) = S.rb[j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS double buffer, one barrier per K-tile (see csrc/encoder.hip for the schedule)
    const int nk = (k_end - k_begin + BK - 1) / BK;
    const int l31 = lane & 31, lhalf = lane >> 5;
    Stage S;
    if (EA) { set_offsets_A(tapA); load_A(S, 0); }
    load_tile(S);
    store_tile(S, As, Bs, 0);
    if (EA && nk > 1) load_A(S, 1);
    if (EA && nk > 2) load_A(S, 0);
    __syncthreads();
    // EA: the K loop is unrolled by two so that the register set holding tile kt+1's activations (kt+1 & 1) is a
    // compile-time choice; the activations of tile kt+3 are requested as soon as tile kt+1's have gone to LDS
    auto ea_ktile = [&](int kt, auto parc) {
        constexpr int PB = decltype(parc)::value;
        const float *Ac = As + PB * (BM * LDK), *Bc = Bs + PB * (BN * LDK);
        float *An = As + (PB ^ 1) * (BM * LDK), *Bn = Bs + (PB ^ 1) * (BN * LDK);
        const bool more = kt + 1 < nk;
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
                if (more) load_tile(S);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk == BK / 8 - 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) store_tile(S, An, Bn, PB ^ 1);
                if (kt + 3 < nk) load_A(S, PB ^ 1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[j][s], a4[i][s], acc[i][j], 0, 0, 0);   // C^T: rows = n
        }
        __syncthreads();
    };
    if constexpr (EA) {
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ea_ktile(kt, std::integral_constant<int, 0>{});
            ea_ktile(kt + 1, std::integral_constant<int, 1>{});
        }
        if (kt < nk) ea_ktile(kt, std::integral_constant<int, 0>{});
    } else
    for (int kt = 0; kt < nk; ++kt) {
        // MFMAs on LDS buffer kt&1 while tile kt+1 goes global -> registers -> buffer (kt+1)&1
        const float *Ac = As + (kt & 1) * (BM * LDK), *Bc = Bs + (kt & 1) * (BN * LDK);
        float *An = As + ((kt + 1) & 1) * (BM * LDK), *Bn = Bs + ((kt + 1) & 1) * (BN * LDK);
        const bool more = kt + 1 < nk;   // UNI offsets are not range-checked against k_end: no loads past it
        if constexpr (SPLIT) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {           // 16 k per fp16 MFMA; lane half h holds k = 8h .. 8h+7
                const char *arow[TM], *brow[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    arow[i] = reinterpret_cast<const char *>(&Ac[(wm * WM + i * 32 + l31) * LDK]) + ks * 32 + lhalf * 16;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    brow[j] = reinterpret_cast<const char *>(&Bc[(wn * WN + j * 32 + l31) * LDK]) + ks * 32 + lhalf * 16;
                f16x8 ah[TM], bh[TN], bl[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const f16x8 *>(brow[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const f16x8 *>(arow[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const f16x8 *>(brow[j] + BK * 2);
                if (ks == 0) {
                    if (more) load_tile(S);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ks == BK / 16 - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) store_tile(S, An, Bn, 0);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                    }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f16x8 al = *reinterpret_cast<const f16x8 *>(arow[i] + BK * 2);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al, acc[i][j], 0, 0, 0);
                }
            }
        } else {
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
                    if (more) load_tile(S);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kk == BK / 8 - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) store_tile(S, An, Bn, 0);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[j][s], a4[i][s], acc[i][j], 0, 0, 0);   // C^T: rows = n
            }
        }
        __syncthreads();
    }

    // ---- epilogue: z = PRE(acc + bias) -> HBM; per-sample partial statistics -------------
    // The MFMAs computed C^T (weights as the A operand), so in a 32x32 tile a lane holds ONE output row
    // m = lane&31 and the columns n = 8g + 4*(lane>>5) + e for register 4g + e: four consecutive channels
    // per register quad, stored with one dwordx4 (the texture addresser takes as long for a 64-lane dword
    // store as for a dwordx4 one; scalar stores made the epilogue 6 % of the kernel).
    const __amdgpu_buffer_rsrc_t srd_y = make_srd(p.y + (SPLITK ? (int64_t)ks * p.M * p.N : 0) + (int64_t)m0 * p.N,
                                                  (unsigned long long)(p.M - m0) * p.N * 4ull);
    const __amdgpu_buffer_rsrc_t srd_bias = make_srd(p.bias, (unsigned long long)p.N * 4ull);
    // The output tile goes through LDS so that it reaches memory as whole rows: straight from the accumulators a store
    // instruction covers 32 rows x 32 B (a quarter cache line per row, four waves completing each 512 B row at different
    // times); from LDS a half-wave writes one row's 512 contiguous bytes.  Measured on a stripped clone of this kernel
    // (tools/ubench/gemm_tile.hip): +2.7 % at K = 384, +1.5 % at K = 768.  LDC = BN + 4: lanes 0..7 of a b128 write land
    // on all 32 banks.
    constexpr int LDC = BN + 4;
    constexpr bool STAGE_C = BM * LDC + 2 * BM * WAVES_N + 8 <= 2 * BM * LDK + 2 * BN * LDK;
    float *Cs = smem;                                   // [BM][LDC]
    float *red1 = smem + (STAGE_C ? BM * LDC : 0);      // [BM][WAVES_N] row (or sub-tile) sums; As/Bs are free now
    float *red2 = red1 + BM * WAVES_N;
    const int G = rps >= BM ? BM : rps;          // rows per statistics group inside the tile
    // No validity selects: columns n >= N (N % 4 == 0) have zero weights and bias (z = 0 adds nothing to
    // the sums) and an out-of-range store offset; rows m >= M lie beyond srd_y and form whole statistics
    // groups (M is a multiple of G) that are never written out.
    const unsigned rowbytes = (unsigned)p.N * 4u;
    // ALL bias chunks first, then nothing but math and stores.  gfx950 counts loads and stores in ONE in-order counter
    // (vmcnt): with a bias load issued after every store, each "wait for the bias" also waited for the previous store
    // to be acknowledged by memory -- eight serialised store round trips per wave made the epilogue 25 k cycles
    // (in-kernel timestamps), as long as four K-tiles.
    f32x4 bias4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * WN + j * 32 + 8 * g + 4 * lhalf;
            bias4[j][g] = buf_load4(srd_bias, (!SPLITK && n < p.N) ? (unsigned)n * 4u : BUF_OOB);     // SPLITK: raw partial sums
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float a1 = 0.f, a2 = 0.f;        // this lane's row m = wm*WM + i*32 + l31
        const unsigned ro = (unsigned)(wm * WM + i * 32 + l31) * rowbytes;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WN + j * 32 + 8 * g + 4 * lhalf;
                const unsigned nb = n < p.N ? (unsigned)n * 4u : BUF_OOB;
                const f32x4 b4v = bias4[j][g];
                f32x4 z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float z = SPLIT ? fmaf(acc[i][j][4 * g + e], p.w_inv_scale, b4v[e]) : acc[i][j][4 * g + e] + b4v[e];
                    if (!SPLITK && !RELU_BN && !p.after_bn) z = act_fn(z, p.act);
                    a1 += z;
                    a2 = fmaf(z, z, a2);
                    z4[e] = z;
                }
                if (STAGE_C) *reinterpret_cast<f32x4 *>(&Cs[(wm * WM + i * 32 + l31) * LDC + wn * WN + j * 32 + 8 * g + 4 * lhalf]) = z4;
                else buf_store4(srd_y, nb + ro, z4);
            }
        a1 += __shfl_xor(a1, 32, 64);    // the other half of the row's columns
        a2 += __shfl_xor(a2, 32, 64);
        if (lhalf == 0) {                // per-row sums of this wave's column range
            red1[(wm * WM + i * 32 + l31) * WAVES_N + wn] = a1;
            red2[(wm * WM + i * 32 + l31) * WAVES_N + wn] = a2;
        }
    }
    __syncthreads();
    if (STAGE_C) {
        constexpr int LPR = BN / 4, RPP = NT / LPR;     // lanes per tile row, rows per pass
        const int cl = tid % LPR, rr = tid / LPR;
        const int n = n0 + cl * 4;
        const unsigned nb = n < p.N ? (unsigned)n * 4u : BUF_OOB;
#pragma unroll
        for (int it = 0; it < BM / RPP; ++it) {
            const int row = it * RPP + rr;
            buf_store4(srd_y, nb + (unsigned)row * rowbytes, *reinterpret_cast<const f32x4 *>(&Cs[row * LDC + cl * 4]));
        }
    }
    if (SPLITK) return;                  // statistics come from splitk_reduce_ln_kernel
    // rows -> statistics groups of G rows (one per sample touched), in a fixed order: thread r owns row r
    float t1 = 0.f, t2 = 0.f;
    if (tid < BM) {
#pragma unroll
        for (int w = 0; w < WAVES_N; ++w) { t1 += red1[tid * WAVES_N + w]; t2 += red2[tid * WAVES_N + w]; }
        for (int o = (G < 64 ? G : 64) >> 1; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
    }
    if (G > 64) {                        // BM = G = 128: one group spanning two waves
        float *x2 = red2 + BM * WAVES_N;
        if (tid < BM && lane == 0) { x2[2 * wave] = t1; x2[2 * wave + 1] = t2; }
        __syncthreads();
        t1 = x2[0] + x2[2];
        t2 = x2[1] + x2[3];
    }
    if (tid < BM && (tid & (G - 1)) == 0) {
        const int mg = m0 + tid;
        if (mg < p.M) {
            const int b = mg >> p.rps_shift;
            const int slot = ((mg & (rps - 1)) / G) * p.n_tiles_n + nt;
            float *o = p.out_part + ((int64_t)b * p.out_P + slot) * 2;
            o[0] = t1;
            o[1] = t2;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The stride-2 three-tap convolutions with FIVE channel blocks per pair of outputs instead of six (Winograd F(2,2) on
// the even input samples; measured on a stripped clone, tools/ubench/gemm_w22.hip: 1.52 ms vs 1.76 ms for the same
// skeleton doing the plain convolution, 0.865 of the fp32 MFMA peak counted in the convolution's own flops).
// Two neighbouring outputs along the convolved axis (input positions 2o .. 2o+4 = e0 o0 e1 o1 e2, right/bottom zero pad):
//     y0 = W0 e0 + W1 o0 + W2 e1          y1 = W0 e1 + W1 o1 + W2 e2
//     A0 = (e0 - e1) W0 + o0 W1           A1 = e1 (W0 + W2)           A2 = (e1 - e2) (-W2) + o1 W1
//     y0 = A0 + A1                        y1 = A1 + A2
// e, o = the LayerNorm + activation of the previous sub-layer applied on load, as in conv_gemm_ln_kernel; the
// differences are taken AFTER it, in the loader.  p.w = the layer's [N][4][Ci] table {W1, W0, -W2, W0 + W2}.
// A workgroup = 128 output rows (64 pairs) x 128 channels, 8 waves of 32 pairs x 32 channels with three accumulator
// blocks (48 VGPRs); a channel chunk of 32 takes four sub-steps through the double-buffered LDS:
//     [o0; o1] x W1 (128 rows: A0, A2)    (e0 - e1) x W0 (A0)    (e1 - e2) x -W2 (A2)    e1 x (W0 + W2) (A1)
// = 80 MFMAs per wave and chunk instead of 96.  Results differ from the plain kernel's by fp32 rounding (the sums are
// associated differently); every summation order is fixed, so they are bit-reproducible run to run.
// Used for full convs with stride 2, pad_lo 0, all three taps live, an even output length along the convolved axis and
// Ci % 32 == 0 (launch_conv_gemm_ln); everything else keeps conv_gemm_ln_kernel.
// FIRST: sub-layer 1 with the C_in = 1 conv folded into the loader as in conv_gemm_ln_kernel<.., FIRST>: p.x is the log-mel
// batch; the 5 x 3 log-mel values a pair needs do not depend on the channel, so they are loaded ONCE per tile and the
// activation operand costs no memory traffic in the loop at all.
template <bool RELU_BN, bool FIRST>
__global__ __launch_bounds__(512, 4) void conv_gemm_ln_w22_kernel(FusedGemmParams p) {
    constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 4, NT = 512, WAVES_N = 4, NP = 64;
    // The weight tiles go straight into LDS (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass; round 5:
    // all conv GEMMs -1.8 %, profiles/r5/NOTES.md section 8), one __shared__ object per buffer so that the compiler's
    // LDS-DMA wait tracking tells them apart (alias scopes are per object).  The image is lane-linear, so rows cannot be
    // padded: 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7) and the fragment reads apply the same XOR (as in
    // scan_f16_qres_kernel).  The activation tiles keep their padded rows (they are written from registers, after the
    // LayerNorm).  The epilogue's C tile lies over all three objects (+ 640 floats so that it fits).
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDK];
    __shared__ __attribute__((aligned(16))) float s_b0[BN * BK];
    __shared__ __attribute__((aligned(16))) float s_b1[BN * BK + 640];
    float *const As = smem;
    __shared__ __attribute__((aligned(16))) float s_w1[FIRST ? 4 * 256 : 4];   // FIRST: w1[3][Ci], b1[Ci]; Ci <= 256
    __shared__ float s_ml[FIRST ? NP * 16 : 1];          // FIRST: the 5 x 3 log-mel values of every pair of the tile
    asm volatile("" :: "s"(p.x), "s"(p.w), "s"(p.in_stats), "s"(p.ln_w), "s"(p.ln_b), "s"(p.in_elems), "s"(p.tap_stride),
                 "s"(p.M), "s"(p.N), "s"(p.Ci), "s"(p.rows_per_sample), "s"(p.To), "s"(p.F), "s"(p.T));
    asm volatile("" :: "s"(p.rps_shift), "s"(p.To_shift), "s"(p.axis), "s"(p.in_len), "s"(p.n_tiles_n), "s"(p.n_samples),
                 "s"(p.dv_group.mul), "s"(p.dv_group.shift), "s"(p.dv_tile.mul), "s"(p.dv_tile.shift), "s"(p.dv_n.mul), "s"(p.dv_n.shift));
    TILE_STAMP(ts_start);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int rps = p.rows_per_sample;
    int mt, nt;
    if (rps >= 2 * BM) {             // same tile order as conv_gemm_ln_kernel: tile ti of 64 neighbouring samples back to back
        constexpr int GS = 64;
        const int tps = rps / BM;
        const int per_group = GS * tps * p.n_tiles_n;
        const int g = fastdiv(L, p.dv_group);
        const int left = p.n_samples - g * GS;
        const int gs = left < GS ? left : GS;
        const int r = L - g * per_group;
        const int ti = gs == GS ? fastdiv(r, p.dv_tile) : r / (gs * p.n_tiles_n);
        const int r2 = r - ti * (gs * p.n_tiles_n);
        const int bi = fastdiv(r2, p.dv_n);
        nt = r2 - bi * p.n_tiles_n;
        mt = (g * GS + bi) * tps + ti;
    } else {
        mt = fastdiv(L, p.dv_n);
        nt = L - mt * p.n_tiles_n;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int b_first = m0 >> p.rps_shift;
    if (FIRST) {     // first-conv weights and bias -> LDS (visible after the barrier below)
        for (int i = tid; i < 4 * p.Ci; i += NT) s_w1[i] = i < 3 * p.Ci ? p.w1[i] : p.b1[i - 3 * p.Ci];
    }
    const int col4 = tid & 7, rowq = tid >> 3;           // loader thread = pair rowq, channels 4*col4 .. +3 of the chunk
    // tile-relative output rows of pair q: along T (axis 0) the pair is two consecutive rows; along F (axis 1) the rows
    // [fo][to] and [fo + 1][to], To apart (m0 is a multiple of 2*To: 128 % (2*To) == 0 is checked by the launcher)
    auto pair_row0 = [&](int q) { return p.axis == 0 ? 2 * q : ((q >> p.To_shift) << (p.To_shift + 1)) + (q & (p.To - 1)); };
    const int prow_stride = p.axis == 0 ? 1 : p.To;
    const int64_t x_elems = FIRST ? (int64_t)p.F * p.T0 : p.in_elems;       // per-sample size of p.x
    const __amdgpu_buffer_rsrc_t srd_a =
        make_srd(p.x + (int64_t)b_first * x_elems, (unsigned long long)(p.n_samples - b_first) * x_elems * 4ull);
    const __amdgpu_buffer_rsrc_t srd_w = make_srd(p.ln_w, (unsigned long long)p.in_elems * 4ull);
    const __amdgpu_buffer_rsrc_t srd_lb = make_srd(p.ln_b, (unsigned long long)p.in_elems * 4ull);
    const __amdgpu_buffer_rsrc_t srd_b = make_srd(p.w, (unsigned long long)p.N * 4ull * p.Ci * 4ull);
    const __amdgpu_buffer_rsrc_t srd_st = make_srd(p.in_stats, (unsigned long long)p.n_samples * 8ull);
    // the five input positions of this thread's pair: byte offsets of the activation (relative to sample b_first) and of
    // the LayerNorm affine rows (sample-relative); out-of-range positions (zero padding, rows >= M) read 0 everywhere
    // positions 0 .. 3 of a pair are in range whenever its rows are (in_len >= 2 * out_len: launcher), position 4 is the
    // right / bottom zero pad for the last pair of a line: two offsets per operand, the position's j * tap_stride rides in
    // the loads' scalar offset (not part of the range check: an out-of-range voffset stays out of range)
    unsigned va[FIRST ? 1 : 2], vr[2];
    const int ts4 = (int)p.tap_stride * 4;
    float amu, ars;
    float mlv[2] = {0.f, 0.f};       // FIRST: this thread's two of the pair's 5 x 3 log-mel values, on their way to s_ml
    {
        const int m = m0 + pair_row0(rowq);
        const bool mok = m < p.M;
        const int b = m >> p.rps_shift;
        amu = buf_load1(srd_st, mok ? (unsigned)b * 8u : BUF_OOB);
        ars = buf_load1(srd_st, mok ? (unsigned)b * 8u + 4u : BUF_OOB);
        const int r = m & (rps - 1);
        const int fo = r >> p.To_shift, to = r & (p.To - 1);
        int rel, a0;
        if (p.axis == 0) { a0 = 2 * to; rel = (fo * p.T + a0) * p.Ci; }
        else { a0 = 2 * fo; rel = (a0 * p.T + to) * p.Ci; }
        const int aoff = (b - b_first) * (int)p.in_elems + rel;
        const bool ok4 = mok && (a0 + 4) < p.in_len;
        if (!FIRST) { va[0] = mok ? (unsigned)(aoff + col4 * 4) * 4u : BUF_OOB; va[1] = ok4 ? va[0] : BUF_OOB; }
        vr[0] = mok ? (unsigned)(rel + col4 * 4) * 4u : BUF_OOB;
        vr[1] = ok4 ? vr[0] : BUF_OOB;
        if (FIRST) {
            va[0] = 0;
            // the pair's 5 x 3 log-mel values (they do not depend on the channel) -> LDS, two per loader thread:
            // mel element ((b - b_first) * F + a0 + j) * T0 + to * s1 - pad1 + t1   (axis 1: a0 = 2 fo)
            const int tq = to * p.s1 - p.pad1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = col4 * 2 + h;              // 0 .. 15; 15 is padding
                const int j = e / 3, t1 = e - j * 3;
                const bool ok = e < 15 && mok && (a0 + j) < p.in_len && (unsigned)(tq + t1) < (unsigned)p.T0;
                mlv[h] = buf_load1(srd_a, ok ? (unsigned)(((b - b_first) * p.F + a0 + j) * p.T0 + tq + t1) * 4u : BUF_OOB);
            }
        }
    }
    unsigned vb[2];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // LDS chunk (wave * 2 + j) * 64 + lane = (row r, slot cs) fetches chunk cs ^ key(r) of weight row n0 + r (N % 128 == 0)
        const int ci = (wave * 2 + j) * 64 + lane, r = ci >> 3, cs = ci & 7;
        vb[j] = (unsigned)(n0 + r) * (unsigned)(4 * p.Ci) * 4u + (unsigned)((cs ^ ((r >> 1) & 7)) * 16);
    }
    f32x4 px[2], pw[2], pb[2], fe1;
    const f32x2 mu2 = {amu, amu}, rs2 = {ars, ars};
    // v = POST((z - mean) * rstd * W + B); out-of-range positions: W = B = 0 -> +-0 -> 0 (as in conv_gemm_ln_kernel)
    auto fx = [&](const f32x4 &z4, const f32x4 &w4, const f32x4 &b4) {
        f32x4 v;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 z = {z4[2 * h], z4[2 * h + 1]}, w2 = {w4[2 * h], w4[2 * h + 1]}, b2 = {b4[2 * h], b4[2 * h + 1]};
            f32x2 t = __builtin_elementwise_fma((z - mu2) * rs2, w2, b2);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (RELU_BN) t[e] = fmaxf(t[e], 0.f);
                else t[e] = p.after_bn ? act_fn(t[e], p.act) : t[e];
                v[2 * h + e] = t[e];
            }
        }
        return v;
    };
    int cc_st = 0;                                       // channel chunk of the sub-step whose operands sit in the registers
    // positions of sub-step KIND's operands: {o0, o1}, {e0, e1}, {e2}
    bool in_loop = false;
    auto issue = [&](auto kindc, int cc, float *Bd) {    // global loads of sub-step (kind, channel chunk cc)
        constexpr int KIND = decltype(kindc)::value;
        constexpr int J0 = KIND == 0 ? 1 : (KIND == 1 ? 0 : 4), J1 = KIND == 0 ? 3 : 2;
        const int so = cc * 4;
        cc_st = cc;
        // the weight tile first (the buffer form of the LDS load: one offset register per chunk, and vmcnt stays in order
        // beside the other buffer loads -- the global_load_lds form is a FLAT instruction, after which the compiler waits
        // for vmcnt(0) at every use of a loaded register)
#ifndef PFANN_ABL
#define PFANN_ABL 0
#endif
        // (PFANN_ABL: timing-only ablation builds, results garbage -- 1: no row requests in the K loop, 2: no weight-tile requests,
        // 4: no LDS refill; tools/ubench only)
        if (!((PFANN_ABL & 2) && in_loop))
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(srd_b, (__attribute__((address_space(3))) void *)&Bd[(wave_u * 2 + j) * 256], 16,
                                                     (int)vb[j], (KIND * p.Ci + cc) * 4, 0, 0);
        if ((PFANN_ABL & 1) && in_loop) return;
        if (KIND <= 2) {
            const int s0 = so + J0 * ts4;
            if (!FIRST) px[0] = buf_load4(srd_a, va[FIRST ? 0 : (J0 == 4)], s0);
            pw[0] = buf_load4(srd_w, vr[J0 == 4], s0); pb[0] = buf_load4(srd_lb, vr[J0 == 4], s0);
        }
        if (KIND <= 1) {
            const int s1 = so + J1 * ts4;
            if (!FIRST) px[1] = buf_load4(srd_a, va[0], s1);
            pw[1] = buf_load4(srd_w, vr[0], s1); pb[1] = buf_load4(srd_lb, vr[0], s1);
        }
    };
    // FIRST: z = b1 + sum_t1 w1[t1] * mel[t1] for this thread's four channels of input row J (same FMA order as
    // conv_first_stats_kernel: bias, then taps 0, 1, 2)
    auto first_z = [&](int J) {
        const int c0 = cc_st + col4 * 4;
        f32x4 z = *reinterpret_cast<const f32x4 *>(&s_w1[3 * p.Ci + c0]);
#pragma unroll
        for (int t1 = 0; t1 < 3; ++t1) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(&s_w1[t1 * p.Ci + c0]);
            const float m = s_ml[rowq * 16 + J * 3 + t1];
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = fmaf(m, w[e], z[e]);
        }
        if (!RELU_BN && !p.after_bn) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = act_fn(z[e], p.act);      // PRE of sub-layer 0
        }
        return z;
    };
    auto stash = [&](auto kindc, float *Ad) {             // transform + LDS refill (activation operand) of that sub-step
        constexpr int KIND = decltype(kindc)::value;
        if ((PFANN_ABL & 4) && in_loop) return;
        if (FIRST) {
            if (KIND == 0) { px[0] = first_z(1); px[1] = first_z(3); }
            else if (KIND == 1) { px[0] = first_z(0); px[1] = first_z(2); }
            else if (KIND == 2) px[0] = first_z(4);
        }
        if (KIND == 0) {
            *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = fx(px[0], pw[0], pb[0]);
            *reinterpret_cast<f32x4 *>(&Ad[(NP + rowq) * LDK + col4 * 4]) = fx(px[1], pw[1], pb[1]);
        } else if (KIND == 1) {
            const f32x4 f0 = fx(px[0], pw[0], pb[0]);
            fe1 = fx(px[1], pw[1], pb[1]);
            *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = f0 - fe1;
        } else if (KIND == 2) {
            *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = fe1 - fx(px[0], pw[0], pb[0]);
        } else {
            *reinterpret_cast<f32x4 *>(&Ad[rowq * LDK + col4 * 4]) = fe1;
        }
    };
    f32x16 acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    const int nch = p.Ci / BK;
    const int l31 = lane & 31, lhalf = lane >> 5;
    issue(K0{}, 0, s_b0);
    if (FIRST) {
        // the log-mel values reach LDS only now, behind the first sub-step's requests: their round trip and that of the
        // weight tile / LayerNorm rows overlap (round 6; the tile trace showed the folded-first prologue at 8.4 us against
        // 4.2 us for the other layers: two memory round trips one after the other)
        s_ml[rowq * 16 + col4 * 2] = mlv[0];
        s_ml[rowq * 16 + col4 * 2 + 1] = mlv[1];
        __syncthreads();
    }
    stash(K0{}, As);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weight tile has landed in LDS (the barrier's fence does not wait for it)
    __syncthreads();
    TILE_STAMP(ts_loop);
    in_loop = PFANN_ABL != 0;
    const int bfrag = (wn * 32 + l31) * BK + ((lhalf ^ ((l31 >> 1) & 7)) * 4);   // this lane's weight chunk of K step 0; step kk: ^ 8 kk
#ifdef PFANN_TILE_TRACE
    // K-loop phases of this wave, shader cycles: [0] barrier release -> stash (fragment reads, requests, MFMAs of kk 0..2),
    // [1] stash (LayerNorm transform + LDS refill, incl. its waits for the requested rows), [2] MFMAs of kk 3,
    // [3] wait for the weight tile (vmcnt), [4] barrier, [5] wait for the requested rows in front of the stash
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
#define PH_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); ph[i] += now_ - ph_last; ph_last = now_; } while (0)
#else
#define PH_STAMP(i)
#endif
    // one sub-step: MFMAs on LDS buffer PB while the next sub-step goes global -> registers -> buffer PB ^ 1
    auto substep = [&](int ch, auto kindc, auto nextc) {
        constexpr int KIND = decltype(kindc)::value, PB = KIND & 1;
        const float *Ac = As + PB * (BM * LDK), *Bc = PB ? s_b1 : s_b0;
        float *An = As + (PB ^ 1) * (BM * LDK), *Bn = PB ? s_b0 : s_b1;
        const bool more = KIND < 3 || ch + 1 < nch;
        const int cc_next = (KIND < 3 ? ch : ch + 1) * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            f32x4 a1 = a0;
            if (KIND == 0) a1 = *reinterpret_cast<const f32x4 *>(&Ac[(NP + wm * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(&Bc[bfrag ^ (kk * 8)]);
#ifdef PFANN_W22_ISSUE_FIRST                    // (rounds 2-5, kept for A/B: the requests in FRONT of the first MFMA group)
            if (kk == 0) {
                if (more) issue(nextc, cc_next, Bn);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            if (kk == BK / 8 - 1) {
                __builtin_amdgcn_sched_barrier(0);
                PH_STAMP(0);
#ifdef PFANN_TILE_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (trace builds: the wait for the requested rows on its own)
                PH_STAMP(5);
#endif
                if (more) stash(nextc, An);
                PH_STAMP(1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (KIND == 0) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[s], a0[s], acc[0], 0, 0, 0);   // C^T: rows = n
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[s], a1[s], acc[2], 0, 0, 0);
                } else if (KIND == 1) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[s], a0[s], acc[0], 0, 0, 0);
                else if (KIND == 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[s], a0[s], acc[2], 0, 0, 0);
                else acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[s], a0[s], acc[1], 0, 0, 0);
            }
#ifndef PFANN_W22_ISSUE_FIRST
            // The next sub-step's requests BEHIND the first MFMA group (round 6): issued in front of it, all eight waves spent
            // their first 200-400 cycles after every barrier on 4-8 memory instructions each while the MFMA pipe waited.
            // A/B/A/B on one box: all conv GEMMs -0.5 .. -0.9 % (profiles/r6/w22_issue_late_abc.txt); interleaving them
            // one by one between the MFMAs (sched_group_barrier) measured +0.7 %.  Same MFMA order: bit-identical results.
            if (kk == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue(nextc, cc_next, Bn);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        PH_STAMP(2);
        if (more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next weight tile has landed in LDS before anyone reads it
        PH_STAMP(3);
        __syncthreads();
        PH_STAMP(4);
    };
    for (int ch = 0; ch < nch; ++ch) {
        substep(ch, K0{}, K1{});
        substep(ch, K1{}, K2{});
        substep(ch, K2{}, K3{});
        substep(ch, K3{}, K0{});
    }

    TILE_STAMP(ts_loop_end);
    // ---- epilogue: y0 = A0 + A1 + bias, y1 = A1 + A2 + bias -> LDS -> whole rows; per-sample partial statistics
    // (same scheme as conv_gemm_ln_kernel: a lane holds ONE pair (lane & 31) and four consecutive channels per register quad)
    const __amdgpu_buffer_rsrc_t srd_y = make_srd(p.y + (int64_t)m0 * p.N, (unsigned long long)(p.M - m0) * p.N * 4ull);
    const __amdgpu_buffer_rsrc_t srd_bias = make_srd(p.bias, (unsigned long long)p.N * 4ull);
    constexpr int LDC = BN + 4;
    // the C tile [BM][LDC] lies over the activation buffers and both weight buffers: the three objects must be one contiguous
    // range (their LDS addresses are constants after lowering: the test folds away; a layout that breaks it traps)
    const unsigned a0_ = (unsigned)(size_t)(__attribute__((address_space(3))) float *)smem;
    const unsigned b0_ = (unsigned)(size_t)(__attribute__((address_space(3))) float *)s_b0;
    const unsigned b1_ = (unsigned)(size_t)(__attribute__((address_space(3))) float *)s_b1;
    const unsigned lo_ = min(a0_, min(b0_, b1_));
    const unsigned hi_ = max(a0_ + (unsigned)(2 * BM * LDK * 4), max(b0_ + (unsigned)(BN * BK * 4), b1_ + (unsigned)((BN * BK + 640) * 4)));
    if (hi_ - lo_ != (unsigned)((2 * BM * LDK + 2 * BN * BK + 640) * 4)) __builtin_trap();
    float *Cs = (float *)(__attribute__((address_space(3))) float *)(size_t)lo_;
    float *red1 = Cs + BM * LDC;                        // [BM][WAVES_N]
    float *red2 = red1 + BM * WAVES_N;
    const int G = rps >= BM ? BM : rps;
    const unsigned rowbytes = (unsigned)p.N * 4u;
    f32x4 bias4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 32 + 8 * g + 4 * lhalf;
        bias4[g] = buf_load4(srd_bias, n < p.N ? (unsigned)n * 4u : BUF_OOB);
    }
    const int prow = pair_row0(wm * 32 + l31);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float a1 = 0.f, a2 = 0.f;
        const int row = prow + i * prow_stride;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 z4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float z = (acc[i][4 * g + e] + acc[i + 1][4 * g + e]) + bias4[g][e];
                if (!RELU_BN && !p.after_bn) z = act_fn(z, p.act);
                a1 += z;
                a2 = fmaf(z, z, a2);
                z4[e] = z;
            }
            *reinterpret_cast<f32x4 *>(&Cs[row * LDC + wn * 32 + 8 * g + 4 * lhalf]) = z4;
        }
        a1 += __shfl_xor(a1, 32, 64);
        a2 += __shfl_xor(a2, 32, 64);
        if (lhalf == 0) {
            red1[row * WAVES_N + wn] = a1;
            red2[row * WAVES_N + wn] = a2;
        }
    }
    __syncthreads();
    {
        constexpr int LPR = BN / 4, RPP = NT / LPR;
        const int cl = tid % LPR, rr = tid / LPR;
        const int n = n0 + cl * 4;
        const unsigned nb = n < p.N ? (unsigned)n * 4u : BUF_OOB;
#pragma unroll
        for (int it = 0; it < BM / RPP; ++it) {
            const int row = it * RPP + rr;
            buf_store4(srd_y, nb + (unsigned)row * rowbytes, *reinterpret_cast<const f32x4 *>(&Cs[row * LDC + cl * 4]));
        }
    }
    float t1 = 0.f, t2 = 0.f;
    if (tid < BM) {
#pragma unroll
        for (int w = 0; w < WAVES_N; ++w) { t1 += red1[tid * WAVES_N + w]; t2 += red2[tid * WAVES_N + w]; }
        for (int o = (G < 64 ? G : 64) >> 1; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
    }
    if (G > 64) {
        float *x2 = red2 + BM * WAVES_N;
        if (tid < BM && lane == 0) { x2[2 * wave] = t1; x2[2 * wave + 1] = t2; }
        __syncthreads();
        t1 = x2[0] + x2[2];
        t2 = x2[1] + x2[3];
    }
    if (tid < BM && (tid & (G - 1)) == 0) {
        const int mg = m0 + tid;
        if (mg < p.M) {
            const int b = mg >> p.rps_shift;
            const int slot = ((mg & (rps - 1)) / G) * p.n_tiles_n + nt;
            float *o = p.out_part + ((int64_t)b * p.out_P + slot) * 2;
            o[0] = t1;
            o[1] = t2;
        }
    }
#ifdef PFANN_TILE_TRACE
    if (tid == 0 && g_tile_trace != nullptr && blockIdx.x < g_tile_trace_cap && (g_tile_trace_rps == 0 || g_tile_trace_rps == rps)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (this wave's output stores have been acknowledged)
        unsigned long long *o = g_tile_trace + (size_t)blockIdx.x * 12;
        o[0] = ts_start; o[1] = ts_loop; o[2] = ts_loop_end; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);         // HW_REG_XCC_ID
        for (int i = 0; i < 6; ++i) o[6 + i] = ph[i];              // (wave 0's K-loop phases)
    }
#endif
}

#ifdef PFANN_TILE_TRACE
extern "C" int pfann_debug_set_tile_trace(void *buf, unsigned cap_blocks, int rows_per_sample) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_rps), &rows_per_sample, sizeof(rows_per_sample)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_cap), &cap_blocks, sizeof(cap_blocks)) != hipSuccess) return -1;
    return 0;
}
#endif

// (mean, rstd) of every sample from its P partial (sum, sum of squares) pairs: one wave per sample,
// fp64, fixed order.  Done once here rather than by every GEMM block that touches the sample.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float *__restrict__ part, int P, double inv_n,
                                                          float *__restrict__ stats, int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float *pp = part + (int64_t)b * P * 2;
    double s1 = 0, s2 = 0;
    for (int i = lane; i < P; i += 64) { s1 += (double)pp[2 * i]; s2 += (double)pp[2 * i + 1]; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane == 0) {
        const double mean = s1 * inv_n;
        double var = s2 * inv_n - mean * mean;
        if (var < 0) var = 0;
        stats[2 * b] = (float)mean;
        stats[2 * b + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// Second half of a SPLITK layer: one workgroup per sample.  z = PRE(bias + sum_s partial[s]) in the order s = 0, 1, ...
// (deterministic), written once; the sample's LayerNorm partial (sum, sum of squares) reduced in a fixed order into slot 0
// of its partial row (the other slots zeroed), which is what the consumer's ln_finalize_kernel expects.
__global__ __launch_bounds__(1024) void splitk_reduce_ln_kernel(const float *__restrict__ part_c, int n_splits, int64_t MN,
                                                                const float *__restrict__ bias, int N, int n_el,
                                                                float *__restrict__ y, float *__restrict__ out_part, int P,
                                                                int act, int after_bn, float *__restrict__ stats, double inv_n) {
    __shared__ float red[2][16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)b * n_el;
    float s1 = 0.f, s2 = 0.f;
    for (int e = tid * 4; e < n_el; e += 4096) {          // N % 4 == 0: a float4 never straddles two rows
        f32x4 z = *reinterpret_cast<const f32x4 *>(bias + (e % N));
        for (int s = 0; s < n_splits; ++s) z += *reinterpret_cast<const f32x4 *>(part_c + (int64_t)s * MN + base + e);
        if (!after_bn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = act_fn(z[q], act);
        }
        *reinterpret_cast<f32x4 *>(y + base + e) = z;
        s1 += (z[0] + z[1]) + (z[2] + z[3]);
        s2 += (z[0] * z[0] + z[1] * z[1]) + (z[2] * z[2] + z[3] * z[3]);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid == 0) {
        float t1 = 0.f, t2 = 0.f;
        for (int w = 0; w < 16; ++w) { t1 += red[0][w]; t2 += red[1][w]; }
        float *o = out_part + (int64_t)b * P * 2;
        o[0] = t1; o[1] = t2;
        for (int i = 1; i < P; ++i) { o[2 * i] = 0.f; o[2 * i + 1] = 0.f; }
        // ... and (mean, rstd) exactly as ln_finalize_kernel would derive them from that partial row, so the consumer
        // can skip its ln_finalize launch
        const double mean = (double)t1 * inv_n;
        double var = (double)t2 * inv_n - mean * mean;
        if (var < 0) var = 0;
        stats[2 * b] = (float)mean;
        stats[2 * b + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

static void live_taps(const SubLayer &L, int in_len, int &k_begin, int &k_end) {
    const int out_len = L.axis == 0 ? L.To : L.Fo;
    int lo = 3, hi = -1;
    for (int tap = 0; tap < 3; ++tap)
        for (int o = 0; o < out_len; ++o) {
            const int pos = o * L.stride - L.pad_lo + tap;
            if (pos >= 0 && pos < in_len) { lo = tap < lo ? tap : lo; hi = tap > hi ? tap : hi; break; }
        }
    k_begin = lo * L.ci;
    k_end = (hi + 1) * L.ci;
}

// tile size used for sub-layer L at batch B (shared by the launcher and the partial-slot planner)
static int gemm_tile(const SubLayer &L, int64_t B) {
    const int64_t M = B * L.Fo * L.To;
    const int64_t blocks128 = (int64_t)cdiv(M, 128) * cdiv(L.co, 128);
    static const int64_t min128 = getenv("PFANN_TILE128_MIN") ? atoll(getenv("PFANN_TILE128_MIN")) : 512;
    return (L.co >= 128 && blocks128 >= min128 && L.ci % 32 == 0) ? 128 : 64;
}

int fused_out_slots(const SubLayer &L, int64_t B) {
    const int rps = L.Fo * L.To;
    if (L.ci == 1 || L.depthwise) return rps / 64 > 0 ? rps / 64 : 1;       // conv_first_stats / conv_dw_ln: 64 rows per block
    const int bt = gemm_tile(L, B);
    return (rps >= bt ? rps / bt : 1) * cdiv(L.co, bt);
}

bool fused_supported(const SubLayer *sub, int n) {
    for (int i = 0; i < n; ++i) {
        const SubLayer &L = sub[i];
        const int rps = L.Fo * L.To;
        if (rps & (rps - 1)) return false;                  // power of two: tiles never straddle samples unevenly
        if (L.depthwise) { if (L.co % 4 || L.axis != 1) return false; continue; }     // conv_dw_ln_kernel
        if (L.ci == 1) { if (i != 0 || rps % 64 || L.co % 4) return false; }
        else if (L.ci % 4 || L.co % 4) return false;
    }
    return true;
}

// Split-K plan of sub-layer L (64x64 tiles only): number of K chunks, 0 = not split.  The cut is chunks of 8 K-tiles whatever
// the batch -- it depends on the layer only -- so a window's bits do not depend on how many windows share the launch;
// WHETHER a layer is split depends on the batch the plan is made for:
//   * plans of at most 64 windows (the one-query regime; pfann_set_plan_batch never pins one): launches of < 192 tiles;
//   * larger plans (round 6: the middle of the batch curve, 65 .. ~1000 windows): the layers whose launch at the PLAN's
//     batch is < 512 tiles of a chip that holds ~1000 -- the deep layers, M = B * (1 .. 8) rows with K loops of 48-96
//     K-tiles: at 76 windows `rows=2 K=3072` was 48 workgroups x 96 K-tiles, 71 us on a sixth of the chip.
// The same predicate sizes the scratch (splitk_scratch_need) and picks the kernel (launch_conv_gemm_ln).
static int splitk_plan(const SubLayer &L, int64_t B, int64_t Bp, bool first, int k_live) {
    static const int chunk_kt = getenv("PFANN_SPLITK_CHUNK") ? atoi(getenv("PFANN_SPLITK_CHUNK")) : 8;
    static const bool no_splitk = getenv("PFANN_NO_SPLITK") != nullptr;
    static const int64_t mid_max = getenv("PFANN_SPLITK_MID_BLOCKS") ? atoll(getenv("PFANN_SPLITK_MID_BLOCKS")) : 512;
    if (no_splitk || first || chunk_kt <= 0 || L.ci % 32 != 0 || k_live % 32 != 0) return 0;
    const int nk = k_live / 32;
    const int n_splits = (nk + chunk_kt - 1) / chunk_kt;
    if (nk < 16 || n_splits <= 1) return 0;
    const int64_t rps = (int64_t)L.Fo * L.To, ntn = cdiv(L.co, 64);
    const int64_t blocks = (int64_t)cdiv(B * rps, 64) * ntn, blocks_p = (int64_t)cdiv(Bp * rps, 64) * ntn;
    const bool small = B <= 64 && Bp <= 64;
    if (small ? blocks >= 192 : (B > Bp || blocks_p >= mid_max)) return 0;
    return n_splits;
}

size_t splitk_scratch_need(const SubLayer &L, int64_t B, int64_t Bp) {
    if (L.depthwise || L.ci == 1 || gemm_tile(L, Bp) != 64) return 0;
    int kb, ke;
    live_taps(L, L.axis == 0 ? L.T : L.F, kb, ke);
    const int n_splits = splitk_plan(L, B, Bp, false, ke - kb);
    return (size_t)n_splits * (size_t)B * L.Fo * L.To * L.co * sizeof(float);
}

// Lfirst != nullptr: x is the log-mel batch and Lfirst (= Lin, the C_in = 1 conv) is folded into the A-loader
static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

int launch_conv_gemm_ln(const SubLayer &L, const SubLayer &Lin, const float *x, const float *in_part, int in_P,
                        float *in_stats, float *y, float *out_part, int64_t B, int act, int after_bn,
                        const SubLayer *Lfirst, int precision, hipStream_t s, float *splitk_scratch, size_t splitk_bytes,
                        int *stats_final, int64_t Bplan) {
    const int64_t Bp = Bplan > 0 ? Bplan : B;      // the batch the kernel variant is chosen for (kernels.h)
    // stats_final (in/out, may be null): in: in_stats already holds (mean, rstd) of the input (written by the previous
    // layer's split-K reduction) -> no ln_finalize launch; out: whether this layer left its OUTPUT statistics there.
    const bool in_final = stats_final != nullptr && *stats_final != 0;
    if (stats_final) *stats_final = 0;
    FusedGemmParams p;
    p.blocks_mn = 0; p.k_chunk = 0;
    if (B * L.Fo * L.To >= (int64_t)0x7FFF0000) { set_error("conv_gemm_ln: batch too large"); return -1; }
    p.x = x; p.w = L.w; p.bias = L.bias; p.y = y;
    p.w_hi = L.w_hi; p.w_lo = L.w_lo; p.w_inv_scale = L.w_inv_scale;
    p.rows_per_sample = L.Fo * L.To;
    p.M = (int)(B * p.rows_per_sample);
    p.rps_shift = ilog2(p.rows_per_sample); p.To_shift = ilog2(L.To);
    p.N = L.co; p.Ci = L.ci; p.K = 3 * L.ci;
    p.To = L.To; p.F = L.F; p.T = L.T;
    p.axis = L.axis; p.stride = L.stride; p.pad_lo = L.pad_lo;
    p.in_len = L.axis == 0 ? L.T : L.F;
    p.tap_stride = L.axis == 0 ? (int64_t)L.ci : (int64_t)L.T * L.ci;
    live_taps(L, p.in_len, p.k_begin, p.k_end);
    p.in_stats = in_stats;
    p.ln_w = Lin.ln_w; p.ln_b = Lin.ln_b;
    p.in_elems = (int64_t)L.F * L.T * L.ci;
    p.out_part = out_part; p.out_P = fused_out_slots(L, Bp);
    p.act = act; p.after_bn = after_bn;
    p.n_samples = (int)B;
    p.w1 = nullptr; p.b1 = nullptr; p.T0 = 0; p.s1 = 1; p.pad1 = 0;
    if (Lfirst != nullptr) {
        p.w1 = Lfirst->w; p.b1 = Lfirst->bias;
        p.T0 = Lfirst->T; p.s1 = Lfirst->stride; p.pad1 = Lfirst->pad_lo;
    }
    if (!in_final && !(PFANN_ABL & 8)) {        // (PFANN_ABL & 8: timing-only build without the ln_finalize launches)
        ProfScope ps("ln_finalize", s);
        PF_LAUNCH(ln_finalize_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, s, in_part, in_P, 1.0 / (double)p.in_elems,
                  in_stats, (int)B);
    }
    const double flops = 2.0 * (double)p.M * p.N * (p.k_end - p.k_begin);
    const bool first = Lfirst != nullptr, relu_bn = act == 0 && after_bn, uni = L.ci % 32 == 0;
    // PFANN_PROF_LAYERS=1: one profiling tag per layer shape (tuning aid; bench.py's roofline wants the plain tags)
    static const bool per_layer = getenv("PFANN_PROF_LAYERS") != nullptr;
    auto layer_tag = [&](const char *base) -> const char * {
        if (!per_layer) return base;
        static std::set<std::string> names;
        char buf[128];
        snprintf(buf, sizeof buf, "%s rows=%d K=%d N=%d%s", base, p.rows_per_sample, p.k_end - p.k_begin, p.N, first ? " first" : "");
        return names.insert(buf).first->c_str();
    };
#define PF_GEMM_LN(BM, BN, WM, WN, NTHR)                                                                  \
    do {                                                                                                  \
        const dim3 g((unsigned)blocks), t(NTHR);                                                          \
        if (first && relu_bn) PF_LAUNCH((conv_gemm_ln_kernel<BM, BN, WM, WN, true, true, UNI_>), g, t, 0, s, p);   \
        else if (first) PF_LAUNCH((conv_gemm_ln_kernel<BM, BN, WM, WN, false, true, UNI_>), g, t, 0, s, p);        \
        else if (relu_bn) PF_LAUNCH((conv_gemm_ln_kernel<BM, BN, WM, WN, true, false, UNI_>), g, t, 0, s, p);      \
        else PF_LAUNCH((conv_gemm_ln_kernel<BM, BN, WM, WN, false, false, UNI_>), g, t, 0, s, p);                  \
    } while (0)
    if (gemm_tile(L, Bp) == 128) {
        p.n_tiles_n = cdiv(p.N, 128);
        p.dv_n = make_fastdiv(p.n_tiles_n); p.dv_tile = make_fastdiv(64 * p.n_tiles_n);
        p.dv_group = make_fastdiv(64 * (p.rows_per_sample >= 128 ? p.rows_per_sample / 128 : 1) * p.n_tiles_n);
        const int64_t blocks = (int64_t)cdiv(p.M, 128) * p.n_tiles_n;
        // five channel blocks per output pair instead of six (conv_gemm_ln_w22_kernel) where the layer allows it
        static const bool no_w22 = getenv("PFANN_NO_W22") != nullptr;
        const int out_len = L.axis == 0 ? L.To : L.Fo;
        if (!no_w22 && (!first || (L.axis == 1 && L.ci <= 256 && getenv("PFANN_NO_W22_FIRST") == nullptr)) && precision == 0 && L.w22 != nullptr && L.stride == 2 && L.pad_lo == 0 && p.k_begin == 0 &&
            p.k_end == 3 * L.ci && out_len % 2 == 0 && p.in_len >= 2 * out_len && p.M % 2 == 0 &&
            p.N % 128 == 0 && (L.axis == 0 || (L.To <= 64 && 128 % (2 * L.To) == 0)) &&
            (int64_t)p.N * 4 * p.Ci * 4 < 0x7FFF0000ll) {
            p.w = L.w22;
            ProfScope ps(layer_tag(per_layer ? "conv_gemm_ln_128 w22" : "conv_gemm_ln_128"), s, flops);
            const dim3 g((unsigned)blocks), t(512);
            if (first && relu_bn) PF_LAUNCH((conv_gemm_ln_w22_kernel<true, true>), g, t, 0, s, p);
            else if (first) PF_LAUNCH((conv_gemm_ln_w22_kernel<false, true>), g, t, 0, s, p);
            else if (relu_bn) PF_LAUNCH((conv_gemm_ln_w22_kernel<true, false>), g, t, 0, s, p);
            else PF_LAUNCH((conv_gemm_ln_w22_kernel<false, false>), g, t, 0, s, p);
            PF_HIP(hipGetLastError());
            return 0;
        }
        ProfScope ps(layer_tag("conv_gemm_ln_128"), s, flops);
        // 8 waves (512 threads), each a 64x32 tile: half the prefetch registers per thread and four
        // waves per SIMD with two resident blocks
        constexpr bool UNI_ = true;
        if (precision == 1 && relu_bn && L.w_hi != nullptr) {
            const dim3 g((unsigned)blocks), t(512);
            if (first) PF_LAUNCH((conv_gemm_ln_kernel<128, 128, 64, 32, true, true, true, 32, true>), g, t, 0, s, p);
            else PF_LAUNCH((conv_gemm_ln_kernel<128, 128, 64, 32, true, false, true, 32, true>), g, t, 0, s, p);
        } else
        PF_GEMM_LN(128, 128, 64, 32, 512);
    } else {
        p.n_tiles_n = cdiv(p.N, 64);
        p.dv_n = make_fastdiv(p.n_tiles_n); p.dv_tile = make_fastdiv(64 * p.n_tiles_n);
        p.dv_group = make_fastdiv(64 * (p.rows_per_sample >= 64 ? p.rows_per_sample / 64 : 1) * p.n_tiles_n);
        const int64_t blocks = (int64_t)cdiv(p.M, 64) * p.n_tiles_n;
        // split-K for launches that cannot fill the chip and have a long K loop (splitk_plan above): chunks of 8 K-tiles
        // (measured on one 19-segment query: 12 -> 344 us, 8 -> 323, 6 -> 323, 4 -> 331 for the whole embed)
        static const int chunk_kt = getenv("PFANN_SPLITK_CHUNK") ? atoi(getenv("PFANN_SPLITK_CHUNK")) : 8;
        const int n_splits = uni ? splitk_plan(L, B, Bp, first, p.k_end - p.k_begin) : 0;
        if (n_splits > 1 && splitk_scratch != nullptr && (size_t)n_splits * p.M * p.N * sizeof(float) <= splitk_bytes) {
            p.blocks_mn = (int)blocks;
            p.k_chunk = chunk_kt * 32;
            p.y = splitk_scratch;
            {
                ProfScope ps(layer_tag("conv_gemm_ln_64"), s, flops);
                const dim3 g((unsigned)(blocks * n_splits)), t(256);
                if (relu_bn) PF_LAUNCH((conv_gemm_ln_kernel<64, 64, 32, 32, true, false, true, 32, false, true>), g, t, 0, s, p);
                else PF_LAUNCH((conv_gemm_ln_kernel<64, 64, 32, 32, false, false, true, 32, false, true>), g, t, 0, s, p);
            }
            ProfScope ps2("splitk_reduce_ln", s);
            PF_LAUNCH(splitk_reduce_ln_kernel, dim3((unsigned)B), dim3(1024), 0, s, splitk_scratch, n_splits, (int64_t)p.M * p.N,
                      L.bias, p.N, p.rows_per_sample * p.N, y, out_part, p.out_P, act, after_bn, in_stats,
                      1.0 / ((double)p.rows_per_sample * p.N));
            PF_HIP(hipGetLastError());
            if (stats_final) *stats_final = 1;
            return 0;
        }
        ProfScope ps(layer_tag("conv_gemm_ln_64"), s, flops);
        if (uni) {
            constexpr bool UNI_ = true;
            PF_GEMM_LN(64, 64, 32, 32, 256);
        } else {
            constexpr bool UNI_ = false;
            PF_GEMM_LN(64, 64, 32, 32, 256);
        }
    }
#undef PF_GEMM_LN
    return 0;
}

// ------------------------------------------------------------------------------------
// First conv (ci == 1) with statistics: one block = 64 output positions x all channels.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_stats_kernel(const float *__restrict__ x,
                                                               const float *__restrict__ w,
                                                               const float *__restrict__ bias,
                                                               float *__restrict__ y, float *__restrict__ part,
                                                               int64_t M, int co, int To, int T, int stride,
                                                               int pad_lo, int rps, int P, int act, int after_bn) {
    __shared__ float red[8];
    const int co4 = co >> 2;
    const int64_t mb = (int64_t)blockIdx.x * 64;          // 64 rows of ONE sample (rps % 64 == 0)
    const int tid = threadIdx.x;
    const int64_t b = mb / rps;
    const int r0 = (int)(mb - b * rps);                   // row inside the sample: 32-bit from here on
    const int F = rps / To;
    const float *xs = x + b * (int64_t)F * T;
    float *ys = y + mb * co;
    float s1 = 0.f, s2 = 0.f;
    const int n_it = (64 * co4 + 255) / 256;
    for (int it = 0; it < n_it; ++it) {
        const int e = tid + it * 256;
        const int rl = e / co4;
        if (rl >= 64 || mb + rl >= M) break;
        const int c = (e - rl * co4) * 4;
        const int r = r0 + rl;
        const int f = r / To, to = r - f * To;
        const int p0 = to * stride - pad_lo;
        f32x4 o = *reinterpret_cast<const f32x4 *>(bias + c);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int t = p0 + tap;
            if ((unsigned)t < (unsigned)T) {
                const float xv = xs[f * T + t];
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + tap * co + c);
                o += xv * wv;
            }
        }
        if (!after_bn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = act_fn(o[q], act);
        }
        if (y != nullptr) *reinterpret_cast<f32x4 *>(ys + rl * co + c) = o;
        s1 += (o[0] + o[1]) + (o[2] + o[3]);
        s2 += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
    __syncthreads();
    if (tid == 0 && mb < M) {
        const int slot = r0 / 64;
        float *o = part + (b * P + slot) * 2;
        o[0] = (red[0] + red[1]) + (red[2] + red[3]);
        o[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

int launch_conv_first_stats(const SubLayer &L, const float *x, float *y, float *part, int64_t B, int act,
                            int after_bn, hipStream_t s) {
    const int rps = L.Fo * L.To;
    const int64_t M = B * rps;
    ProfScope ps(y != nullptr ? "conv_first_stats" : "conv_first_stats_only", s,
                 y != nullptr ? 4.0 * ((double)M * L.co + (double)B * L.F * L.T) : 4.0 * (double)B * L.F * L.T);
    PF_LAUNCH(conv_first_stats_kernel, dim3((unsigned)cdiv(M, 64)), dim3(256), 0, s, x, L.w, L.bias, y,
                       part, M, L.co, L.To, L.T, L.stride, L.pad_lo, rps, fused_out_slots(L, B), act, after_bn);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// LayerNorm statistics of the first conv WITHOUT computing it (PRE = identity, i.e. ReLU after
// LayerNorm): with x = the three log-mel taps of an output position,
//   sum_c z_c   = Bsum + sum_t Wsum[t] x_t
//   sum_c z_c^2 = bb + 2 sum_t h[t] x_t + sum_{t,u} G[t][u] x_t x_u ,   G = W W^T (3x3), h = W b.
// One lane per output position, one wave per 64-row partial slot: a few flops per position
// instead of 3*Co FMAs.  gram = {Bsum, Wsum[3], bb, h[3], G00, G01, G02, G11, G12, G22}.
// ------------------------------------------------------------------------------------
struct GramParams { float v[14]; };

__global__ __launch_bounds__(256) void conv_first_gram_stats_kernel(const float *__restrict__ x, float *__restrict__ part,
                                                                    int64_t M, int To, int T, int stride, int pad_lo,
                                                                    int rps, int P, GramParams g, FastDiv dv_bps, FastDiv dv_to,
                                                                    int F) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float s1 = 0.f, s2 = 0.f;
    int64_t b = 0;
    int r = 0;
    if (m < M) {
        // sample and row of this position.  rps % 256 == 0 (dv_bps.d = workgroups per sample > 0): the sample is the
        // workgroup's, by one 32-bit multiply-high; the generic 64-bit division per thread was most of this kernel
        // (two emulated divisions against three loads and twenty flops)
        int f, to;
        if (dv_bps.d > 0) {
            const int bb = fastdiv((int)blockIdx.x, dv_bps);
            b = bb;
            r = ((int)blockIdx.x - bb * dv_bps.d) * 256 + (int)threadIdx.x;
            f = fastdiv(r, dv_to);
            to = r - f * To;
        } else {
            b = m / rps;
            r = (int)(m - b * rps);
            f = r / To; to = r - f * To;
        }
        const float *xs = x + (b * F + f) * (int64_t)T;
        float xt[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int ti = to * stride - pad_lo + t;
            xt[t] = (unsigned)ti < (unsigned)T ? xs[ti] : 0.f;
        }
        s1 = g.v[0] + g.v[1] * xt[0] + g.v[2] * xt[1] + g.v[3] * xt[2];
        s2 = g.v[4] + 2.f * (g.v[5] * xt[0] + g.v[6] * xt[1] + g.v[7] * xt[2]) +
             g.v[8] * xt[0] * xt[0] + g.v[11] * xt[1] * xt[1] + g.v[13] * xt[2] * xt[2] +
             2.f * (g.v[9] * xt[0] * xt[1] + g.v[10] * xt[0] * xt[2] + g.v[12] * xt[1] * xt[2]);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int64_t m_wave = m - lane;                      // first row of this wave's 64-row slot
    if (lane == 0 && m_wave < M) {
        const int64_t bw = dv_bps.d > 0 ? b : m_wave / rps;
        const int slot = dv_bps.d > 0 ? r / 64 : (int)((m_wave - bw * rps) / 64);
        float *o = part + (bw * P + slot) * 2;
        o[0] = s1;
        o[1] = s2;
    }
}

int launch_conv_first_gram_stats(const SubLayer &L, const float *x, float *part, int64_t B, const float *gram14,
                                 hipStream_t s) {
    const int rps = L.Fo * L.To;
    const int64_t M = B * rps;
    GramParams g;
    for (int i = 0; i < 14; ++i) g.v[i] = gram14[i];
    ProfScope ps("conv_first_gram_stats", s, 4.0 * (double)B * L.F * L.T);
    FastDiv dv_bps = make_fastdiv(1), dv_to = make_fastdiv(L.To);
    dv_bps.d = 0;                                         // (0: the generic index arithmetic)
    if (rps % 256 == 0 && cdiv(M, 256) < 0x7FFFFFFF) dv_bps = make_fastdiv(rps / 256);
    PF_LAUNCH(conv_first_gram_stats_kernel, dim3((unsigned)cdiv(M, 256)), dim3(256), 0, s, x, part, M, L.To, L.T,
              L.stride, L.pad_lo, rps, fused_out_slots(L, B), g, dv_bps, dv_to, L.Fo);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Depthwise 3x1 conv along F (conv2 of non-"fuller" models: configs/seg.json, n640d64.json) in the fused scheme:
// LayerNorm(+activation) of its INPUT applied on load -- v = POST((z - mean) * rstd * W + B) -- and the partial
// LayerNorm statistics of its OUTPUT produced by the same pass, so a depthwise model's activations are written once
// (raw) and read once, as in the full-conv models (the separate LayerNorm kernel made three more passes over every
// tensor: 46 % of the n640d64 encoder).  HBM-bound elementwise work: one workgroup = up to 64 output rows of ONE sample
// x all channels, float4 along C; statistics reduced in a fixed order.
// ------------------------------------------------------------------------------------
// FIRST: the input is the C_in = 1 first conv computed on the fly from the log-mel (x = mel [B][F][T0]): its
// 1 MiB/segment output (n640d64) is never written or read back, as in the full-conv models' folded first layer.
template <bool FIRST>
__global__ __launch_bounds__(256) void conv_dw_ln_kernel(const float *__restrict__ x, const float *__restrict__ in_stats,
                                                         const float *__restrict__ lw, const float *__restrict__ lb,
                                                         const float *__restrict__ w, const float *__restrict__ bias,
                                                         float *__restrict__ y, float *__restrict__ part, int C, int To,
                                                         int F, int T, int stride, int pad_lo, int rps, int P, int act,
                                                         int after_bn, const float *__restrict__ w1,
                                                         const float *__restrict__ b1, int T0, int st1, int pad1) {
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x / P;
    const int slot = (int)(blockIdx.x - b * P);
    const int rows = rps < 64 ? rps : 64;
    const int r0 = slot * rows;
    const int c4n = C >> 2;
    const float mean = in_stats[2 * b], rstd = in_stats[2 * b + 1];
    const float *xs = x + b * (int64_t)F * (FIRST ? T0 : T * C);
    float *ys = y + (b * rps + r0) * (int64_t)C;
    float s1 = 0.f, s2 = 0.f;
    for (int e = tid; e < rows * c4n; e += 256) {
        const int rl = e / c4n, c = (e - rl * c4n) * 4;
        const int r = r0 + rl;
        const int fo = r / To, to = r - fo * To;
        f32x4 o = *reinterpret_cast<const f32x4 *>(bias + c);
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
            const int f = fo * stride - pad_lo + tap;
            if ((unsigned)f < (unsigned)F) {
                const int64_t idx = ((int64_t)f * T + to) * C + c;
                f32x4 z;
                if (FIRST) {           // z = b1 + sum_t1 w1[t1] * mel[f][to*s1 - pad1 + t1]  (bias, then taps 0, 1, 2)
                    z = *reinterpret_cast<const f32x4 *>(b1 + c);
#pragma unroll
                    for (int t1 = 0; t1 < 3; ++t1) {
                        const int tq = to * st1 - pad1 + t1;
                        if ((unsigned)tq < (unsigned)T0) {
                            const float mv = xs[(int64_t)f * T0 + tq];
                            const f32x4 k1 = *reinterpret_cast<const f32x4 *>(w1 + t1 * C + c);
#pragma unroll
                            for (int q = 0; q < 4; ++q) z[q] = fmaf(mv, k1[q], z[q]);
                        }
                    }
                    if (!after_bn) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) z[q] = act_fn(z[q], act);
                    }
                } else {
                    z = *reinterpret_cast<const f32x4 *>(xs + idx);
                }
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(lw + idx), bv = *reinterpret_cast<const f32x4 *>(lb + idx);
                const f32x4 kv = *reinterpret_cast<const f32x4 *>(w + tap * C + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = fmaf((z[q] - mean) * rstd, wv[q], bv[q]);
                    if (after_bn) t = act_fn(t, act);
                    o[q] = fmaf(t, kv[q], o[q]);
                }
            }
        }
        if (!after_bn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = act_fn(o[q], act);
        }
        *reinterpret_cast<f32x4 *>(ys + (int64_t)rl * C + c) = o;
        s1 += (o[0] + o[1]) + (o[2] + o[3]);
        s2 += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
    __syncthreads();
    if (tid == 0) {
        float *o = part + (b * P + slot) * 2;
        o[0] = (red[0] + red[1]) + (red[2] + red[3]);
        o[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

int launch_conv_dw_ln(const SubLayer &L, const SubLayer &Lin, const float *x, const float *in_part, int in_P, float *in_stats,
                      float *y, float *out_part, int64_t B, int act, int after_bn, const SubLayer *Lfirst, hipStream_t s) {
    const int rps = L.Fo * L.To;
    const int64_t in_elems = (int64_t)L.F * L.T * L.ci;
    {
        ProfScope ps("ln_finalize", s);
        PF_LAUNCH(ln_finalize_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, s, in_part, in_P, 1.0 / (double)in_elems, in_stats, (int)B);
    }
    const int P = fused_out_slots(L, B);
    ProfScope ps("conv_dw_ln", s, 4.0 * (double)B * ((double)in_elems + (double)rps * L.co));
    if (Lfirst != nullptr)
        PF_LAUNCH(conv_dw_ln_kernel<true>, dim3((unsigned)(B * P)), dim3(256), 0, s, x, in_stats, Lin.ln_w, Lin.ln_b, L.w, L.bias, y,
                  out_part, L.co, L.To, L.F, L.T, L.stride, L.pad_lo, rps, P, act, after_bn, Lfirst->w, Lfirst->bias, Lfirst->T,
                  Lfirst->stride, Lfirst->pad_lo);
    else
        PF_LAUNCH(conv_dw_ln_kernel<false>, dim3((unsigned)(B * P)), dim3(256), 0, s, x, in_stats, Lin.ln_w, Lin.ln_b, L.w, L.bias, y,
                  out_part, L.co, L.To, L.F, L.T, L.stride, L.pad_lo, rps, P, act, after_bn, nullptr, nullptr, 0, 1, 0);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Materialise POST(LN(z)) from partials (verification taps; also the generic fallback).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_apply_kernel(const float *__restrict__ z, const float *__restrict__ part,
                                                       int P, const float *__restrict__ w,
                                                       const float *__restrict__ b, float *__restrict__ out, int n,
                                                       int act, int after_bn) {
    __shared__ double red[8];
    __shared__ float stat[2];
    const int tid = threadIdx.x;
    const float *pp = part + (int64_t)blockIdx.x * P * 2;
    double s1 = 0, s2 = 0;
    for (int i = tid; i < P; i += 256) { s1 += (double)pp[2 * i]; s2 += (double)pp[2 * i + 1]; }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((tid & 63) == 0) { red[tid >> 6] = s1; red[4 + (tid >> 6)] = s2; }
    __syncthreads();
    if (tid == 0) {
        const double t1 = (red[0] + red[1]) + (red[2] + red[3]), t2 = (red[4] + red[5]) + (red[6] + red[7]);
        const double mean = t1 / n;
        double var = t2 / n - mean * mean;
        if (var < 0) var = 0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const float *zi = z + (int64_t)blockIdx.x * n;
    float *oi = out + (int64_t)blockIdx.x * n;
    for (int i = tid; i < n; i += 256) {
        const float t = (zi[i] - mean) * rstd * w[i] + b[i];
        oi[i] = after_bn ? act_fn(t, act) : t;
    }
}

int launch_ln_apply(const SubLayer &L, const float *z, const float *part, int P, float *out, int64_t B, int act,
                    int after_bn, hipStream_t s) {
    const int n = L.co * L.Fo * L.To;
    PF_LAUNCH(ln_apply_kernel, dim3((unsigned)B), dim3(256), 0, s, z, part, P, L.ln_w, L.ln_b, out, n, act,
                       after_bn);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Projection head with the last LayerNorm(+act) applied on load.
// ------------------------------------------------------------------------------------
// SEG segments per workgroup: thread g (one output dimension) keeps its v inputs of each segment in
// registers and walks its u hidden units once, so every weight row is loaded once per SEG segments
// (float4 loads when v % 4 == 0) instead of once per segment.
template <int SEG, int VMAX>
__global__ void myg_ln_kernel(const float *__restrict__ z, const float *__restrict__ part, int P,
                              const float *__restrict__ lw, const float *__restrict__ lb, int act, int after_bn,
                              const float *__restrict__ w1, const float *__restrict__ b1,
                              const float *__restrict__ w2, const float *__restrict__ b2, int d, int u, int v,
                              float *__restrict__ emb, int normalize, int64_t B) {
    __shared__ float red[SEG][16];
    __shared__ float stat[SEG][2];
    const int g = threadIdx.x;
    const int h = d * v;
    const int64_t seg0 = (int64_t)blockIdx.x * SEG;
    if (g < SEG && seg0 + g < B) {
        const float *pp = part + (seg0 + g) * P * 2;
        double s1 = 0, s2 = 0;
        for (int i = 0; i < P; ++i) { s1 += (double)pp[2 * i]; s2 += (double)pp[2 * i + 1]; }
        const double mean = s1 / h;
        double var = s2 / h - mean * mean;
        if (var < 0) var = 0;
        stat[g][0] = (float)mean;
        stat[g][1] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    float y[SEG];
#pragma unroll
    for (int sg = 0; sg < SEG; ++sg) y[sg] = 0.f;
    if (g < d) {
        float xin[SEG][VMAX];
#pragma unroll
        for (int sg = 0; sg < SEG; ++sg) {
            const bool ok = seg0 + sg < B;
            const float mean = ok ? stat[sg][0] : 0.f, rstd = ok ? stat[sg][1] : 0.f;
            const float *xs = z + (seg0 + sg) * h;
#pragma unroll
            for (int j = 0; j < VMAX; ++j) {
                float t = 0.f;
                if (ok && j < v) {
                    t = (xs[g * v + j] - mean) * rstd * lw[g * v + j] + lb[g * v + j];
                    t = after_bn ? act_fn(t, act) : t;
                }
                xin[sg][j] = t;
            }
        }
        for (int k = 0; k < u; ++k) {
            const float *wr = w1 + ((int64_t)g * u + k) * v;
            float wv[VMAX];
            if ((v & 3) == 0) {
#pragma unroll
                for (int j = 0; j < VMAX; j += 4)
                    if (j < v) {
                        const float4 t = *reinterpret_cast<const float4 *>(wr + j);
                        wv[j] = t.x; wv[j + 1] = t.y; wv[j + 2] = t.z; wv[j + 3] = t.w;
                    }
            } else {
#pragma unroll
                for (int j = 0; j < VMAX; ++j) wv[j] = j < v ? wr[j] : 0.f;
            }
            const float bb = b1[g * u + k], ww = w2[g * u + k];
#pragma unroll
            for (int sg = 0; sg < SEG; ++sg) {
                float hsum = 0.f;
#pragma unroll
                for (int j = 0; j < VMAX; ++j)
                    if (j < v) hsum = fmaf(wv[j], xin[sg][j], hsum);
                hsum += bb;
                hsum = hsum > 0.f ? hsum : expm1f(hsum);
                y[sg] = fmaf(ww, hsum, y[sg]);
            }
        }
#pragma unroll
        for (int sg = 0; sg < SEG; ++sg) y[sg] += b2[g];
    }
    if (normalize) {
#pragma unroll
        for (int sg = 0; sg < SEG; ++sg) {
            const float ss = wave_sum(g < d ? y[sg] * y[sg] : 0.f);
            if ((g & 63) == 0) red[sg][g >> 6] = ss;
        }
        __syncthreads();
#pragma unroll
        for (int sg = 0; sg < SEG; ++sg) {
            float tot = 0.f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[sg][i];
            y[sg] = y[sg] / fmaxf(sqrtf(tot), 1e-12f);
        }
    }
#pragma unroll
    for (int sg = 0; sg < SEG; ++sg)
        if (g < d && seg0 + sg < B) emb[(seg0 + sg) * d + g] = y[sg];
}

// Small batches (one query): the kernel above walks a thread's u hidden units one after the other, u dependent
// round trips to the weights (39 us for 19 segments).  Here one workgroup per segment spreads the d*u hidden units
// over all its threads (every weight row is one independent load) and leaves them in LDS; thread g then adds its u
// terms IN THE SAME ORDER as myg_ln_kernel does, so the two kernels give bit-identical embeddings.
__global__ __launch_bounds__(1024) void myg_ln_small_kernel(const float *__restrict__ z, const float *__restrict__ part, int P,
                                                            const float *__restrict__ lw, const float *__restrict__ lb, int act,
                                                            int after_bn, const float *__restrict__ w1, const float *__restrict__ b1,
                                                            const float *__restrict__ w2, const float *__restrict__ b2, int d, int u,
                                                            int v, float *__restrict__ emb, int normalize) {
    extern __shared__ float sm[];               // xn[h] | hid[d*u]
    __shared__ float stat[2];
    __shared__ float red[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int h = d * v;
    const int64_t b = blockIdx.x;
    float *xn = sm, *hid = sm + h;
    if (tid == 0) {
        const float *pp = part + b * P * 2;
        double s1 = 0, s2 = 0;
        for (int i = 0; i < P; ++i) { s1 += (double)pp[2 * i]; s2 += (double)pp[2 * i + 1]; }
        const double mean = s1 / h;
        double var = s2 / h - mean * mean;
        if (var < 0) var = 0;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    for (int e = tid; e < h; e += nt) {
        float t = (z[b * h + e] - mean) * rstd * lw[e] + lb[e];
        xn[e] = after_bn ? act_fn(t, act) : t;
    }
    __syncthreads();
    for (int pi = tid; pi < d * u; pi += nt) {              // hidden unit (g, k) = pi
        const int g = pi / u;
        const float *wr = w1 + (int64_t)pi * v;
        float hsum = 0.f;
        for (int j = 0; j < v; ++j) hsum = fmaf(wr[j], xn[g * v + j], hsum);
        hsum += b1[pi];
        hid[pi] = hsum > 0.f ? hsum : expm1f(hsum);
    }
    __syncthreads();
    float y = 0.f;
    if (tid < d) {
        for (int k = 0; k < u; ++k) y = fmaf(w2[tid * u + k], hid[tid * u + k], y);
        y += b2[tid];
    }
    if (normalize) {
        const float ss = wave_sum(tid < d ? y * y : 0.f);
        if ((tid & 63) == 0 && (tid >> 6) < 16) red[tid >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
        for (int i = 0; i < (d + 63) / 64; ++i) tot += red[i];       // the waves that hold outputs, in order
        y = y / fmaxf(sqrtf(tot), 1e-12f);
    }
    if (tid < d) emb[b * d + tid] = y;
}

int launch_myg_ln(const SubLayer &Llast, const float *z, const float *part, int P, int act, int after_bn,
                  const float *w1, const float *b1, const float *w2, const float *b2, int d, int u, int v, int64_t B,
                  float *emb, int normalize, hipStream_t s, int64_t Bplan) {
    if (v > 32) { set_error("MyG: h/d = %d > 32 unsupported", v); return -1; }
    const int nt = ((d + 63) / 64) * 64;
    if (nt > 1024) { set_error("MyG: d = %d > 1024 unsupported", d); return -1; }
    ProfScope ps("myg_ln", s);
    static const bool no_small = getenv("PFANN_NO_SMALL_HEAD") != nullptr;
    const size_t small_lds = ((size_t)d * v + (size_t)d * u) * sizeof(float);
    if (B <= 64 && (Bplan <= 0 || Bplan <= 64) && !no_small && small_lds <= 60 * 1024) {
        int thr = d * u < 1024 ? ((d * u + 63) / 64) * 64 : 1024;
        if (thr < nt) thr = nt;
        PF_LAUNCH(myg_ln_small_kernel, dim3((unsigned)B), dim3(thr), small_lds, s, z, part, P, Llast.ln_w, Llast.ln_b, act, after_bn,
                  w1, b1, w2, b2, d, u, v, emb, normalize);
        PF_HIP(hipGetLastError());
        return 0;
    }
    if (v <= 8)
        PF_LAUNCH((myg_ln_kernel<4, 8>), dim3((unsigned)cdiv(B, 4)), dim3(nt), 0, s, z, part, P, Llast.ln_w, Llast.ln_b, act,
                  after_bn, w1, b1, w2, b2, d, u, v, emb, normalize, B);
    else
        PF_LAUNCH((myg_ln_kernel<1, 32>), dim3((unsigned)B), dim3(nt), 0, s, z, part, P, Llast.ln_w, Llast.ln_b, act,
                  after_bn, w1, b1, w2, b2, d, u, v, emb, normalize, B);
    PF_HIP(hipGetLastError());
    return 0;
}

// pfann_prewarm: one empty launch per translation unit makes the runtime load this unit's code object now
__global__ void noop_encoder_fused_kernel() {}
int prewarm_encoder_fused() {
    hipLaunchKernelGGL(noop_encoder_fused_kernel, dim3(1), dim3(1), 0, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace pfann
