// Front-end kernels for gfx950: segment mean-removal + L2/max normalise + STFT + |.|^p +
// sparse mel filterbank + log, all in one kernel (reference datautil/musicdata.py:88 and
// datautil/melspec.py:33-50 around torchaudio MelSpectrogram/torch.stft).
//
// One 256-thread workgroup per 1-second segment.  The segment is read from HBM/L2 with
// coalesced loads (each sample is touched by 4 overlapping frames, served by L1/L2); each of
// the 4 waves runs a 1024-point real FFT for one frame at a time as a 512-point complex
// radix-2 FFT in LDS (SoA re/im, host-computed twiddles staged in LDS), the power spectrum
// goes through the 0.7 %-dense mel bank as a CSR gather-MAC (never a dense GEMM), and the
// [n_mels][n_frames] tile is assembled in LDS so the store to HBM is one coalesced stream.
#include <stdlib.h>

#include "kernels.h"

namespace pfann {

struct MelArgs {
    const float *segs; float *out;
    int64_t seg_stride;
    const int64_t *starts;   // optional [B]: window b begins at segs + starts[b]
    int seg_len, n_fft, hop, n_mels, n_frames, log2n;
    int power, pad_reflect, log_mode, spec_norm_max, remove_mean;
    float log_eps;
    const float *window; const float2 *twiddle;
    const int *fb_ptr, *fb_idx; const float *fb_val;
    int fb_nnz;
    int parts;               // workgroups per segment (small batches: each takes n_frames / parts frames = whole output groups)
    int group_out;           // 0: whole [n_mels][n_frames] tile in LDS; else frames per output group (4, 8 or 16): the
                             // group leaves through a small LDS tile as group_out*4-byte row pieces (3 workgroups / CU)
};

__device__ __forceinline__ unsigned bitrev(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }          // a * (-i)

// forward 8-point DFT in registers (radix-2 DIF), natural order in and out
__device__ __forceinline__ void dft8(float2 (&a)[8]) {
    const float c = 0.70710678118654752440f;
    float2 u[8], v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { u[j] = cadd(a[j], a[j + 4]); u[j + 4] = csub(a[j], a[j + 4]); }
    u[5] = make_float2(c * (u[5].x + u[5].y), c * (u[5].y - u[5].x));       // * W8^1 = c(1 - i)
    u[6] = mul_mi(u[6]);                                                    // * W8^2 = -i
    u[7] = make_float2(c * (u[7].y - u[7].x), -c * (u[7].x + u[7].y));      // * W8^3 = -c(1 + i)
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
        v[h] = cadd(u[h], u[h + 2]); v[h + 2] = csub(u[h], u[h + 2]);
        v[h + 1] = cadd(u[h + 1], u[h + 3]); v[h + 3] = mul_mi(csub(u[h + 1], u[h + 3]));
    }
    a[0] = cadd(v[0], v[1]); a[4] = csub(v[0], v[1]);
    a[2] = cadd(v[2], v[3]); a[6] = csub(v[2], v[3]);
    a[1] = cadd(v[4], v[5]); a[5] = csub(v[4], v[5]);
    a[3] = cadd(v[6], v[7]); a[7] = csub(v[6], v[7]);
}

__global__ __launch_bounds__(256) void melspec_kernel(MelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int M = a.n_fft >> 1;           // complex FFT length
    const int log2m = a.log2n - 1;
    const int n_freqs = M + 1;
    // LDS carve-up
    float *tw_re = smem;                  // [M]   exp(-2*pi*i*j/n_fft), j < M
    float *tw_im = tw_re + M;             // [M]
    const int WSZ = 2 * M > 1152 ? 2 * M : 1152;
    float *work = tw_im + M;              // [4][WSZ]  per-wave FFT buffer (re[M], im[M]; radix-8 path: 576 float2)
    float *pw = work + 4 * WSZ;           // [4][M+4]  power spectrum per wave
    float *tile = pw + 4 * (M + 4);       // [n_mels][n_frames+1], or [n_mels][group_out+1]
    const int tp = a.group_out ? a.group_out + 1 : a.n_frames + 1;       // tile row pitch
    float *red = tile + a.n_mels * tp;    // [8]
    int *s_ptr = reinterpret_cast<int *>(red + 8);    // mel bank CSR, resident in LDS: [n_mels+1], [nnz], [nnz]
    int *s_idx = s_ptr + a.n_mels + 1;
    float *s_val = reinterpret_cast<float *>(s_idx + a.fb_nnz);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = a.parts > 1 ? (int)blockIdx.x / a.parts : (int)blockIdx.x, part = a.parts > 1 ? (int)blockIdx.x % a.parts : 0;
    const float *x = a.segs + (a.starts ? a.starts[seg] : (int64_t)seg * a.seg_stride);

    for (int j = tid; j < M; j += 256) { const float2 t = a.twiddle[j]; tw_re[j] = t.x; tw_im[j] = t.y; }
    for (int j = tid; j <= a.n_mels; j += 256) s_ptr[j] = a.fb_ptr[j];
    for (int j = tid; j < a.fb_nnz; j += 256) { s_idx[j] = a.fb_idx[j]; s_val[j] = a.fb_val[j]; }

    // ---- segment statistics: mean (optional), then L2 norm or max-abs of (x - mean).  The segment is
    // read once, all loads in flight together (32 registers per thread), instead of two latency-bound
    // passes over global memory; summation order is the same as a strided loop's.
    constexpr int SMAX = 32;
    const bool inreg = a.seg_len <= SMAX * 256;
    float xr[SMAX];
#pragma unroll
    for (int j = 0; j < SMAX; ++j) {
        const int i = tid + 256 * j;
        xr[j] = (inreg && i < a.seg_len) ? x[i] : 0.f;
    }
    float mean = 0.f;
    if (a.remove_mean) {
        float s = 0.f;
        if (inreg) {
#pragma unroll
            for (int j = 0; j < SMAX; ++j) if (tid + 256 * j < a.seg_len) s += xr[j];
        } else {
            for (int i = tid; i < a.seg_len; i += 256) s += x[i];
        }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)a.seg_len;
        __syncthreads();
    }
    float denom;
    {
        float s = 0.f;
        if (a.spec_norm_max) {
            if (inreg) {
#pragma unroll
                for (int j = 0; j < SMAX; ++j) if (tid + 256 * j < a.seg_len) s = fmaxf(s, fabsf(xr[j] - mean));
            } else {
                for (int i = tid; i < a.seg_len; i += 256) s = fmaxf(s, fabsf(x[i] - mean));
            }
            s = wave_max(s);
        } else {
            if (inreg) {
#pragma unroll
                for (int j = 0; j < SMAX; ++j)
                    if (tid + 256 * j < a.seg_len) { const float v = xr[j] - mean; s = fmaf(v, v, s); }
            } else {
                for (int i = tid; i < a.seg_len; i += 256) { const float v = x[i] - mean; s = fmaf(v, v, s); }
            }
            s = wave_sum(s);
        }
        if (lane == 0) red[4 + wave] = s;
        __syncthreads();
        const float t = a.spec_norm_max ? fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]))
                                        : sqrtf((red[4] + red[5]) + (red[6] + red[7]));
        denom = fmaxf(t, 1e-12f);
    }
    // v / denom for the 16 samples a lane gathers per frame, correctly rounded without the division sequence (~10
    // instructions each, a third of the frame loop's arithmetic): with y = RN(1 / denom) (one real division per window),
    // q = RN(v y), r = v - q denom (exact in an fma), RN(q + r y) IS RN(v / denom) (Markstein's theorem; no overflow or
    // underflow here: |v| is 0 or >= 2^-24-ish PCM differences against denom in [1e-12, 3e6]).  Same bits as before.
    const float rden = 1.0f / denom;
    auto div_denom = [&](float v) {
        const float q = v * rden;
        return fmaf(fmaf(-q, denom, v), rden, q);
    };

    // each wave owns its FFT buffers: LDS operations of one wave execute in order, so a wave-level
    // fence (no s_barrier) is all the stages need
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    float *zre = work + wave * WSZ, *zim = zre + M, *pwr = pw + wave * (M + 4);
    const bool radix8 = a.n_fft == 1024;
    float win[8][2];                      // radix-8 path: this lane's 16 window taps, the same for every frame
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
#pragma unroll
        for (int e = 0; e < 2; ++e) win[n1][e] = radix8 ? a.window[2 * (64 * n1 + lane) + e] : 0.f;
    // radix-8 path: the lane's 7 + 7 twiddles of the two inter-stage multiplications do not depend on the frame either: in
    // registers (read from the LDS table per frame, lanes 2 k1 or 16 c words apart, they were 2- to 8-way bank conflicts;
    // round 5, with the pitch-9 transpose: 1.149 -> 1.090 ms per 9728 windows, same bits)
    float2 tw1[8], tw2[8];
#pragma unroll
    for (int j = 1; j < 8; ++j) {
        const int e1 = radix8 ? 2 * lane * j : 0, e2 = radix8 ? 16 * (lane & 7) * j : 0;
        tw1[j] = make_float2((e1 >= 512 ? -1.f : 1.f) * tw_re[e1 & 511], (e1 >= 512 ? -1.f : 1.f) * tw_im[e1 & 511]);     // W_512^(n2 k1)
        tw2[j] = make_float2((e2 >= 512 ? -1.f : 1.f) * tw_re[e2 & 511], (e2 >= 512 ? -1.f : 1.f) * tw_im[e2 & 511]);     // W_64^(b c)
    }
    const int n_groups = (a.n_frames + 3) >> 2;
    const int g_lo = part * (n_groups / a.parts), g_hi = a.parts > 1 ? g_lo + n_groups / a.parts : n_groups;
    for (int g = g_lo; g < g_hi; ++g) {
        const int t = g * 4 + wave;
        const bool live = t < a.n_frames;
        if (radix8) {
            // ---- 512-point complex FFT as 8 x 8 x 8, eight points per lane in registers, two LDS
            // transposes (Cooley-Tukey n = 64 n1 + n2, then n2 = 8 a + b); the frame is gathered straight
            // into registers (lane = n2), no bit-reversal pass
            float2 *tb = reinterpret_cast<float2 *>(zre);            // [8][72] float2 (pitch 72: conflict-free reads)
            float2 v[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const int m = 64 * n1 + lane;
                float w[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int n = 2 * m + e;
                    int idx = t * a.hop - M + n;
                    float s = 0.f;
                    if (live) {
                        if (a.pad_reflect) {
                            if (idx < 0) idx = -idx;
                            if (idx > a.seg_len - 1) idx = 2 * (a.seg_len - 1) - idx;
                            s = div_denom(x[idx] - mean);
                        } else if (idx >= 0 && idx < a.seg_len) {
                            s = div_denom(x[idx] - mean);
                        }
                    }
                    w[e] = s * win[n1][e];
                }
                v[n1] = make_float2(w[0], w[1]);
            }
            const int hi = lane >> 3, lo = lane & 7;                  // (k1, b) then (k1, c)
            dft8(v);                                                   // over n1 -> k1
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) v[k1] = cmul(v[k1], tw1[k1]);                  // W_512^(n2 k1)
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) tb[k1 * 72 + lane] = v[k1];
            wave_sync();
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = tb[hi * 72 + 8 * q + lo];               // n2 = 8a + b
            dft8(v);                                                   // over a -> c
#pragma unroll
            for (int c = 1; c < 8; ++c) v[c] = cmul(v[c], tw2[c]);                       // W_64^(b c)
            wave_sync();
#pragma unroll
            // (row pitch 9: lane (k1, c) then reads its eight b's from 32 different bank pairs per half wave; pitch 8 put
            // the lanes with equal (k1 + c) mod 4 on the same banks, 8 ways)
            for (int c = 0; c < 8; ++c) tb[hi * 72 + c * 9 + lo] = v[c];                 // [k1][c][b]
            wave_sync();
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = tb[hi * 72 + lo * 9 + q];               // lane (k1, c): b = 0..7
            dft8(v);                                                   // over b -> e;  X[k1 + 8c + 64e]
            wave_sync();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                zre[hi + 8 * lo + 64 * e] = v[e].x;
                zim[hi + 8 * lo + 64 * e] = v[e].y;
            }
            wave_sync();
        } else {
            // ---- gather frame (reflect / zero padding), normalise, window; write bit-reversed
            for (int m = lane; m < M; m += 64) {
                float v[2];
    #pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int n = 2 * m + e;
                    int idx = t * a.hop - M + n;
                    float s = 0.f;
                    if (live) {
                        if (a.pad_reflect) {
                            if (idx < 0) idx = -idx;
                            if (idx > a.seg_len - 1) idx = 2 * (a.seg_len - 1) - idx;
                            s = div_denom(x[idx] - mean);
                        } else if (idx >= 0 && idx < a.seg_len) {
                            s = div_denom(x[idx] - mean);
                        }
                    }
                    v[e] = s * a.window[n];
                }
                const unsigned r = bitrev((unsigned)m, log2m);
                zre[r] = v[0];
                zim[r] = v[1];
            }
            wave_sync();
            // ---- radix-2 DIT stages
            for (int s = 1; s <= log2m; ++s) {
                const int half = 1 << (s - 1);
                for (int bf = lane; bf < (M >> 1); bf += 64) {
                    const int grp = bf >> (s - 1), j = bf & (half - 1);
                    const int i0 = (grp << s) + j, i1 = i0 + half;
                    const int k = (j << (log2m - s)) << 1;   // W_M^(j*M/2^s) = tw[2*...]
                    const float wr = tw_re[k], wi = tw_im[k];
                    const float xr = zre[i1], xi = zim[i1];
                    const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
                    const float ur = zre[i0], ui = zim[i0];
                    zre[i0] = ur + tr; zim[i0] = ui + ti;
                    zre[i1] = ur - tr; zim[i1] = ui - ti;
                }
                wave_sync();
            }
        }
        // ---- real-FFT split + power:  X[k] = E[k] + W_N^k O[k],  k = 0..M
        for (int k = lane; k <= M; k += 64) {
            const int k0 = k & (M - 1), k1 = (M - k) & (M - 1);
            const float ar = zre[k0], ai = zim[k0], br = zre[k1], bi = -zim[k1];  // Z[k], conj(Z[M-k])
            const float er = 0.5f * (ar + br), ei = 0.5f * (ai + bi);
            const float dr = 0.5f * (ar - br), di = 0.5f * (ai - bi);            // (Z - conj)/2
            const float orr = di, oi = -dr;                                      // O = -i * d
            float wr, wi;
            if (k < M) { wr = tw_re[k]; wi = tw_im[k]; } else { wr = -1.f; wi = 0.f; }
            const float xr = er + (wr * orr - wi * oi), xi = ei + (wr * oi + wi * orr);
            const float p2 = xr * xr + xi * xi;
            pwr[k] = a.power == 2 ? p2 : sqrtf(p2);
        }
        wave_sync();
        // ---- sparse mel + log into the LDS tile
        if (live) {
            for (int m = lane; m < a.n_mels; m += 64) {
                float acc = 0.f;
                const int e1 = s_ptr[m + 1];
                int e = s_ptr[m];
                for (; e + 3 < e1; e += 4) {      // four independent gathers in flight; same summation order
                    const float v0 = s_val[e], v1 = s_val[e + 1], v2 = s_val[e + 2], v3 = s_val[e + 3];
                    const float p0 = pwr[s_idx[e]], p1 = pwr[s_idx[e + 1]], p2 = pwr[s_idx[e + 2]], p3 = pwr[s_idx[e + 3]];
                    acc = fmaf(v0, p0, acc); acc = fmaf(v1, p1, acc); acc = fmaf(v2, p2, acc); acc = fmaf(v3, p3, acc);
                }
                for (; e < e1; ++e) acc = fmaf(s_val[e], pwr[s_idx[e]], acc);
                acc += a.log_eps;
                if (a.log_mode == 1) acc = logf(acc);
                else if (a.log_mode == 2) acc = log10f(acc);
                if (a.group_out) tile[m * tp + (t % a.group_out)] = acc;
                else tile[m * tp + t] = acc;
            }
        }
        wave_sync();
        if (a.group_out && ((4 * g + 4) % a.group_out == 0)) {
            // a finished group of frames: group_out*4 contiguous bytes of every mel row of the [n_mels][n_frames]
            // output (16-byte pieces made HBM write traffic 4.1x the tensor: partial 64-byte requests)
            __syncthreads();
            const int q4 = a.group_out >> 2;                          // float4 pieces per row
            float *og = a.out + (int64_t)seg * a.n_mels * a.n_frames + (4 * g + 4 - a.group_out);
            for (int i = tid; i < a.n_mels * q4; i += 256) {
                const int m = i / q4, c = i - m * q4;
                const float *tr = tile + m * tp + 4 * c;
                *reinterpret_cast<float4 *>(og + (int64_t)m * a.n_frames + 4 * c) = make_float4(tr[0], tr[1], tr[2], tr[3]);
            }
            __syncthreads();
        }
    }
    if (a.group_out) return;
    __syncthreads();                      // the [n_mels][n_frames] tile is complete
    // ---- spec_norm == 'max': subtract the tile maximum (melspec.py:48-49)
    float sub = 0.f;
    if (a.spec_norm_max) {
        float mx = -INFINITY;
        for (int i = tid; i < a.n_mels * a.n_frames; i += 256)
            mx = fmaxf(mx, tile[(i / a.n_frames) * (a.n_frames + 1) + (i % a.n_frames)]);
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        sub = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    float *o = a.out + (int64_t)seg * a.n_mels * a.n_frames;
    if (256 % a.n_frames == 0) {          // thread -> fixed frame, mel rows advance by 256 / n_frames: no divisions
        const int tf = tid % a.n_frames, mstep = 256 / a.n_frames;
        int m = tid / a.n_frames;
        for (int i = tid; i < a.n_mels * a.n_frames; i += 256, m += mstep) o[i] = tile[m * (a.n_frames + 1) + tf] - sub;
    } else {
        for (int i = tid; i < a.n_mels * a.n_frames; i += 256)
            o[i] = tile[(i / a.n_frames) * (a.n_frames + 1) + (i % a.n_frames)] - sub;
    }
}

int launch_melspec(const MelPlan &mp, const float *segs, int64_t B, int64_t seg_stride, const int64_t *starts,
                   int remove_mean, float *out, hipStream_t s) {
    if (B <= 0) return 0;
    MelArgs a;
    a.segs = segs; a.out = out; a.seg_stride = seg_stride; a.starts = starts;
    a.seg_len = mp.seg_len; a.n_fft = mp.n_fft; a.hop = mp.hop; a.n_mels = mp.n_mels;
    a.n_frames = mp.n_frames; a.log2n = mp.log2n;
    a.power = mp.power; a.pad_reflect = mp.pad_reflect; a.log_mode = mp.log_mode;
    a.spec_norm_max = mp.spec_norm_max; a.remove_mean = remove_mean; a.log_eps = mp.log_eps;
    a.window = mp.window; a.twiddle = mp.twiddle;
    a.fb_ptr = mp.fb_ptr; a.fb_idx = mp.fb_idx; a.fb_val = mp.fb_val; a.fb_nnz = mp.fb_nnz;
    const int M = mp.n_fft / 2;
    const int WSZ = 2 * M > 1152 ? 2 * M : 1152;
    auto lds_for = [&](int gf) {
        return sizeof(float) * (size_t)(2 * M + 4 * WSZ + 4 * (M + 4) + mp.n_mels * (gf ? gf + 1 : mp.n_frames + 1) + 8 +
                                        mp.n_mels + 1 + 2 * (size_t)mp.fb_nnz);
    };
    // widest output group (16, 8 or 4 frames = 64 / 32 / 16-byte row pieces) that still leaves three workgroups per CU
    a.group_out = 0;
    if (!mp.spec_norm_max) {
        for (int gf : {16, 8, 4})
            if (mp.n_frames % gf == 0 && (a.group_out == 0 ? (3 * lds_for(gf) <= 160 * 1024 || gf == 4) : false)) a.group_out = gf;
    }
    if (getenv("PFANN_MEL_GROUP")) a.group_out = atoi(getenv("PFANN_MEL_GROUP"));
    // small batches (one query = 19 segments): a workgroup per output group instead of per segment, so that the launch
    // covers more than a handful of CUs; same arithmetic in the same order (every workgroup recomputes the segment statistics)
    a.parts = 1;
    if (a.group_out && B <= 192 && mp.n_frames % a.group_out == 0 && (mp.n_frames / a.group_out) * a.group_out == mp.n_frames &&
        a.group_out % 4 == 0 && mp.n_frames % 4 == 0)
        a.parts = mp.n_frames / a.group_out;
    const size_t lds = lds_for(a.group_out);
    if (lds > 160 * 1024) { set_error("melspec: LDS need %zu B > 160 KiB", lds); return -1; }
    if (ensure_dyn_lds((const void *)melspec_kernel, 160 * 1024)) return -1;
    ProfScope ps("melspec", s);
    PF_LAUNCH(melspec_kernel, dim3((unsigned)(B * a.parts)), dim3(256), lds, s, a);
    PF_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// int16 interleaved PCM -> float mono (musicdata.py:48,72-80)
// ------------------------------------------------------------------------------------
__global__ void stereo_power_kernel(const int16_t *__restrict__ pcm, int64_t n, double *pw) {
    double p1 = 0, p2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float l = (float)pcm[2 * i] * (1.0f / 32768.0f), r = (float)pcm[2 * i + 1] * (1.0f / 32768.0f);
        p1 += (double)((l - r) * (l - r));
        p2 += (double)((l + r) * (l + r));
    }
    p1 = wave_sum_d(p1); p2 = wave_sum_d(p2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&pw[0], p1); atomicAdd(&pw[1], p2); }
}
__global__ void pcm_to_mono_kernel(const int16_t *__restrict__ pcm, int64_t n, int n_ch, const double *pw,
                                   float *__restrict__ wav) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n_ch == 1) { wav[i] = (float)pcm[i] * (1.0f / 32768.0f); return; }
    float acc = 0.f;
    const bool flip = n_ch == 2 && (float)(pw[0] / (double)n) > (float)(pw[1] / (double)n) * 1000.0f;
    for (int c = 0; c < n_ch; ++c) {
        float v = (float)pcm[i * n_ch + c] * (1.0f / 32768.0f);
        if (flip && c == 1) v = -v;
        acc += v;
    }
    wav[i] = acc / (float)n_ch;
}

// ------------------------------------------------------------------------------------
// Files that are not at the model's sample rate (datautil/musicdata.py:28-65): julius.ResampleFrac as a polyphase
// FIR, applied piece by piece like the reference does (60 s pieces starting every 59 s, each with replicate padding at
// ITS edges, half a second dropped at the inner seams).  One thread per output sample and channel:
//   y[j * new + i] = sum_t K[i][t] * x[clamp(j * old + t - width, 0, len - 1)],  x = int16 / 32768
// plan[p] = {in_start, in_len, out_skip, out_keep, out_off}: piece p contributes outputs [out_off, out_off + out_keep).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_kernel(const int16_t *__restrict__ pcm, int n_ch, const float *__restrict__ K,
                                                       int old_r, int new_r, int width, const int64_t *__restrict__ plan,
                                                       int n_pieces, int64_t n_out, float *__restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (o >= n_out) return;
    int lo = 0, hi = n_pieces - 1;                 // last piece whose out_off <= o
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (plan[5 * mid + 4] <= o) lo = mid; else hi = mid - 1;
    }
    const int64_t in_start = plan[5 * lo], in_len = plan[5 * lo + 1];
    const int64_t jl = o - plan[5 * lo + 4] + plan[5 * lo + 2];
    const int64_t frame = jl / new_r;
    const int phase = (int)(jl - frame * new_r);
    const int taps = 2 * width + old_r;
    const float *kr = K + (int64_t)phase * taps;
    const int64_t x0 = frame * old_r - width;
    const int16_t *xs = pcm + in_start * n_ch + c;
    float acc = 0.f;
    for (int t = 0; t < taps; ++t) {
        int64_t xi = x0 + t;
        xi = xi < 0 ? 0 : (xi >= in_len ? in_len - 1 : xi);
        acc = fmaf(kr[t], (float)xs[xi * n_ch] * (1.0f / 32768.0f), acc);
    }
    out[(int64_t)c * n_out + o] = acc;
}
__global__ void planar_power_kernel(const float *__restrict__ x, int64_t n, double *pw) {
    double p1 = 0, p2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float l = x[i], r = x[n + i];
        p1 += (double)((l - r) * (l - r));
        p2 += (double)((l + r) * (l + r));
    }
    p1 = wave_sum_d(p1); p2 = wave_sum_d(p2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&pw[0], p1); atomicAdd(&pw[1], p2); }
}
__global__ void planar_to_mono_kernel(const float *__restrict__ x, int64_t n, int n_ch, const double *pw, float *__restrict__ wav) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n_ch == 1) { wav[i] = x[i]; return; }
    const bool flip = n_ch == 2 && (float)(pw[0] / (double)n) > (float)(pw[1] / (double)n) * 1000.0f;
    float acc = 0.f;
    for (int c = 0; c < n_ch; ++c) {
        float v = x[(int64_t)c * n + i];
        if (flip && c == 1) v = -v;
        acc += v;
    }
    wav[i] = acc / (float)n_ch;
}

int launch_resample_to_mono(const int16_t *pcm, int n_ch, const float *K, int old_r, int new_r, int width, const int64_t *plan,
                            int n_pieces, int64_t n_out, float *tmp, float *wav, float *scratch2, hipStream_t s) {
    if (n_out <= 0) return 0;
    if (n_ch < 1 || old_r < 1 || new_r < 1 || width < 1 || n_pieces < 1) { set_error("resample: bad arguments"); return -1; }
    double *pw = reinterpret_cast<double *>(scratch2);
    ProfScope ps("resample_to_mono", s);
    PF_LAUNCH(resample_kernel, dim3((unsigned)cdiv(n_out, 256), (unsigned)n_ch), dim3(256), 0, s, pcm, n_ch, K, old_r, new_r, width,
              plan, n_pieces, n_out, tmp);
    if (n_ch == 2) {
        PF_HIP(hipMemsetAsync(pw, 0, 2 * sizeof(double), s));
        PF_LAUNCH(planar_power_kernel, dim3(512), dim3(256), 0, s, tmp, n_out, pw);
    }
    PF_LAUNCH(planar_to_mono_kernel, dim3((unsigned)cdiv(n_out, 256)), dim3(256), 0, s, tmp, n_out, n_ch, pw, wav);
    PF_HIP(hipGetLastError());
    return 0;
}

// mono input, eight samples per thread: one 16-byte load, two 16-byte stores (the one-sample-per-thread form moves 128 bytes per
// load instruction); same arithmetic per sample
__global__ __launch_bounds__(256) void pcm_mono8_kernel(const int16_t *__restrict__ pcm, int64_t n8, float *__restrict__ wav) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const s16x8 v = reinterpret_cast<const s16x8 *>(pcm)[i];
    float4 lo, hi;
    lo.x = (float)v[0] * (1.0f / 32768.0f); lo.y = (float)v[1] * (1.0f / 32768.0f);
    lo.z = (float)v[2] * (1.0f / 32768.0f); lo.w = (float)v[3] * (1.0f / 32768.0f);
    hi.x = (float)v[4] * (1.0f / 32768.0f); hi.y = (float)v[5] * (1.0f / 32768.0f);
    hi.z = (float)v[6] * (1.0f / 32768.0f); hi.w = (float)v[7] * (1.0f / 32768.0f);
    reinterpret_cast<float4 *>(wav)[2 * i] = lo;
    reinterpret_cast<float4 *>(wav)[2 * i + 1] = hi;
}

int launch_pcm16_to_mono(const int16_t *pcm, int64_t n_frames, int n_ch, float *wav, float *scratch2,
                         hipStream_t s) {
    if (n_frames <= 0) return 0;
    double *pw = reinterpret_cast<double *>(scratch2);
    ProfScope ps("pcm16_to_mono", s);
    if (n_ch == 1 && n_frames >= 8 && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0 && (reinterpret_cast<uintptr_t>(wav) & 15) == 0) {
        const int64_t n8 = n_frames / 8, rest = n_frames - 8 * n8;
        PF_LAUNCH(pcm_mono8_kernel, dim3((unsigned)cdiv(n8, 256)), dim3(256), 0, s, pcm, n8, wav);
        if (rest > 0)
            PF_LAUNCH(pcm_to_mono_kernel, dim3(1), dim3(256), 0, s, pcm + 8 * n8, rest, 1, pw, wav + 8 * n8);
        PF_HIP(hipGetLastError());
        return 0;
    }
    if (n_ch == 2) {
        PF_HIP(hipMemsetAsync(pw, 0, 2 * sizeof(double), s));
        PF_LAUNCH(stereo_power_kernel, dim3(512), dim3(256), 0, s, pcm, n_frames, pw);
    }
    PF_LAUNCH(pcm_to_mono_kernel, dim3((unsigned)cdiv(n_frames, 256)), dim3(256), 0, s, pcm, n_frames,
                       n_ch, pw, wav);
    PF_HIP(hipGetLastError());
    return 0;
}

// pfann_prewarm: one empty launch per translation unit makes the runtime load this unit's code object now
__global__ void noop_mel_kernel() {}
int prewarm_mel() {
    hipLaunchKernelGGL(noop_mel_kernel, dim3(1), dim3(1), 0, 0);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace pfann
