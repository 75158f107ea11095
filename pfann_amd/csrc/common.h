// Internal helpers shared by the HIP translation units of libpfann_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/pfann_amd.h"

namespace pfann {

void set_error(const char *fmt, ...);

#define PF_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            pfann::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr,                 \
                             hipGetErrorString(_e));                                      \
            return -1;                                                                    \
        }                                                                                 \
    } while (0)

// Kernel launch with a clean error slate: other libraries in the process (PyTorch probes host
// pointers with hipPointerGetAttributes) leave stale errors in HIP's per-thread last-error slot,
// which the post-launch hipGetLastError() check must not inherit.
#define PF_LAUNCH(...)                 \
    do {                               \
        (void)hipGetLastError();       \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)

// ---- per-kernel HIP-event profiling (bench.py roofline leg) -----------------------------
bool prof_on();
void prof_begin(const char *tag, hipStream_t s);
void prof_end(const char *tag, hipStream_t s, double work);

// `work` = algorithmic work of the bracketed launch (flops for MFMA-bound kernels, bytes for
// HBM-bound ones), summed per tag and read back by bench.py for the roofline.
struct ProfScope {
    const char *tag; hipStream_t s; bool on; double work;
    ProfScope(const char *t, hipStream_t st, double w = 0.0) : tag(t), s(st), on(prof_on()), work(w) {
        if (on) prof_begin(tag, s);
    }
    ~ProfScope() { if (on) prof_end(tag, s, work); }
};

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over lanes 0..31 (both halves hold the same values): xor 16..1
__device__ __forceinline__ float wave_sum_half(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- bounds-checked buffer access (gfx950 SRD): out-of-range lanes read 0 / drop stores, so
// predicated tile loads need neither branches nor selects and can sit in MFMA shadows.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
static constexpr unsigned BUF_OOB = 0x80000000u;          // any offset >= num_records
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *base, unsigned long long bytes) {
    const unsigned n = bytes > 0x7FFFFFF0ull ? 0x7FFFFFF0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)n, 0x00020000);
}
// soff: wave-uniform byte offset (SGPR), added to the address but not part of the range check
__device__ __forceinline__ f32x4_t buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff = 0) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, soff, 0);
    return __builtin_bit_cast(f32x4_t, v);
}

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2_t buf_load2u(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff = 0) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, soff, 0);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, 0);
}

__device__ __forceinline__ float act_fn(float v, int act) {   // 0 ReLU, 1 ELU (model.py:7-12)
    return act == 0 ? fmaxf(v, 0.0f) : (v > 0.0f ? v : expm1f(v));
}

// Bijective XCD remap: hardware places block b on XCD b%8; give each XCD a contiguous
// chunk of the logical tile order so blocks sharing an operand panel share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// n / d for 0 <= n < 2^31 and a launch-constant divisor d >= 1, as one multiply-high and a shift (a 32-bit scalar
// division is ~40 dependent instructions; in a GEMM prologue that shares its SIMD with another workgroup's MFMA stream
// every dependent instruction costs 10+ cycles).  k = ceil(log2 d), mul = floor(2^(31+k) / d) + 1 < 2^32 for d >= 2:
// n * mul / 2^(31+k) = n/d + n*e/2^(31+k) with 0 < e <= 1, and n*e/2^(31+k) < 2^-k <= 1/d, so the floor is exact.
struct FastDiv {
    unsigned mul;      // 0: d == 1
    int shift;         // k - 1
    int d;
};
inline FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = d; f.mul = 0; f.shift = 0;
    if (d <= 1) return f;
    int k = 0;
    while ((1ll << k) < d) ++k;
    f.mul = (unsigned)(((1ull << (31 + k)) / (unsigned long long)d) + 1ull);
    f.shift = k - 1;
    return f;
}
__device__ __forceinline__ int fastdiv(int n, const FastDiv &f) {
    return f.mul == 0 ? n : (int)(__umulhi((unsigned)n, f.mul) >> f.shift);
}

// ---- encoder plan shared between api and kernels ---------------------------------------
struct SubLayer {          // one conv (+LN+act) sub-layer, channels-last activations
    int ci, co;            // input / output channels
    int F, T;              // input spatial dims
    int Fo, To;            // output spatial dims
    int axis;              // 0: conv along T (conv1, 1x3), 1: conv along F (conv2, 3x1)
    int stride, pad_lo;    // along the convolved axis
    int depthwise;         // conv2 of non-"fuller" models
    float *w;              // [co][3][ci] (full) or [co][3] (depthwise / ci==1)
    float *bias;           // [co]
    float *ln_w, *ln_b;    // [Fo][To][co]  (channels-last re-layout of [co][Fo][To])
    // optional 2-term fp16 split of w * w_scale (w_scale a power of two), same [co][3][ci] layout:
    // w * w_scale = hi + lo to 2^-22 relative (encoder_fused.hip, SPLIT kernels)
    void *w_hi, *w_lo;
    float w_inv_scale;
    // [co][4][ci] = {W1, W0, -W2, W0 + W2}: the weights of conv_gemm_ln_w22_kernel (stride-2 convs computed with five
    // channel blocks per output pair instead of six)
    float *w22;
};

}  // namespace pfann
