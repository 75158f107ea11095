// A minimal victim for the observation of profiles/r5/NOTES.md section 6: a kernel that does nothing but a long chain of
// fp32 multiply-adds in registers -- once with packed instructions (v_pk_fma_f32: two lanes' worth per instruction), once with
// plain v_fma_f32 -- no LDS, no memory traffic beyond one store per thread.  tools/ubench/pk_victim.py runs it on one stream
// while the library's batched fp16 scan runs on another and compares its output with the quiet run's, bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -shared -fPIC tools/ubench/pk_victim.hip -o tools/ubench/libpk_victim.so
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ __launch_bounds__(256) void victim_kernel(float *out, int iters, float seed) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = seed + 1e-3f * (float)(tid & 1023), x1 = x0 * 0.5f + 0.25f;
    float a0 = 0.f, a1 = 0.f, b0 = 1.f, b1 = -1.f;
    const float c0 = 0.999f, c1 = 1.001f;
    for (int i = 0; i < iters; ++i) {
        if (PACKED) {
            f32x2 a = {a0, a1}, b = {b0, b1}, x = {x0, x1}, c = {c0, c1};
            a = __builtin_elementwise_fma(x, c, a);          // v_pk_fma_f32
            b = __builtin_elementwise_fma(b, c, x);
            x = x * c + a * 1e-3f;                            // v_pk_mul_f32 / v_pk_fma_f32
            a0 = a[0]; a1 = a[1]; b0 = b[0]; b1 = b[1]; x0 = x[0]; x1 = x[1];
        } else {
            a0 = __builtin_fmaf(x0, c0, a0); a1 = __builtin_fmaf(x1, c1, a1);
            b0 = __builtin_fmaf(b0, c0, x0); b1 = __builtin_fmaf(b1, c1, x1);
            x0 = __builtin_fmaf(a0, 1e-3f, x0 * c0); x1 = __builtin_fmaf(a1, 1e-3f, x1 * c1);
        }
        // keep the values bounded (and the loop from being folded)
        if (x0 > 4.f) x0 -= 3.f;
        if (x1 > 4.f) x1 -= 3.f;
    }
    out[tid] = (a0 + a1) + (b0 + b1) + (x0 + x1);
}

extern "C" int pk_victim_launch(void *stream, float *out, int n_threads, int iters, int packed) {
    const dim3 grid((unsigned)(n_threads / 256)), block(256);
    if (packed) hipLaunchKernelGGL(victim_kernel<true>, grid, block, 0, (hipStream_t)stream, out, iters, 0.5f);
    else hipLaunchKernelGGL(victim_kernel<false>, grid, block, 0, (hipStream_t)stream, out, iters, 0.5f);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
