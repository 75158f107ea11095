// What slows the fp32 MFMA stream when a GEMM loop also streams operands?  (tuning aid, not product code)
// A stripped 128x128x32 K-tile loop in the fused GEMM's shape (8 waves, 64x32 wave tiles, LDS double buffer, one barrier per
// K-tile), with switches:  LOADS (8 x 16 B global loads per thread and K-tile from an L2-resident array), REFILL (4
// ds_write_b128), FRAGS (12 ds_read_b128), and ACC_AGPR (accumulators in AGPRs through inline asm instead of VGPRs).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mix.hip -o tools/ubench/mfma_mix && tools/ubench/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LOADS, bool REFILL, bool FRAGS, bool AGPR>
__global__ __launch_bounds__(512, 4) void k(const float *__restrict__ g, float *out, int nk) {
    constexpr int LDK = 36;
    __shared__ __attribute__((aligned(16))) float As[2 * 128 * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2 * 128 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhalf = lane >> 5, col4 = tid & 7, rowq = tid >> 3;
    for (int i = tid; i < 2 * 128 * LDK; i += 512) { As[i] = 0.001f * (i & 63); Bs[i] = 0.002f * (i & 31); }
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    if (AGPR) asm volatile("" : "+a"(acc0), "+a"(acc1));
    const f32x4 *gp = reinterpret_cast<const f32x4 *>(g) + (blockIdx.x & 63) * 4096 + tid;
    f32x4 st[8];
    for (int j = 0; j < 8; ++j) st[j] = f32x4{1.f, 2.f, 3.f, 4.f};
    for (int kt = 0; kt < nk; ++kt) {
        const float *Ac = As + (kt & 1) * (128 * LDK), *Bc = Bs + (kt & 1) * (128 * LDK);
        float *An = As + ((kt + 1) & 1) * (128 * LDK), *Bn = Bs + ((kt + 1) & 1) * (128 * LDK);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a0, a1, b0;
            if (FRAGS || kt == 0) {
                a0 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + l31) * LDK + kk * 8 + lhalf * 4]);
                a1 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + 32 + l31) * LDK + kk * 8 + lhalf * 4]);
                b0 = *reinterpret_cast<const f32x4 *>(&Bc[(wn * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            } else { a0 = st[0]; a1 = st[1]; b0 = st[2]; }
            if (kk == 0 && LOADS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) st[j] = gp[((kt * 8 + j) & 7) * 512];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kk == 3 && REFILL) {
                __builtin_amdgcn_sched_barrier(0);
                *reinterpret_cast<f32x4 *>(&An[rowq * LDK + col4 * 4]) = st[0] + st[1] + st[2];
                *reinterpret_cast<f32x4 *>(&An[(rowq + 64) * LDK + col4 * 4]) = st[3] + st[4] + st[5];
                *reinterpret_cast<f32x4 *>(&Bn[rowq * LDK + col4 * 4]) = st[6];
                *reinterpret_cast<f32x4 *>(&Bn[(rowq + 64) * LDK + col4 * 4]) = st[7];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (AGPR) {
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(b0[s]), "v"(a0[s]));
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(b0[s]), "v"(a1[s]));
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a1[s], acc1, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (AGPR) asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r];
    for (int j = 0; j < 8; ++j) t += st[j][0];
    if (t == 12345.678f) out[0] = t;
}

template <typename F> static void run(const char *name, F launch, int nk) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = 512.0 * 128 * 128 * 32 * 2 * nk;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)  %s\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    float *g, *out;
    (void)hipMalloc(&g, 64 * 4096 * 16 + (1 << 20)); (void)hipMalloc(&out, 64);
    (void)hipMemset(g, 0, 64 * 4096 * 16);
    const int nk = 2000;
#define T(name, L, R, F, A) run(name, [&] { hipLaunchKernelGGL((k<L, R, F, A>), dim3(512), dim3(512), 0, 0, g, out, nk); }, nk)
    T("mfma only (frags once)              vgpr", false, false, false, false);
    T("mfma only (frags once)              agpr", false, false, false, true);
    T("+ 12 ds_read_b128                   vgpr", false, false, true, false);
    T("+ 12 ds_read_b128                   agpr", false, false, true, true);
    T("+ ds_read + 8 global loads          vgpr", true, false, true, false);
    T("+ ds_read + 8 global loads          agpr", true, false, true, true);
    T("+ ds_read + loads + 4 ds_write      vgpr", true, true, true, false);
    T("+ ds_read + loads + 4 ds_write      agpr", true, true, true, true);
    T("+ ds_read + 4 ds_write (no loads)   vgpr", false, true, true, false);
    return 0;
}
