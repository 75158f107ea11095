// L2 -> CU load-path throughput (tuning aid, round 6): every workgroup streams 16 KB pieces of an L2-resident buffer, as the
// batched scan's db tiles arrive, (0) with buffer_load_dwordx4 into registers, (1) with buffer_load_dwordx4 ... lds.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2bw.hip -o tools/ubench/l2bw && tools/ubench/l2bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline __amdgpu_buffer_rsrc_t srd(const void *p, uint64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <int MODE, int DEPTH>     // DEPTH = 16 KB pieces in flight per workgroup
__global__ __launch_bounds__(256) void k(const char *src, unsigned region, int steps, unsigned *out, unsigned spread) {
    __shared__ __attribute__((aligned(1024))) char buf[DEPTH][16384];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const __amdgpu_buffer_rsrc_t r = srd(src, region);
    // spread = 0: every workgroup walks the SAME sequence of pieces from the same start (as the 76 query tiles of a db slice do);
    // spread = n: workgroup b starts (b % n) pieces further on
    unsigned pos = spread ? (unsigned)(((unsigned long long)(blockIdx.x % spread) * 16384ull) % region) : 0u;
    u32x4 acc = {0, 0, 0, 0};
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int dpt = 0; dpt < DEPTH; ++dpt) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned off = pos + (unsigned)((wave * 4 + u) * 1024 + lane * 16);
                if (MODE == 0) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
                    acc ^= v;
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)&buf[dpt][(wave_u * 4 + u) * 1024], 16,
                                                             (int)(pos + (unsigned)(lane * 16)), (wave_u * 4 + u) * 1024, 0, 0);
                }
            }
            pos += 16384u;
            if (pos >= region) pos -= region;
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 1) { __syncthreads(); acc[0] = *reinterpret_cast<unsigned *>(&buf[0][tid * 4]); }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

template <int MODE, int DEPTH>
static void run(const char *name, const char *src, unsigned region, int wgs, unsigned *out, unsigned spread) {
    const int steps = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, src, region, 256, out, spread);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, src, region, steps, out, spread);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * steps * 16384.0;
    printf("%-30s region %5u KB, start spread %4u, %4d workgroups (%d x 16 KB in flight each): %7.2f TB/s = %5.1f B/clk/CU at 2.2 GHz\n", name, region >> 10, spread, wgs,
           DEPTH, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256.0 / 2.2e9);
}

int main() {
    char *src; unsigned *out;
    hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20);
    hipMalloc(&out, 1 << 20);
    for (unsigned region : {2u << 20, 32u << 20}) {
        for (unsigned spread : {0u, 4u, 16u, 64u, 2048u}) {
            for (int wgs : {768}) {
                run<0, 1>("registers, 1 piece in flight", src, region, wgs, out, spread);
                run<0, 2>("registers, 2 pieces in flight", src, region, wgs, out, spread);
                run<1, 1>("LDS-DMA, 1 piece in flight", src, region, wgs, out, spread);
                run<1, 2>("LDS-DMA, 2 pieces in flight", src, region, wgs, out, spread);
            }
        }
    }
    for (int wgs : {64, 256, 1536}) {
        run<0, 2>("registers, 2 pieces in flight", src, 32u << 20, wgs, out, 2048u);
        run<1, 2>("LDS-DMA, 2 pieces in flight", src, 32u << 20, wgs, out, 2048u);
        run<1, 3>("LDS-DMA, 3 pieces in flight", src, 32u << 20, wgs, out, 2048u);
    }
    return 0;
}
