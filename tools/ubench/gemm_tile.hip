// A stripped clone of the fused conv GEMM's tile life (tuning aid, not product code; results are not checked):
// one workgroup per 128x128 output tile, K = nk x 32, operands streamed like the K = 384 layers do (activations [rows][128]
// fp32 from HBM with a stride-2 three-tap im2col, LayerNorm weight/bias rows and the weight matrix from L2), the
// (z - mu) * rs * w + b transform on the way into LDS, and a choice of epilogues.  Which part of the distance between the
// bare loop (0.94 of peak, mfma_mix.hip) and the real kernel (0.72-0.74 on K = 384) is structural?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_tile.hip -o tools/ubench/gemm_tile && tools/ubench/gemm_tile
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

// EPI: 0 none, 1 direct (32 B per lane-pair row pieces, as the product kernel), 2 through LDS (whole 512 B rows)
// OPS: bit0 activation loads, bit1 LayerNorm-affine loads, bit2 weight loads, bit3 transform+refill
template <int EPI, int OPS, bool STATS, bool PF = false>
__global__ __launch_bounds__(512, 4) void k(const float *__restrict__ x, const float *__restrict__ lw, const float *__restrict__ lb,
                                            const float *__restrict__ w, const float *__restrict__ stats, float *__restrict__ y,
                                            float *__restrict__ part, int nk, int rps_out, int prio, unsigned long long *ts) {
    constexpr int LDK = 36, BM = 128, BN = 128, C = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem, *Bs = smem + 2 * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhalf = lane >> 5, col4 = tid & 7, rowq = tid >> 3;
    // prio: 1 = the workgroup whose LDS allocation starts at 0 gets s_setprio 3 (the co-resident one stays at 0);
    //       2 = by wave slot (HW_ID.wave_id & 2)
    if (prio == 1) { if (__builtin_amdgcn_s_getreg(6 | (7 << 11)) == 0) __builtin_amdgcn_s_setprio(3); }
    else if (prio == 2) { if (__builtin_amdgcn_s_getreg(4 | (3 << 11)) & 2) __builtin_amdgcn_s_setprio(3); }
    const unsigned long long t_start = __builtin_readcyclecounter();
    const int m0 = blockIdx.x * BM;                         // output rows m0..m0+127, all in one sample
    const int b = m0 / rps_out, r0 = m0 % rps_out;
    const int rps_in = 2 * rps_out;
    // prologue: per-sample statistics, row offsets
    const float mu = stats[2 * b], rs = stats[2 * b + 1];
    const __amdgpu_buffer_rsrc_t sx = srd(x + (int64_t)b * rps_in * C, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t slw = srd(lw, (uint64_t)rps_in * C * 4), slb = srd(lb, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t sw = srd(w, (uint64_t)BN * nk * 32 * 4);
    unsigned va[2], vb[2];
    int arow[2];
    for (int i = 0; i < 2; ++i) {
        arow[i] = 2 * (r0 + rowq + 64 * i) - 1;             // stride 2, pad 1
        vb[i] = (unsigned)(rowq + 64 * i) * (unsigned)(nk * 32) * 4u + col4 * 16u;
    }
    auto offs = [&](int tap) {
        for (int i = 0; i < 2; ++i) va[i] = (unsigned)(arow[i] + tap) < (unsigned)rps_in ? (unsigned)((arow[i] + tap) * C + col4 * 4) * 4u : 0x80000000u;
    };
    int tapA = 0, cA = 0, tap = 0, c = 0;
    unsigned vaA[2], vaL[2];
    offs(0); vaA[0] = va[0]; vaA[1] = va[1]; vaL[0] = va[0]; vaL[1] = va[1];
    f32x4 ra[2][2], rw[2], rbb[2], rb[2];
    for (int i = 0; i < 2; ++i)      // "stale" registers hold random values too (same switching activity in the MFMAs)
        for (int e = 0; e < 4; ++e) {
            unsigned h = (unsigned)(tid * 8 + i * 4 + e + blockIdx.x * 4096) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const float r = (float)(h & 0xFFFFFF) / 8388608.f - 1.f;
            ra[0][i][e] = r; ra[1][i][e] = -r; rw[i][e] = 1.f + 0.2f * r; rbb[i][e] = 0.3f + 0.5f * r; rb[i][e] = 0.05f * r;
        }
    auto load_A = [&](int par) {
        if (OPS & 1) for (int i = 0; i < 2; ++i) ra[par][i] = ld4(sx, vaA[i], cA * 4);
        cA += 32;
        if (cA >= C) { cA = 0; ++tapA; offs(tapA); vaA[0] = va[0]; vaA[1] = va[1]; }
    };
    auto load_tile = [&]() {
        for (int i = 0; i < 2; ++i) {
            if (OPS & 2) { rw[i] = ld4(slw, vaL[i], c * 4); rbb[i] = ld4(slb, vaL[i], c * 4); }
            if (OPS & 4) rb[i] = ld4(sw, vb[i], (tap * C + c) * 4);
        }
        c += 32;
        if (c >= C) { c = 0; ++tap; offs(tap); vaL[0] = va[0]; vaL[1] = va[1]; }
    };
    auto store_tile = [&](float *Ad, float *Bd, int par) {
        if (!(OPS & 8)) return;
        for (int i = 0; i < 2; ++i) {
            f32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf((ra[par][i][e] - mu) * rs, rw[i][e], rbb[i][e]), 0.f);
            *reinterpret_cast<f32x4 *>(&Ad[(rowq + 64 * i) * LDK + col4 * 4]) = v;
            *reinterpret_cast<f32x4 *>(&Bd[(rowq + 64 * i) * LDK + col4 * 4]) = rb[i];
        }
    };
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (!(OPS & 8)) for (int i = tid; i < 2 * BM * LDK; i += 512) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        As[i] = (float)(h & 0xFFFFFF) / 8388608.f - 1.f; Bs[i] = 0.05f * ((float)((h >> 3) & 0xFFFFF) / 524288.f - 1.f);
    }
    load_A(0);
    load_tile();
    store_tile(As, Bs, 0);
    if (nk > 1) load_A(1);
    if (nk > 2) load_A(0);
    __syncthreads();
    // PF: the barrier moves in front of the last quarter's MFMAs and the next K-tile's first fragments are read right
    // after it into a second register set: barrier wait and ds_read latency hide behind eight MFMAs per wave
    f32x4 fr[2][3];
    auto frags = [&](const float *Ac, const float *Bc, int kk, int set) {
        fr[set][0] = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + l31) * LDK + kk * 8 + lhalf * 4]);
        fr[set][1] = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + 32 + l31) * LDK + kk * 8 + lhalf * 4]);
        fr[set][2] = *reinterpret_cast<const f32x4 *>(&Bc[(wn * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
    };
    if (PF) frags(As, Bs, 0, 0);
    auto ktile = [&](int kt, int PB) {
        const float *Ac = As + PB * (BM * LDK), *Bc = Bs + PB * (BN * LDK);
        float *An = As + (PB ^ 1) * (BM * LDK), *Bn = Bs + (PB ^ 1) * (BN * LDK);
        const bool more = kt + 1 < nk;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (PF) {
                if (kk < 3) frags(Ac, Bc, kk + 1, (kk + 1) & 1);
                if (kk == 0) { if (more) load_tile(); __builtin_amdgcn_sched_barrier(0); }
                if (kk == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) store_tile(An, Bn, PB ^ 1);
                    if (kt + 3 < nk) load_A(PB ^ 1);
                    __syncthreads();
                    if (more) frags(An, Bn, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[kk & 1][2][s], fr[kk & 1][0][s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[kk & 1][2][s], fr[kk & 1][1][s], acc[1], 0, 0, 0);
                }
                continue;
            }
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + l31) * LDK + kk * 8 + lhalf * 4]);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&Bc[(wn * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            if (kk == 0) { if (more) load_tile(); __builtin_amdgcn_sched_barrier(0); }
            if (kk == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) store_tile(An, Bn, PB ^ 1);
                if (kt + 3 < nk) load_A(PB ^ 1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a1[s], acc[1], 0, 0, 0);
            }
        }
        if (!PF) __syncthreads();
    };
    const unsigned long long t_loop0 = __builtin_readcyclecounter();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) { ktile(kt, 0); ktile(kt + 1, 1); }
    if (kt < nk) ktile(kt, 0);

    if (PF) __syncthreads();
    // epilogue
    const unsigned long long t_loop1 = __builtin_readcyclecounter();
    const __amdgpu_buffer_rsrc_t sy = srd(y + (int64_t)m0 * BN, (uint64_t)BM * BN * 4);
    float a1s[2] = {0.f, 0.f}, a2s[2] = {0.f, 0.f};
    if (EPI == 1) {
        for (int i = 0; i < 2; ++i) {
            const unsigned ro = (unsigned)(wm * 64 + i * 32 + l31) * (BN * 4u);
            for (int g = 0; g < 4; ++g) {
                f32x4 z4;
                for (int e = 0; e < 4; ++e) { const float z = acc[i][4 * g + e] + 0.5f; a1s[i] += z; a2s[i] = fmaf(z, z, a2s[i]); z4[e] = z; }
                st4(sy, ro + (unsigned)(wn * 32 + 8 * g + 4 * lhalf) * 4u, z4);
            }
        }
    } else if (EPI == 2) {
        // C tile through LDS ([128][132] floats = 67.6 KB over As/Bs), then whole 512 B rows per half-wave
        constexpr int LDC = 132;
        float *Cs = smem;
        for (int i = 0; i < 2; ++i)
            for (int g = 0; g < 4; ++g) {
                f32x4 z4;
                for (int e = 0; e < 4; ++e) { const float z = acc[i][4 * g + e] + 0.5f; a1s[i] += z; a2s[i] = fmaf(z, z, a2s[i]); z4[e] = z; }
                *reinterpret_cast<f32x4 *>(&Cs[(wm * 64 + i * 32 + l31) * LDC + wn * 32 + 8 * g + 4 * lhalf]) = z4;
            }
        __syncthreads();
        for (int it = 0; it < 8; ++it) {
            const int row = it * 16 + wave * 2 + lhalf;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&Cs[row * LDC + l31 * 4]);
            st4(sy, (unsigned)row * (BN * 4u) + l31 * 16u, v);
        }
        __syncthreads();
    } else {
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) a1s[i] += acc[i][r];
    }
    if (STATS) {
        float *red = smem + 128 * 132;      // past the C tile
        for (int i = 0; i < 2; ++i) {
            float a1 = a1s[i] + __shfl_xor(a1s[i], 32, 64), a2 = a2s[i] + __shfl_xor(a2s[i], 32, 64);
            if (lhalf == 0) { red[(wm * 64 + i * 32 + l31) * 4 + wn] = a1; red[512 + (wm * 64 + i * 32 + l31) * 4 + wn] = a2; }
        }
        __syncthreads();
        float t1 = 0.f, t2 = 0.f;
        if (tid < 128) {
            for (int q = 0; q < 4; ++q) { t1 += red[tid * 4 + q]; t2 += red[512 + tid * 4 + q]; }
            for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
            if (lane == 0) { part[(blockIdx.x * 2 + wave) * 2] = t1; part[(blockIdx.x * 2 + wave) * 2 + 1] = t2; }
        }
    } else if (a1s[0] + a1s[1] == 12345.678f) part[0] = a1s[0];
    if (ts != nullptr && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long *o = ts + (size_t)blockIdx.x * 6;
        o[0] = t_start; o[1] = t_loop0; o[2] = t_loop1; o[3] = __builtin_readcyclecounter();
        o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(6 | (31 << 11)) << 32);
    }
}

// Persistent variant: gridDim.x workgroups walk the tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the operand stream runs
// straight across tile boundaries (the next tile's first K-tiles are requested and staged under the current tile's last
// ones; its second K-tile's L2 operands are requested BEFORE the epilogue's stores so that no later wait sits behind
// them in the in-order vmcnt queue).  delay > 0: the workgroup whose LDS allocation does not start at 0 idles that many
// cycles first, so that the two co-resident workgroups run half a tile out of phase.
template <bool STATS>
__global__ __launch_bounds__(512, 4) void kpers(const float *__restrict__ x, const float *__restrict__ lw, const float *__restrict__ lb,
                                                const float *__restrict__ w, const float *__restrict__ stats, float *__restrict__ y,
                                                float *__restrict__ part, int nk, int rps_out, int ntiles, int delay, unsigned long long *ts, int tstride) {
    constexpr int LDK = 36, BM = 128, BN = 128, C = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem, *Bs = smem + 2 * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhalf = lane >> 5, col4 = tid & 7, rowq = tid >> 3;
    if (delay > 0 && __builtin_amdgcn_s_getreg(6 | (7 << 11)) != 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)delay) __builtin_amdgcn_s_sleep(16);
    }
    const int rps_in = 2 * rps_out;
    const __amdgpu_buffer_rsrc_t slw = srd(lw, (uint64_t)rps_in * C * 4), slb = srd(lb, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t sw = srd(w, (uint64_t)BN * nk * 32 * 4);
    unsigned vb[2];
    for (int i = 0; i < 2; ++i) vb[i] = (unsigned)(rowq + 64 * i) * (unsigned)(nk * 32) * 4u + col4 * 16u;
    // activation cursor (runs two to three K-tiles ahead of the L2 cursor)
    int tileA = blockIdx.x, tapA = 0, cA = 0, rowA[2];
    unsigned vaA[2];
    __amdgpu_buffer_rsrc_t sxA;
    auto offsA = [&]() {
        for (int i = 0; i < 2; ++i) vaA[i] = (unsigned)(rowA[i] + tapA) < (unsigned)rps_in ? (unsigned)((rowA[i] + tapA) * C + col4 * 4) * 4u : 0x80000000u;
    };
    auto setA = [&]() {
        const int m0 = tileA * BM, b = __builtin_amdgcn_readfirstlane(m0 / rps_out), r0 = __builtin_amdgcn_readfirstlane(m0 % rps_out);
        const uint64_t pa = (uint64_t)(x + (int64_t)b * rps_in * C);
        const uint64_t pu = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pa);
        sxA = srd((const float *)pu, (uint64_t)rps_in * C * 4);
        for (int i = 0; i < 2; ++i) rowA[i] = 2 * (r0 + rowq + 64 * i) - 1;
        offsA();
    };
    int tileL = blockIdx.x, tapL = 0, cL = 0, rowL[2];
    float mu = 0.f, rs = 0.f, mu_n = 0.f, rs_n = 0.f;
    bool validS = true;
    unsigned vaL[2];
    auto offsL = [&]() {
        for (int i = 0; i < 2; ++i) vaL[i] = (unsigned)(rowL[i] + tapL) < (unsigned)rps_in ? (unsigned)((rowL[i] + tapL) * C + col4 * 4) * 4u : 0x80000000u;
    };
    auto setL = [&]() {
        const int r0 = __builtin_amdgcn_readfirstlane((tileL * BM) % rps_out);
        const int b = __builtin_amdgcn_readfirstlane((tileL * BM) / rps_out);
        mu_n = stats[2 * b]; rs_n = stats[2 * b + 1];
        for (int i = 0; i < 2; ++i) rowL[i] = 2 * (r0 + rowq + 64 * i) - 1;
        offsL();
    };
    f32x4 ra[2][2], rw[2], rbb[2], rb[2];
    auto load_A = [&](int par) {
        if (__builtin_amdgcn_readfirstlane(tileA >= ntiles ? 1 : 0)) return;
        for (int i = 0; i < 2; ++i) ra[par][i] = ld4(sxA, vaA[i], __builtin_amdgcn_readfirstlane(cA * 4));
        cA = __builtin_amdgcn_readfirstlane(cA + 32);
        if (cA >= C) {
            cA = 0;
            tapA = __builtin_amdgcn_readfirstlane(tapA + 1);
            if (tapA == 3) { tapA = 0; tileA = __builtin_amdgcn_readfirstlane(tileA + tstride); if (tileA < ntiles) setA(); } else offsA();
        }
    };
    auto load_tile = [&]() {
        if (!validS) return;
        if ((cL | tapL) == 0) { mu = mu_n; rs = rs_n; }
        for (int i = 0; i < 2; ++i) {
            rw[i] = ld4(slw, vaL[i], __builtin_amdgcn_readfirstlane(cL * 4)); rbb[i] = ld4(slb, vaL[i], __builtin_amdgcn_readfirstlane(cL * 4));
            rb[i] = ld4(sw, vb[i], __builtin_amdgcn_readfirstlane((tapL * C + cL) * 4));
        }
        cL = __builtin_amdgcn_readfirstlane(cL + 32);
        if (cL >= C) {
            cL = 0;
            tapL = __builtin_amdgcn_readfirstlane(tapL + 1);
            if (tapL == 3) { tapL = 0; tileL = __builtin_amdgcn_readfirstlane(tileL + tstride); validS = tileL < ntiles; if (validS) setL(); } else offsL();
        }
    };
    auto store_tile = [&](float *Ad, float *Bd, int par) {
        if (!validS) return;
        for (int i = 0; i < 2; ++i) {
            f32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf((ra[par][i][e] - mu) * rs, rw[i][e], rbb[i][e]), 0.f);
            *reinterpret_cast<f32x4 *>(&Ad[(rowq + 64 * i) * LDK + col4 * 4]) = v;
            *reinterpret_cast<f32x4 *>(&Bd[(rowq + 64 * i) * LDK + col4 * 4]) = rb[i];
        }
    };
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    setA(); setL();
    load_A(0);
    load_tile();
    store_tile(As, Bs, 0);
    load_A(1);
    load_A(0);
    __syncthreads();
    load_tile();
    auto ktile = [&](int PB) {
        const float *Ac = As + PB * (BM * LDK), *Bc = Bs + PB * (BN * LDK);
        float *An = As + (PB ^ 1) * (BM * LDK), *Bn = Bs + (PB ^ 1) * (BN * LDK);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + l31) * LDK + kk * 8 + lhalf * 4]);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(&Ac[(wm * 64 + 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&Bc[(wn * 32 + l31) * LDK + kk * 8 + lhalf * 4]);
            if (kk == 3) {
                __builtin_amdgcn_sched_barrier(0);
                store_tile(An, Bn, PB ^ 1);
                load_A(PB ^ 1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a1[s], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();
        load_tile();
        __builtin_amdgcn_sched_barrier(0);
    };
    unsigned long long c_loop = 0, c_epi = 0, c_first = 0, n_t = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += tstride) {
        const unsigned long long q0 = __builtin_readcyclecounter();
        ktile(0);
        const unsigned long long q1 = __builtin_readcyclecounter();
        ktile(1);
        for (int kt = 2; kt < nk; kt += 2) { ktile(0); ktile(1); }
        const unsigned long long q2 = __builtin_readcyclecounter();
        const __amdgpu_buffer_rsrc_t sy = srd(y + (int64_t)tile * BM * BN, (uint64_t)BM * BN * 4);
        float a1s[2] = {0.f, 0.f}, a2s[2] = {0.f, 0.f};
        for (int i = 0; i < 2; ++i) {
            const unsigned ro = (unsigned)(wm * 64 + i * 32 + l31) * (BN * 4u);
            for (int g = 0; g < 4; ++g) {
                f32x4 z4;
                for (int e = 0; e < 4; ++e) { const float z = acc[i][4 * g + e] + 0.5f; a1s[i] += z; a2s[i] = fmaf(z, z, a2s[i]); z4[e] = z; acc[i][4 * g + e] = 0.f; }
                st4(sy, ro + (unsigned)(wn * 32 + 8 * g + 4 * lhalf) * 4u, z4);
            }
        }
        if (STATS) {      // per-row sums by cross-lane adds only (LDS is busy with the next tile), one atomic-free slot per wave and row group
            for (int i = 0; i < 2; ++i) {
                float a1 = a1s[i], a2 = a2s[i];
                for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
                if (lane == 0) { part[((size_t)tile * 16 + wave * 2 + i) * 2] = a1; part[((size_t)tile * 16 + wave * 2 + i) * 2 + 1] = a2; }
            }
        } else if (a1s[0] + a1s[1] == 12345.678f) part[0] = a1s[0];
        c_loop += q2 - q0; c_first += q1 - q0; c_epi += __builtin_readcyclecounter() - q2; ++n_t;
    }
    if (ts != nullptr && tid == 0) { ts[blockIdx.x * 4] = c_loop; ts[blockIdx.x * 4 + 1] = c_epi; ts[blockIdx.x * 4 + 2] = c_first; ts[blockIdx.x * 4 + 3] = n_t; }
}

// Three 4-wave workgroups per CU: 64x128 tiles (four 64x32 wave tiles), LDS rows unpadded and XOR-swizzled by the row
// (chunk c of row r lives at chunk c ^ (r & 7)): 2 x (64 + 128) x 128 B = 48 KB per workgroup, 144 KB per CU.  Three
// barrier domains instead of two at the same MFMAs per barrier; the weight tile is re-read per 64 rows instead of 128.
// EPI 1: direct stores; 2: the 64x128 output tile through LDS, whole 512 B rows.
template <int EPI>
__global__ __launch_bounds__(256, 3) void k3(const float *__restrict__ x, const float *__restrict__ lw, const float *__restrict__ lb,
                                             const float *__restrict__ w, const float *__restrict__ stats, float *__restrict__ y,
                                             float *__restrict__ part, int nk, int rps_out) {
    constexpr int BM = 64, BN = 128, C = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem, *Bs = smem + 2 * BM * 32;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int l31 = lane & 31, lhalf = lane >> 5, col4 = tid & 7, rowq = tid >> 3;     // rowq 0..31
    const int m0 = blockIdx.x * BM;
    const int b = m0 / rps_out, r0 = m0 % rps_out;
    const int rps_in = 2 * rps_out;
    const float mu = stats[2 * b], rs = stats[2 * b + 1];
    const __amdgpu_buffer_rsrc_t sx = srd(x + (int64_t)b * rps_in * C, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t slw = srd(lw, (uint64_t)rps_in * C * 4), slb = srd(lb, (uint64_t)rps_in * C * 4);
    const __amdgpu_buffer_rsrc_t sw = srd(w, (uint64_t)BN * nk * 32 * 4);
    unsigned va[2], vb[4];
    int arow[2];
    for (int i = 0; i < 2; ++i) arow[i] = 2 * (r0 + rowq + 32 * i) - 1;
    for (int j = 0; j < 4; ++j) vb[j] = (unsigned)(rowq + 32 * j) * (unsigned)(nk * 32) * 4u + col4 * 16u;
    auto offs = [&](int tap) {
        for (int i = 0; i < 2; ++i) va[i] = (unsigned)(arow[i] + tap) < (unsigned)rps_in ? (unsigned)((arow[i] + tap) * C + col4 * 4) * 4u : 0x80000000u;
    };
    int tapA = 0, cA = 0, tap = 0, c = 0;
    unsigned vaA[2], vaL[2];
    offs(0); vaA[0] = va[0]; vaA[1] = va[1]; vaL[0] = va[0]; vaL[1] = va[1];
    f32x4 ra[2][2], rw[2], rbb[2], rb[4];
    auto load_A = [&](int par) {
        for (int i = 0; i < 2; ++i) ra[par][i] = ld4(sx, vaA[i], cA * 4);
        cA += 32;
        if (cA >= C) { cA = 0; ++tapA; offs(tapA); vaA[0] = va[0]; vaA[1] = va[1]; }
    };
    auto load_tile = [&]() {
        for (int i = 0; i < 2; ++i) { rw[i] = ld4(slw, vaL[i], c * 4); rbb[i] = ld4(slb, vaL[i], c * 4); }
        for (int j = 0; j < 4; ++j) rb[j] = ld4(sw, vb[j], (tap * C + c) * 4);
        c += 32;
        if (c >= C) { c = 0; ++tap; offs(tap); vaL[0] = va[0]; vaL[1] = va[1]; }
    };
    const int sw4 = (col4 ^ (rowq & 7)) * 4;            // (rowq + 32 i) & 7 == rowq & 7
    auto store_tile = [&](float *Ad, float *Bd, int par) {
        for (int i = 0; i < 2; ++i) {
            f32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf((ra[par][i][e] - mu) * rs, rw[i][e], rbb[i][e]), 0.f);
            *reinterpret_cast<f32x4 *>(&Ad[(rowq + 32 * i) * 32 + sw4]) = v;
        }
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(&Bd[(rowq + 32 * j) * 32 + sw4]) = rb[j];
    };
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    load_A(0);
    load_tile();
    store_tile(As, Bs, 0);
    if (nk > 1) load_A(1);
    if (nk > 2) load_A(0);
    __syncthreads();
    auto ktile = [&](int kt, int PB) {
        const float *Ac = As + PB * (BM * 32), *Bc = Bs + PB * (BN * 32);
        float *An = As + (PB ^ 1) * (BM * 32), *Bn = Bs + (PB ^ 1) * (BN * 32);
        const bool more = kt + 1 < nk;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = ((kk * 2 + lhalf) ^ (l31 & 7)) * 4;
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&Ac[l31 * 32 + ch]);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(&Ac[(32 + l31) * 32 + ch]);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&Bc[(wn * 32 + l31) * 32 + ch]);
            if (kk == 0) { if (more) load_tile(); __builtin_amdgcn_sched_barrier(0); }
            if (kk == 3) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) store_tile(An, Bn, PB ^ 1);
                if (kt + 3 < nk) load_A(PB ^ 1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a0[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a1[s], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) { ktile(kt, 0); ktile(kt + 1, 1); }
    if (kt < nk) ktile(kt, 0);
    const __amdgpu_buffer_rsrc_t sy = srd(y + (int64_t)m0 * BN, (uint64_t)BM * BN * 4);
    float a1s[2] = {0.f, 0.f}, a2s[2] = {0.f, 0.f};
    constexpr int LDC = 132;
    float *Cs = smem;
    for (int i = 0; i < 2; ++i) {
        const unsigned ro = (unsigned)(i * 32 + l31) * (BN * 4u);
        for (int g = 0; g < 4; ++g) {
            f32x4 z4;
            for (int e = 0; e < 4; ++e) { const float z = acc[i][4 * g + e] + 0.5f; a1s[i] += z; a2s[i] = fmaf(z, z, a2s[i]); z4[e] = z; }
            if (EPI == 1) st4(sy, ro + (unsigned)(wn * 32 + 8 * g + 4 * lhalf) * 4u, z4);
            else *reinterpret_cast<f32x4 *>(&Cs[(i * 32 + l31) * LDC + wn * 32 + 8 * g + 4 * lhalf]) = z4;
        }
    }
    float *red = smem + 64 * LDC;
    for (int i = 0; i < 2; ++i) {
        const float a1 = a1s[i] + __shfl_xor(a1s[i], 32, 64), a2 = a2s[i] + __shfl_xor(a2s[i], 32, 64);
        if (lhalf == 0) { red[(i * 32 + l31) * 4 + wn] = a1; red[256 + (i * 32 + l31) * 4 + wn] = a2; }
    }
    __syncthreads();
    if (EPI == 2)
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + wn * 2 + lhalf;
            st4(sy, (unsigned)row * (BN * 4u) + l31 * 16u, *reinterpret_cast<const f32x4 *>(&Cs[row * LDC + l31 * 4]));
        }
    if (tid < 64) {
        float t1 = 0.f, t2 = 0.f;
        for (int q = 0; q < 4; ++q) { t1 += red[tid * 4 + q]; t2 += red[256 + tid * 4 + q]; }
        for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if (lane == 0) { part[blockIdx.x * 2] = t1; part[blockIdx.x * 2 + 1] = t2; }
    }
}

__global__ void fill(float *p, size_t n, float scale, float bias) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = bias + scale * ((float)(h & 0xFFFFFF) / 8388608.f - 1.f);
    }
}

int main(int argc, char **argv) {
    const int rps_out = 1024, B = argc > 1 ? atoi(argv[1]) : 2048, nk = argc > 2 ? atoi(argv[2]) : 12;
    const int prio = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t M = (int64_t)B * rps_out;
    const int ntiles = (int)(M / 128);
    float *x, *lw, *lb, *w, *stats, *y, *part;
    (void)hipMalloc(&x, M * 2 * 128 * 4); (void)hipMalloc(&y, M * 128 * 4);
    (void)hipMalloc(&lw, 2 * rps_out * 128 * 4); (void)hipMalloc(&lb, 2 * rps_out * 128 * 4);
    (void)hipMalloc(&w, 128 * nk * 32 * 4); (void)hipMalloc(&stats, B * 8); (void)hipMalloc(&part, (size_t)ntiles * 128 + 64);
    // random operands: an all-zero GEMM draws less power and clocks higher than the real thing
    fill<<<4096, 256>>>(x, (size_t)M * 2 * 128, 1.f, 0.f); fill<<<256, 256>>>(lw, (size_t)2 * rps_out * 128, 0.2f, 1.f);
    fill<<<256, 256>>>(lb, (size_t)2 * rps_out * 128, 0.5f, 0.3f); fill<<<64, 256>>>(w, (size_t)128 * nk * 32, 0.05f, 0.f);
    fill<<<16, 256>>>(stats, (size_t)B * 2, 0.01f, 1.f);
    const size_t lds = getenv("ONE_WG") ? 100 * 1024 : (128 * 132 + 1024 + 64) * 4;     // ONE_WG=1: one workgroup per CU
    const double flop = 2.0 * M * 128 * nk * 32;
    printf("M = %lld rows, N = 128, K = %d, %d tiles, LDS %zu B\n", (long long)M, nk * 32, ntiles, lds);
#define T(name, EPI, OPS, ST) TT(name, EPI, OPS, ST, false)
#define TT(name, EPI, OPS, ST, PF) { auto kp = k<EPI, OPS, ST, PF>; \
        (void)hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); \
        hipLaunchKernelGGL(kp, dim3(ntiles), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, nk, rps_out, prio, nullptr); (void)hipDeviceSynchronize(); \
        (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kp, dim3(ntiles), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, nk, rps_out, prio, nullptr); \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3; \
        printf("%-52s %8.3f ms  %6.1f TFLOP/s  %.3f  %s\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError())); }
    T("loop only (no operands, no epilogue)", 0, 0, false);
    T("+ refill/transform (stale regs)", 0, 8, false);
    T("+ activation loads (HBM)", 0, 9, false);
    T("+ LN-affine loads (L2)", 0, 11, false);
    T("+ weight loads (L2) = full loop", 0, 15, false);
    T("full loop + direct stores", 1, 15, false);
    T("full loop + direct stores + stats", 1, 15, true);
    T("full loop + LDS-staged row stores", 2, 15, false);
    T("full loop + LDS-staged row stores + stats", 2, 15, true);
    TT("PF: + refill/transform (stale regs)", 0, 8, false, true);
    TT("PF: full loop", 0, 15, false, true);
    TT("PF: full loop + direct stores + stats", 1, 15, true, true);
    TT("PF: full loop + LDS-staged row stores + stats", 2, 15, true, true);
    T("no operands + direct stores", 1, 0, false);
    T("no operands + LDS-staged row stores", 2, 0, false);
    for (int epi = 1; epi <= 2; ++epi) {
        const size_t lds3 = 48 * 1024;
        auto kq = epi == 1 ? k3<1> : k3<2>;
        (void)hipFuncSetAttribute((const void *)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int nt3 = (int)(M / 64);
        hipLaunchKernelGGL(kq, dim3(nt3), dim3(256), lds3, 0, x, lw, lb, w, stats, y, part, nk, rps_out); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kq, dim3(nt3), dim3(256), lds3, 0, x, lw, lb, w, stats, y, part, nk, rps_out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        printf("3 x 4-wave workgroups/CU, 64x128 tiles, %-22s %8.3f ms  %6.1f TFLOP/s  %.3f  %s\n", epi == 1 ? "direct stores + stats" : "row stores + stats", ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
    }
    unsigned long long *pts; (void)hipMalloc(&pts, 512 * 32);
    for (int delay = 0; delay <= 30000; delay += 30000) {
        auto kq = kpers<true>;
        (void)hipFuncSetAttribute((const void *)kq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int grid = getenv("ONE_WG") ? 256 : 512;
        hipLaunchKernelGGL(kq, dim3(grid), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, nk, rps_out, ntiles, delay, pts, grid); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kq, dim3(grid), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, nk, rps_out, ntiles, delay, pts, grid);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        { unsigned long long h[2048]; (void)hipMemcpy(h, pts, grid * 32, hipMemcpyDeviceToHost); double a = 0, b2 = 0, c2 = 0, n = 0;
          for (int i = 0; i < grid; ++i) { a += h[4 * i]; b2 += h[4 * i + 1]; c2 += h[4 * i + 2]; n += h[4 * i + 3]; }
          printf("  per tile: loop %.0f cycles (first K-tile %.0f), epilogue %.0f\n", a / n, c2 / n, b2 / n); }
        printf("persistent, direct stores + stats, delay %-6d        %8.3f ms  %6.1f TFLOP/s  %.3f  %s\n", delay, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, hipGetErrorString(hipGetLastError()));
    }
    if (getenv("TS_OUT")) {       // per-tile phase timestamps of the full variant -> binary file (tools/ubench/gemm_tile_phases.py)
        unsigned long long *ts; (void)hipMalloc(&ts, (size_t)ntiles * 48);
        auto kp = k<1, 15, true>;
        hipLaunchKernelGGL(kp, dim3(ntiles), dim3(512), lds, 0, x, lw, lb, w, stats, y, part, nk, rps_out, prio, ts);
        (void)hipDeviceSynchronize();
        unsigned long long *h = (unsigned long long *)malloc((size_t)ntiles * 48);
        (void)hipMemcpy(h, ts, (size_t)ntiles * 48, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("TS_OUT"), "wb"); fwrite(h, 48, ntiles, f); fclose(f);
    }
    return 0;
}
