// probe: the step body of the parity-mode convolution reduced to its two pipes -- per k-step a wave reads 2 (RT + CT) operand
// fragments of 1 KB from LDS (ds_read_b128, conflict-free rows) and issues 3 RT CT MFMAs (hi hi + hi lo + lo hi), double-buffered
// fragments, one s_barrier per step of two k-steps, random fp16 data -- for the tilings
//   A  8 waves per CU (2 per SIMD), RT 2 x CT 2   (conv_s3_kernel<128> today: 0.67 KB of LDS reads per MFMA)
//   B  4 waves per CU (1 per SIMD), RT 4 x CT 2   (0.50 KB per MFMA, 128 accumulator registers)
//   C  4 waves per CU (1 per SIMD), RT 4 x CT 4   (0.33 KB per MFMA, 256 accumulator registers)
// No DMA, no epilogue: what do LDS reads + barrier alone leave of the 1.67 PF that MFMAs alone sustain (mfma_peak.hip)?
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_lds.hip -o scripts/probes/mfma_lds && scripts/probes/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int RT, int CT>
struct Frags {
    f16x8 ah[RT], al[RT], bh[CT], bl[CT];
};

__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst_wave_base) : "memory");
}
typedef __attribute__((address_space(3))) void lds_void;

template <int NW, int RT, int CT, int DMA, int JUNK = 0>
__global__ __launch_bounds__(NW * 64) void step_loop(const uint4 *__restrict__ src, int steps, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int kBytes = 96 * 1024;
    for (int i = threadIdx.x; i < kBytes / 16; i += NW * 64) reinterpret_cast<uint4 *>(lds)[i] = src[i & 65535];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragment rows of 1 KB (64 lanes x 16 B, conflict-free); the base moves every k-step like the taps / chunks of the real loop
    auto load = [&](Frags<RT, CT> &f, int ks) {
        const unsigned base = (unsigned)((ks * 4096 + wave * 1024) & (kBytes - 32768 - 1)) + lane * 16;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            f.ah[i] = *reinterpret_cast<const f16x8 *>(lds + base + i * 2048);
            f.al[i] = *reinterpret_cast<const f16x8 *>(lds + base + i * 2048 + 1024);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            f.bh[j] = *reinterpret_cast<const f16x8 *>(lds + base + 16384 + j * 2048);
            f.bl[j] = *reinterpret_cast<const f16x8 *>(lds + base + 16384 + j * 2048 + 1024);
        }
    };
    auto mma = [&](const Frags<RT, CT> &f) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
        }
    };
    Frags<RT, CT> f0, f1;
    load(f0, 0);
    // DMA: per step and wave DMA requests of 1 KB (global -> LDS, a 16 KB ring region behind the fragment area), streamed from a
    // 4 MB window of the source (L2-resident), counted wait: the requests of the previous step have landed at the barrier
    const unsigned ring = (unsigned)(size_t)(lds_void *)(lds + kBytes - 32768) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * (DMA > 0 ? 32768 / NW : 0);
    const unsigned char *gp = reinterpret_cast<const unsigned char *>(src) + (size_t)(blockIdx.x & 63) * 65536 + wave * 4096 + lane * 16;
    // JUNK: the bookkeeping instructions of the real kernel's step (profile of conv_s3_kernel<128>'s ISA: per step and wave ~40 VALU
    // address ops, ~36 SALU ops, 4 uniform branches), as opaque asm blocks of 5 that the scheduler may place between the MFMAs
    unsigned jv = lane, js = (unsigned)steps;
    auto junk_valu = [&]() {
        asm volatile("v_add_u32 %0, %0, %1\n\tv_lshlrev_b32 %0, 1, %0\n\tv_xor_b32 %0, %0, %1\n\tv_lshrrev_b32 %0, 1, %0\n\tv_and_or_b32 %0, %0, 64, %1" : "+v"(jv) : "v"(lane));
    };
    auto junk_salu = [&]() {
        asm volatile("s_add_i32 %0, %0, 3\n\ts_lshl_b32 %0, %0, 1\n\ts_and_b32 %0, %0, 0xffff\n\ts_xor_b32 %0, %0, 5\n\ts_addk_i32 %0, 0x40" : "+s"(js) : : "scc");
    };
    for (int s = 0; s < steps; ++s) {
        load(f1, 2 * s + 1);
        __builtin_amdgcn_sched_barrier(0x07F);  // (the reads may not sink to their first use)
        if (JUNK & 1) { junk_valu(); junk_valu(); junk_valu(); junk_valu(); }
        if (JUNK & 2) { junk_salu(); junk_salu(); junk_salu(); junk_salu(); }
        mma(f0);
        if (JUNK & 4) {  // uniform branches in the middle of the step (scheduling regions end there)
            if (__builtin_amdgcn_readfirstlane(js) & 1) asm volatile("s_nop 0");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < DMA; ++q) dma16(gp + ((s * DMA + q) & 3) * 1024, ring + (unsigned)(q * 1024) % (32768u / NW));
        if (JUNK & 4) {
            if (__builtin_amdgcn_readfirstlane(js) & 2) asm volatile("s_nop 0");
            __builtin_amdgcn_sched_barrier(0);
        }
        load(f0, 2 * s + 2);
        __builtin_amdgcn_sched_barrier(0x07F);
        if (JUNK & 1) { junk_valu(); junk_valu(); junk_valu(); junk_valu(); }
        if (JUNK & 2) { junk_salu(); junk_salu(); junk_salu(); }
        mma(f1);
        if (JUNK & 4) {
            if (__builtin_amdgcn_readfirstlane(js) & 4) asm volatile("s_nop 0");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA > 0)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DMA) : "memory");
        else
            asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 12345.678f || (JUNK && jv + js == 0x7fffffffu)) out[0] = t;
}


// PING-PONG variant (round 4): the same per-step work, but the two waves of a SIMD never do the same thing at the same time.
// Waves 0-3 (one per SIMD: a block's waves go to SIMDs cyclically, so w and w + 4 share one) and waves 4-7 run the SAME stream
// offset by one barrier: a LOAD segment (the step's DMA requests, all 2 * 2 (RT + CT) ds_read_b128 of the step, counted wait) and
// a COMPUTE segment (the step's 6 RT CT MFMAs, nothing else), separated by s_barrier -- one half's MFMAs run beside the other
// half's loads, the matrix pipe of a SIMD always belongs to exactly one wave.  PAIR = 0: halves {0-3} / {4-7} (partners share a
// SIMD), PAIR = 1: halves = even / odd waves (the control: both waves of a SIMD in the same role).  SEG = steps per segment.
template <int RT, int CT, int DMA, int PAIR, int SEG>
__global__ __launch_bounds__(512) void pp_loop(const uint4 *__restrict__ src, int steps, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int kBytes = 96 * 1024, NW = 8;
    for (int i = threadIdx.x; i < kBytes / 16; i += NW * 64) reinterpret_cast<uint4 *>(lds)[i] = src[i & 65535];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = PAIR == 0 ? wave >> 2 : wave & 1;
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto load = [&](Frags<RT, CT> &f, int ks) {
        const unsigned base = (unsigned)((ks * 4096 + wave * 1024) & (kBytes - 32768 - 1)) + lane * 16;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            f.ah[i] = *reinterpret_cast<const f16x8 *>(lds + base + i * 2048);
            f.al[i] = *reinterpret_cast<const f16x8 *>(lds + base + i * 2048 + 1024);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            f.bh[j] = *reinterpret_cast<const f16x8 *>(lds + base + 16384 + j * 2048);
            f.bl[j] = *reinterpret_cast<const f16x8 *>(lds + base + 16384 + j * 2048 + 1024);
        }
    };
    auto mma = [&](const Frags<RT, CT> &f) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
    };
    Frags<RT, CT> f[2 * SEG];
    const unsigned ring = (unsigned)(size_t)(lds_void *)(lds + kBytes - 32768) + (unsigned)wave * (DMA > 0 ? 32768 / NW : 0);
    const unsigned char *gp = reinterpret_cast<const unsigned char *>(src) + (size_t)(blockIdx.x & 63) * 65536 + wave * 4096 + lane * 16;
    if (half == 1) asm volatile("s_barrier" ::: "memory");
    for (int s = 0; s < steps; s += SEG) {
        // ---- load segment ----
#pragma unroll
        for (int q = 0; q < DMA * SEG; ++q) dma16(gp + ((s * DMA + q) & 3) * 1024, ring + (unsigned)(q * 1024) % (32768u / NW));
#pragma unroll
        for (int k = 0; k < 2 * SEG; ++k) load(f[k], 2 * s + k);
        __builtin_amdgcn_sched_barrier(0);
        if (DMA > 0)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(DMA * SEG) : "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- compute segment ----
#pragma unroll
        for (int k = 0; k < 2 * SEG; ++k) mma(f[k]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (half == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 12345.678f) out[0] = t;
}

template <int RT, int CT, int DMA, int PAIR, int SEG>
static void run_pp(const char *tag, const uint4 *d, float *out, int cus) {
    auto fn = pp_loop<RT, CT, DMA, PAIR, SEG>;
    const size_t lds = 96 * 1024;
    constexpr int NW = 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int steps = 24 * 48 * 1024 / (RT * CT * NW);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(cus), dim3(NW * 64), lds, 0, d, 64, out);
    hipDeviceSynchronize();
    float best = 1e30f, ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(cus), dim3(NW * 64), lds, 0, d, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("launch failed\n");
    const double mfmas = (double)cus * NW * steps * 2 * 3 * RT * CT;
    printf("%-58s %8.2f ms  %7.1f TFLOP/s   (%d steps)\n", tag, best, mfmas * 32 * 32 * 16 * 2 / (best * 1e-3) / 1e12, steps);
}

// K = 16 STEPS (round 4): the step of the 512-pixel x 128-channel tiling -- one tap of a 16-channel half-chunk = ONE k-step:
// 3 RT CT MFMAs, 2 (RT + CT) reads, DMA requests, the junk of JUNK, and a barrier every BAR k-steps (BAR = 1: weight slots of one
// tap; 3: of a tap row).  4 waves x (4 x 4) = one wave per SIMD with the 256 accumulators of a 128 x 128 wave tile.
template <int NW, int RT, int CT, int DMA, int JUNK, int BAR>
__global__ __launch_bounds__(NW * 64) void k16_loop(const uint4 *__restrict__ src, int ksteps, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int kBytes = 96 * 1024;
    for (int i = threadIdx.x; i < kBytes / 16; i += NW * 64) reinterpret_cast<uint4 *>(lds)[i] = src[i & 65535];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto load = [&](Frags<RT, CT> &f, int ks) {
        const unsigned base = (unsigned)((ks * 4096 + wave * 1024) & (kBytes - 32768 - 1)) + lane * 16;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            f.ah[i] = *reinterpret_cast<const f16x8 *>(lds + base + i * 2048);
            f.al[i] = *reinterpret_cast<const f16x8 *>(lds + base + i * 2048 + 1024);
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            f.bh[j] = *reinterpret_cast<const f16x8 *>(lds + base + 16384 + j * 2048);
            f.bl[j] = *reinterpret_cast<const f16x8 *>(lds + base + 16384 + j * 2048 + 1024);
        }
    };
    auto mma = [&](const Frags<RT, CT> &f) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
        }
    };
    unsigned jv = lane, js = (unsigned)ksteps;
    auto junk_valu = [&]() {
        asm volatile("v_add_u32 %0, %0, %1\n\tv_lshlrev_b32 %0, 1, %0\n\tv_xor_b32 %0, %0, %1\n\tv_lshrrev_b32 %0, 1, %0\n\tv_and_or_b32 %0, %0, 64, %1" : "+v"(jv) : "v"(lane));
    };
    auto junk_salu = [&]() {
        asm volatile("s_add_i32 %0, %0, 3\n\ts_lshl_b32 %0, %0, 1\n\ts_and_b32 %0, %0, 0xffff\n\ts_xor_b32 %0, %0, 5\n\ts_addk_i32 %0, 0x40" : "+s"(js) : : "scc");
    };
    Frags<RT, CT> f0, f1;
    load(f0, 0);
    const unsigned ring = (unsigned)(size_t)(lds_void *)(lds + kBytes - 32768) + (unsigned)wave * (DMA > 0 ? 32768 / NW : 0);
    const unsigned char *gp = reinterpret_cast<const unsigned char *>(src) + (size_t)(blockIdx.x & 63) * 65536 + wave * 4096 + lane * 16;
    auto half_step = [&](Frags<RT, CT> &cur, Frags<RT, CT> &nxt, int ks, bool bar) {
        load(nxt, ks + 1);
        __builtin_amdgcn_sched_barrier(0x07F);
        if (JUNK & 1) { junk_valu(); junk_valu(); junk_valu(); junk_valu(); }
        if (JUNK & 2) { junk_salu(); junk_salu(); junk_salu(); junk_salu(); }
#pragma unroll
        for (int q = 0; q < DMA; ++q) dma16(gp + ((ks * DMA + q) & 3) * 1024, ring + (unsigned)(q * 1024) % (32768u / NW));
        mma(cur);
        if (bar) {
            if (DMA > 0)
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DMA) : "memory");
            else
                asm volatile("s_barrier" ::: "memory");
        }
    };
    for (int ks = 0; ks < ksteps; ks += 2 * BAR) {
#pragma unroll
        for (int u = 0; u < BAR; ++u) {
            half_step(f0, f1, ks + 2 * u, BAR == 1);
            half_step(f1, f0, ks + 2 * u + 1, BAR == 1 || u == BAR - 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) t += acc[i][j][e];
    if (t == 12345.678f || (JUNK && jv + js == 0x7fffffffu)) out[0] = t;
}

template <int NW, int RT, int CT, int DMA, int JUNK, int BAR>
static void run_k16(const char *tag, const uint4 *d, float *out, int cus) {
    auto fn = k16_loop<NW, RT, CT, DMA, JUNK, BAR>;
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int ksteps = 2 * 24 * 48 * 1024 / (RT * CT * NW) / 6 * 6;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(cus), dim3(NW * 64), lds, 0, d, 66, out);
    hipDeviceSynchronize();
    float best = 1e30f, ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(cus), dim3(NW * 64), lds, 0, d, ksteps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("launch failed\n");
    const double mfmas = (double)cus * NW * ksteps * 3 * RT * CT;
    printf("%-58s %8.2f ms  %7.1f TFLOP/s   (%d k-steps)\n", tag, best, mfmas * 32 * 32 * 16 * 2 / (best * 1e-3) / 1e12, ksteps);
}

template <int NW, int RT, int CT, int DMA = 0, int JUNK = 0>
static void run(const char *tag, const uint4 *d, float *out, int cus) {
    auto fn = step_loop<NW, RT, CT, DMA, JUNK>;
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int steps = 24 * 48 * 1024 / (RT * CT * NW);  // the same number of MFMAs per CU in every configuration
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(cus), dim3(NW * 64), lds, 0, d, 64, out);
    hipDeviceSynchronize();
    float best = 1e30f, ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fn, dim3(cus), dim3(NW * 64), lds, 0, d, steps, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("launch failed\n");
    const double mfmas = (double)cus * NW * steps * 2 * 3 * RT * CT;
    const double kb_per_mfma = 2.0 * (RT + CT) / (3.0 * RT * CT);
    printf("%-44s %8.2f ms  %7.1f TFLOP/s   (%.2f KB of LDS reads per MFMA, %d steps)\n", tag, best,
           mfmas * 32 * 32 * 16 * 2 / (best * 1e-3) / 1e12, kb_per_mfma, steps);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<_Float16> h((size_t)8 * 65536 * 8);  // 8 MB: the LDS image comes from the first MB, the DMA stream from the first 4.5
    srand(1);
    for (auto &v : h) {
        float s = 0.f;
        for (int k = 0; k < 12; ++k) s += (float)rand() / RAND_MAX;
        v = (_Float16)(s - 6.f);
    }
    uint4 *d;
    float *out;
    hipMalloc(&d, (size_t)8 * 65536 * 16);
    hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), (size_t)8 * 65536 * 16, hipMemcpyHostToDevice);
    run<8, 2, 2>("A  8 waves / CU, 2 x 2 tiles per wave", d, out, cus);
    run<4, 4, 2>("B  4 waves / CU, 4 x 2 tiles per wave", d, out, cus);
    run<4, 4, 4>("C  4 waves / CU, 4 x 4 tiles per wave", d, out, cus);
    run<8, 4, 2>("D  8 waves / CU, 4 x 2 tiles per wave", d, out, cus);
    run<8, 2, 2, 3>("A + 3 LDS-DMA requests per wave and step", d, out, cus);
    run<4, 4, 2, 6>("B + 6 LDS-DMA requests per wave and step", d, out, cus);
    run<8, 4, 2, 3>("D + 3 LDS-DMA requests per wave and step", d, out, cus);
    run<8, 2, 2>("A  again", d, out, cus);
    run_pp<2, 2, 0, 0, 1>("P  ping-pong halves {0-3}/{4-7}, 2 x 2, no DMA", d, out, cus);
    run_pp<2, 2, 3, 0, 1>("P + 3 LDS-DMA requests per wave and step", d, out, cus);
    run_pp<2, 2, 3, 1, 1>("P + 3 DMA, CONTROL: halves = even / odd waves", d, out, cus);
    run_pp<2, 2, 3, 0, 2>("P + 3 DMA, two steps per segment (48 MFMAs)", d, out, cus);
    run_pp<4, 2, 3, 0, 1>("P 4 x 2 tiles + 3 DMA (48 MFMAs per segment)", d, out, cus);
    run<8, 2, 2, 3>("A + 3 DMA again", d, out, cus);
    run<8, 2, 2, 3, 1>("A + 3 DMA + 40 VALU per step", d, out, cus);
    run<8, 2, 2, 3, 2>("A + 3 DMA + 35 SALU per step", d, out, cus);
    run<8, 2, 2, 3, 3>("A + 3 DMA + 40 VALU + 35 SALU", d, out, cus);
    run<8, 2, 2, 3, 7>("A + 3 DMA + 40 VALU + 35 SALU + 3 branches", d, out, cus);
    run<8, 2, 2, 3, 4>("A + 3 DMA + 3 branches", d, out, cus);
    run<8, 4, 2, 3, 7>("D + 3 DMA + 40 VALU + 35 SALU + 3 branches", d, out, cus);
    run_k16<4, 4, 4, 0, 0, 1>("E  4 waves x (4 x 4), K=16 steps, barrier per step", d, out, cus);
    run_k16<4, 4, 4, 3, 0, 1>("E + 3 DMA per step", d, out, cus);
    run_k16<4, 4, 4, 3, 3, 1>("E + 3 DMA + 20 VALU + 20 SALU per step", d, out, cus);
    run_k16<4, 4, 4, 3, 3, 3>("E + 3 DMA + junk, barrier every 3 steps", d, out, cus);
    run_k16<8, 4, 2, 2, 3, 1>("F  8 waves x (4 x 2), K=16 steps, 2 DMA + junk", d, out, cus);
    run_k16<8, 4, 2, 2, 3, 3>("F  ... barrier every 3 steps", d, out, cus);
    run_k16<8, 2, 2, 2, 3, 1>("G  8 waves x (2 x 2), K=16 steps, 2 DMA + junk", d, out, cus);
    run<8, 2, 2, 3>("A + 3 DMA again", d, out, cus);
    return 0;
}
