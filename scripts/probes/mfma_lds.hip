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

template <int NW, int RT, int CT, int DMA>
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
    for (int s = 0; s < steps; ++s) {
        load(f1, 2 * s + 1);
        __builtin_amdgcn_sched_barrier(0x07F);  // (the reads may not sink to their first use)
        mma(f0);
#pragma unroll
        for (int q = 0; q < DMA; ++q) dma16(gp + ((s * DMA + q) & 3) * 1024, ring + (unsigned)(q * 1024) % (32768u / NW));
        load(f0, 2 * s + 2);
        __builtin_amdgcn_sched_barrier(0x07F);
        mma(f1);
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
    if (t == 12345.678f) out[0] = t;
}

template <int NW, int RT, int CT, int DMA = 0>
static void run(const char *tag, const uint4 *d, float *out, int cus) {
    auto fn = step_loop<NW, RT, CT, DMA>;
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
    return 0;
}
