// probe: under the power cap, does the ORDER in which a 4 x 4 register tile walks its operand fragments change what the matrix pipes
// sustain?  v_mfma_f32_32x32x16_f16(first, second, acc): 16 accumulators, 4 "first" fragments F[j], 4 "second" fragments S[i], N(0,1) data,
// nothing else in the loop.
//   0  second fixed over four MFMAs, first cycles      (conv_c4_kernel's order: first = weights bh[j], second = activations al[i])
//   1  first fixed over four MFMAs, second cycles
//   2  both change with every MFMA (diagonal walk)
//   3  the same two fragments every time
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_order.hip -o scripts/probes/mfma_order && scripts/probes/mfma_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ORDER>
__global__ __launch_bounds__(256, 1) void k(const f16x8 *__restrict__ src, int iters, float *out) {
    f16x8 F[4], S[4];
    const int lane = (threadIdx.x + blockIdx.x * blockDim.x) & 65535;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        F[i] = src[(size_t)(2 * i) * 65536 + lane];
        S[i] = src[(size_t)(2 * i + 1) * 65536 + lane];
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[j], S[i], acc[i][j], 0, 0, 0);
        } else if (ORDER == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[j], S[i], acc[i][j], 0, 0, 0);
        } else if (ORDER == 2) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][(i + d) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[(i + d) & 3], S[i], acc[i][(i + d) & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F[0], S[0], acc[i][j], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    if (s == 12345.678f) out[0] = s;
}
template <int ORDER>
static void run(const char *name, f16x8 *d, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 120000;
    hipLaunchKernelGGL((k<ORDER>), dim3(256), dim3(256), 0, 0, d, 4000, out);
    hipDeviceSynchronize();
    float best = 1e30f, last = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ORDER>), dim3(256), dim3(256), 0, 0, d, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&last, e0, e1);
        if (last < best) best = last;
    }
    const double mfmas = 256.0 * 4 * iters * 16;
    printf("%-58s best %7.2f ms = %7.1f TFLOP/s (last %7.1f); pipe never idle <=> %.2f GHz\n", name, best, mfmas * 32768 / (best * 1e-3) / 1e12,
           mfmas * 32768 / (last * 1e-3) / 1e12, (double)iters * 16 * 32 / (best * 1e-3) / 1e9);
}
int main() {
    const size_t nfrag = (size_t)8 * 65536;
    std::vector<_Float16> h(nfrag * 8);
    srand(1);
    for (auto &v : h) {
        float s = 0.f;
        for (int q = 0; q < 12; ++q) s += (float)rand() / RAND_MAX;
        v = (_Float16)(s - 6.f);
    }
    f16x8 *d;
    float *out;
    hipMalloc(&d, nfrag * sizeof(f16x8));
    hipMalloc(&out, 4);
    hipMemcpy(d, h.data(), nfrag * sizeof(f16x8), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("0 second operand fixed over 4 MFMAs, first cycles (c4)", d, out);
        run<1>("1 first operand fixed over 4 MFMAs, second cycles", d, out);
        run<2>("2 both change with every MFMA", d, out);
        run<3>("3 the same two fragments every time", d, out);
    }
    // does the rate depend on how many mantissa bits the operands carry?  order 0, N(0,1) values with the low m mantissa bits cleared
    for (int m : {0, 1, 2, 3, 5, 8, 10}) {
        std::vector<_Float16> g(h);
        for (auto &v : g) {
            unsigned short u;
            __builtin_memcpy(&u, &v, 2);
            u &= (unsigned short)(0xFFFFu << m);
            __builtin_memcpy(&v, &u, 2);
        }
        hipMemcpy(d, g.data(), nfrag * sizeof(f16x8), hipMemcpyHostToDevice);
        char name[80];
        snprintf(name, sizeof name, "order 0, low %2d mantissa bits of both operands cleared", m);
        run<0>(name, d, out);
    }
    return 0;
}
