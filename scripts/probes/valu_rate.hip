// probe: issue cost of the epilogue's vector instructions with NOTHING beside them (one wave per SIMD, no MFMA, no memory):
// v_fma_f32, v_pk_fma_f32 (two fp32 FMAs per lane), v_max_f32, v_cvt_pkrtz_f16_f32, v_fma_mix_f32, v_max3_f32 -- ns and clocks per instruction,
// independent chains of 8.   hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_rate.hip -o scripts/probes/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(int iters, float *sink, unsigned long long *clk) {
    float a[8];
    f32x2 p[8];
    unsigned h[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.f + i + threadIdx.x * 1e-3f; p[i] = f32x2{a[i], a[i] + 0.5f}; h[i] = i; }
    const float c = 0.999f;
    const f32x2 c2 = {0.999f, 0.998f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
                if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(c));
                if (OP == 4) asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(a[i]) : "v"(c), "v"(h[i]));
                if (OP == 5) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a[i]) : "v"(c), "v"(p[i].x));
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + h[i];
    if (s == 12345.6789f) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int OP>
static void run(const char *name, float *sink, unsigned long long *clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, 100, sink, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, iters, sink, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%-22s %6.2f ns per instruction and wave (one wave per SIMD, 256 CUs)\n", name, ms * 1e6 / ((double)iters * 64));
}
int main() {
    float *sink;
    unsigned long long *clk;
    hipMalloc(&sink, 4);
    hipMalloc(&clk, 8);
    run<0>("v_fma_f32", sink, clk);
    run<1>("v_pk_fma_f32", sink, clk);
    run<2>("v_max_f32", sink, clk);
    run<3>("v_cvt_pkrtz_f16_f32", sink, clk);
    run<4>("v_fma_mix_f32", sink, clk);
    run<5>("v_max3_f32 |.|", sink, clk);
    run<6>("v_pk_mul_f32", sink, clk);
    run<7>("v_pk_add_f32", sink, clk);
    return 0;
}
