// probe: does a wave keep issuing vector ALU work while its own global stores are being taken by the memory pipe?  One wave per SIMD
// (4 per CU), every wave runs `iters` rounds of { 2 stores of 16 B per lane, 2 V independent v_fma_f32 } in three arrangements:
//   pair    store store  fma x 2V            (what the conv epilogues do: the hi and the lo record piece back to back)
//   spread  store  fma x V  store  fma x V
//   and the two ingredients alone.  If the arrangements cost max(stores, fma) the wave overlaps them; if pair > spread, the second
//   store of a pair waits for the first and holds the wave's in-order issue.
// Stores go to a 64 KB window per wave (L2-resident, default policy): the question is issue, not HBM.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/store_valu.hip -o scripts/probes/store_valu && scripts/probes/store_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define FMA(a, c) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c))
template <int V>
__device__ __forceinline__ void fmas(float (&a)[8], float c) {
#pragma unroll
    for (int i = 0; i < V; ++i) FMA(a[i & 7], c);
}
__device__ __forceinline__ void st(unsigned char *p, const u32x4 &v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_nt(unsigned char *p, const u32x4 &v) { asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }

template <int MODE, int V, bool NT>
__global__ __launch_bounds__(256) void k(unsigned char *out, int iters, float *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *base = out + ((size_t)blockIdx.x * 4 + wave) * 65536 + lane * 16;
    u32x4 v = {(unsigned)lane, 2u, 3u, 4u};
    float a[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const float c = 0.999f + 1e-9f * lane;
    for (int i = 0; i < iters; ++i) {
        unsigned char *p = base + ((i * 2048) & 65535);
        if (MODE == 0) {  // pair
            if (NT) { st_nt(p, v); st_nt(p + 1024, v); } else { st(p, v); st(p + 1024, v); }
            fmas<2 * V>(a, c);
        } else if (MODE == 1) {  // spread
            if (NT) st_nt(p, v); else st(p, v);
            fmas<V>(a, c);
            if (NT) st_nt(p + 1024, v); else st(p + 1024, v);
            fmas<V>(a, c);
        } else if (MODE == 2) {  // fma only
            fmas<2 * V>(a, c);
        } else {  // stores only
            if (NT) { st_nt(p, v); st_nt(p + 1024, v); } else { st(p, v); st(p + 1024, v); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.6789f) sink[0] = s;
}

template <int MODE, int V, bool NT>
static double run(unsigned char *buf, float *sink, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, V, NT>), dim3(blocks), dim3(256), 0, 0, buf, iters, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, V, NT>), dim3(blocks), dim3(256), 0, 0, buf, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;  // ns per round
}
template <int V, bool NT>
static void table(unsigned char *buf, float *sink, int blocks) {
    const int iters = 20000;
    const double pair = run<0, V, NT>(buf, sink, blocks, iters), spread = run<1, V, NT>(buf, sink, blocks, iters);
    const double fma = run<2, V, NT>(buf, sink, blocks, iters), sto = run<3, V, NT>(buf, sink, blocks, iters);
    printf("%3d CUs, %s, 2 stores + %3d fma per round: pair %6.1f ns  spread %6.1f ns  | fma alone %6.1f  stores alone %6.1f  (sum %6.1f, max %6.1f)\n",
           blocks, NT ? "nt     " : "default", 2 * V, pair, spread, fma, sto, fma + sto, fma > sto ? fma : sto);
}
int main() {
    unsigned char *buf;
    float *sink;
    hipMalloc(&buf, (size_t)256 * 4 * 65536);
    hipMalloc(&sink, 4);
    for (int blocks : {8, 256}) {
        table<8, false>(buf, sink, blocks);
        table<16, false>(buf, sink, blocks);
        table<24, false>(buf, sink, blocks);
        table<48, false>(buf, sink, blocks);
        table<24, true>(buf, sink, blocks);
        table<48, true>(buf, sink, blocks);
    }
    return 0;
}
