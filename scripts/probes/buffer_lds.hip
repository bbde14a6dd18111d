// probe: buffer_load_dwordx4 ... offen lds on gfx950 -- (1) the LDS destination is M0 + lane * 16 (also beyond 64 KB), (2) a lane whose
// voffset is >= num_records writes ZEROS (the halo's SAME padding without a zero line or a per-lane branch), (3) soffset is added to
// the address but not range-checked.  conv_c4_kernel (unet_c4.hip) relies on all three.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/buffer_lds.hip -o scripts/probes/buffer_lds && scripts/probes/buffer_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned *x, unsigned *y, unsigned lds_dst, unsigned soff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 40960; i += 64) reinterpret_cast<unsigned *>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned long long)x);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned long long)x >> 32) & 0xffff);
    r.z = 0xFFF00000;
    r.w = 0x00020000;
    const unsigned lane = threadIdx.x;
    // lanes 0-47: 16 bytes each from x + lane * 32 (a gather); lanes 48-63: out of range
    const unsigned voff = lane < 48 ? lane * 32 : 0xFFF00000u + lane * 16;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(lds_dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) y[lane * 4 + i] = reinterpret_cast<unsigned *>(smem + lds_dst)[lane * 4 + i];
    y[256 + lane] = reinterpret_cast<unsigned *>(smem + lds_dst)[256 + lane];  // the KB after the destination: untouched?
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *x, *y;
    hipMalloc(&x, 4096 * 4);
    hipMalloc(&y, 512 * 4);
    hipMemcpy(x, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    int bad = 0;
    for (unsigned dst : {1024u, 70000u - 70000u % 16, 150000u - 150000u % 16}) {
        for (unsigned soff : {0u, 64u}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, x, y, dst, soff);
            std::vector<unsigned> o(512);
            hipMemcpy(o.data(), y, 512 * 4, hipMemcpyDeviceToHost);
            int b = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const unsigned want = lane < 48 ? (unsigned)(lane * 8 + soff / 4 + i) : 0u;
                    if (o[lane * 4 + i] != want) ++b;
                }
            for (int i = 0; i < 64; ++i)
                if (o[256 + i] != 0xdeadbeefu) ++b;
            printf("lds dst %6u soffset %2u: %s (lane 0 -> %u %u.., lane 47 -> %u, lane 48 (out of range) -> %u)\n", dst, soff,
                   b ? "MISMATCH" : "ok", o[0], o[1], o[47 * 4], o[48 * 4]);
            bad += b;
        }
    }
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad != 0;
}
