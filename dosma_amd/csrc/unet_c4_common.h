// unet_c4_common.h -- device helpers shared by the one-wave-per-SIMD convolution kernels of the parity mode
// (unet_c4.hip: conv_c4_kernel; unet_d4.hip: deconv_d4_kernel): LDS-DMA through buffer descriptors, the split of fp32 values
// into fp16 hi + lo parts, the flattened zero-framed image stack.
#ifndef QMRI_UNET_C4_COMMON_H_
#define QMRI_UNET_C4_COMMON_H_
#include <hip/hip_runtime.h>

namespace qmri {
namespace c4 {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 h16x2;  // what v_cvt_pkrtz_f16_f32 returns
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) f16x8 lds_f16x8;

constexpr unsigned kPadOff = 0xFFF00000u;  // a voffset beyond num_records: the lane's 16 bytes arrive as zeros

__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)(lds_void *)p; }

// LDS-DMA of 16 bytes per lane: LDS destination = M0 + lane * 16, source = descriptor base + voffset (per lane, range-checked
// against num_records: beyond it the lane receives zeros) + soffset (scalar, not range-checked).  Inline asm for the same reason
// as unet_s3.hip's dma16: hipcc neither counts nor drains it; every wait in this file is a hand-counted s_waitcnt vmcnt(N).
__device__ __forceinline__ void dma_buf16(unsigned voff, const i32x4 &rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc),
                 "s"(__builtin_amdgcn_readfirstlane((int)soff)), "s"(__builtin_amdgcn_readfirstlane((int)lds_dst))
                 : "memory");
}

__device__ __forceinline__ void nt_store16(void *dst, const uint4 &v) {
    u32x4 t = {v.x, v.y, v.z, v.w};
#ifdef QMRI_PLAIN_STORES  // (A/B build: default-policy stores)
    *reinterpret_cast<u32x4 *>(dst) = t;
#else
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(dst));
#endif
}

// raw buffer descriptor (gfx9 V#): base, stride 0, num_records = kPadOff bytes, DATA_FORMAT = 32 (0x00020000)
__device__ __forceinline__ i32x4 make_rsrc(const void *base) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xffffu));
    r.z = (int)kPadOff;
    r.w = 0x00020000;
    return r;
}

// decode a flat position of the zero-framed image stack: f = R * P + c, R = b * (H + 1) + y + 1, c = x + 1
__device__ __forceinline__ int flat_to_pix(int f, int P, int H, int W, int B) {
    if (f < P) return -1;
    const int R = f / P, c = f - R * P;
    if (c < 1 || c > W) return -1;
    const int r1 = R - 1;
    const int b = r1 / (H + 1), y = r1 - b * (H + 1);
    if (y >= H || b >= B) return -1;
    return (b * H + y) * W + (c - 1);
}

// v - float(hi) for the two halves of a packed fp16 pair, one instruction each (v_fma_mix_f32: v * 1.0 - hi, a single rounding
// like the subtraction it replaces; hipcc emits v_cvt_f32_f16 + v_sub_f32)
__device__ __forceinline__ float sub_hi0(float v, unsigned hpair) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(hpair));
    return r;
}
__device__ __forceinline__ float sub_hi1(float v, unsigned hpair) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(hpair));
    return r;
}
// split four fp32 values into fp16 hi parts (round toward zero, saturating) and lo = rtz(v - hi): 2 x (hi pair, lo pair)
__device__ __forceinline__ void split4(const float (&v)[4], uint2 &hi, uint2 &lo) {
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[0], v[1]));
    const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v[2], v[3]));
    const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(sub_hi0(v[0], h0), sub_hi1(v[1], h0));
    const h16x2 l1 = __builtin_amdgcn_cvt_pkrtz(sub_hi0(v[2], h1), sub_hi1(v[3], h1));
    hi = make_uint2(h0, h1);
    lo = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}

}  // namespace c4
}  // namespace qmri
#endif
