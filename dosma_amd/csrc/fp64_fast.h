// fp64_fast.h -- fp64 division / logarithm building blocks for the gfx950 kernels (dess.hip, monoexp_lm.hip).
//
// On CDNA an IEEE fp64 division is 13 VALU instructions (v_div_scale x2, v_rcp_f64, 7 FMAs, v_div_fmas, v_div_fixup)
// and OCML's double-double log 78; the per-voxel kernels here were issue-bound on them.  The fast paths below are
// 1-2 ulp and fall back to the IEEE / OCML forms outside a safe magnitude range; the cold blocks carry an asm marker
// because the compiler otherwise if-converts them (executes the slow form on EVERY call and selects).
#ifndef QMRI_FP64_FAST_H
#define QMRI_FP64_FAST_H
#include <hip/hip_runtime.h>

#include <cmath>

namespace qmri {

#ifndef QMRI_COLD_PATH
#define QMRI_COLD_PATH() asm volatile("; cold path" ::: "memory")
#endif

__device__ __forceinline__ double rcp_nr(double b) {  // b finite, normal, far from the range limits
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    return fma(r, e, r);
}

__device__ __forceinline__ double div_fast(double a, double b) {
    const double ab = fabs(b), aa = fabs(a);
    if (ab > 1e-140 && ab < 1e140 && aa < 1e140) {
        const double r = rcp_nr(b);
        const double q = a * r;
        return fma(fma(-b, q, a), r, q);  // one residual step: correctly rounded when r = RN(1/b), else <= 1 ulp
    }
    QMRI_COLD_PATH();
    return a / b;
}

// x / p10 with the host's correctly rounded reciprocal ip10 = RN(1 / p10): q = RN(x ip10), r = x - p10 q (exact in
// an fma), q + r ip10 rounds to RN(x / p10) (Markstein) -- numpy.around's division must be reproduced exactly, an ulp
// off would make most rounded values compare unequal
__device__ __forceinline__ double div_p10(double x, double p10, double ip10) {
    if (fabs(x) < 1e290) {
        const double q = x * ip10;
        return fma(fma(-p10, q, x), ip10, q);
    }
    QMRI_COLD_PATH();
    return x / p10;
}

// a * b + K, a * K and a * K + c with the 64-bit constant K as a SCALAR operand that is made where it is used: two s_mov_b32
// (they issue on the scalar unit) into s[100:101] and ONE vector instruction.  A 64-bit constant cannot be an immediate of
// v_fma_f64 / v_mul_f64; left to itself hipcc hoists such constants out of the caller's loop and keeps them there -- in
// scalar registers that the loop then spills, or in vector registers with a v_mov_b64 + v_fmac_f64 pair per step.
//
// s[100:101] is the top pair of the 102 addressable SGPRs of gfx9: hipcc reserves s96..s101 for itself ("clobber list
// contains reserved registers", silenced below) and never allocates them, and naming the pair as a clobber makes it
// count the kernel's SGPR block up to s101 (.amdhsa_next_free_sgpr 102), so the registers exist in every wave that
// runs the asm.  tests/test_abi.py::test_reserved_sgpr_pair_is_only_touched_by_the_constant_macros checks both facts
// on the generated code of every kernel that contains the pair.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <unsigned long long K>
__device__ __forceinline__ double fma_sk_bits(double a, double b) {
    double d;
    asm("s_mov_b32 s100, %3\n\ts_mov_b32 s101, %4\n\tv_fma_f64 %0, %1, %2, s[100:101]"
        : "=v"(d) : "v"(a), "v"(b), "n"((unsigned)(K & 0xffffffffull)), "n"((unsigned)(K >> 32)) : "s100", "s101");
    return d;
}
template <unsigned long long K>
__device__ __forceinline__ double mul_sk_bits(double a) {
    double d;
    asm("s_mov_b32 s100, %2\n\ts_mov_b32 s101, %3\n\tv_mul_f64 %0, %1, s[100:101]"
        : "=v"(d) : "v"(a), "n"((unsigned)(K & 0xffffffffull)), "n"((unsigned)(K >> 32)) : "s100", "s101");
    return d;
}
template <unsigned long long K>
__device__ __forceinline__ double fma_ks_bits(double a, double c) {
    double d;
    asm("s_mov_b32 s100, %3\n\ts_mov_b32 s101, %4\n\tv_fma_f64 %0, %1, s[100:101], %2"
        : "=v"(d) : "v"(a), "v"(c), "n"((unsigned)(K & 0xffffffffull)), "n"((unsigned)(K >> 32)) : "s100", "s101");
    return d;
}
#pragma clang diagnostic pop
#define QMRI_K64(k) __builtin_bit_cast(unsigned long long, static_cast<double>(k))
#define QMRI_FMA_SK(a, b, k) ::qmri::fma_sk_bits<QMRI_K64(k)>((a), (b))   /* a * b + k */
#define QMRI_MUL_SK(a, k) ::qmri::mul_sk_bits<QMRI_K64(k)>((a))          /* a * k */
#define QMRI_FMA_KS(a, k, c) ::qmri::fma_ks_bits<QMRI_K64(k)>((a), (c))  /* a * k + c */

// a * b + k with the constant k as a scalar operand the COMPILER places: inside a loop body that uses the same constants
// several times (one exp per sample) it makes them once per round, which is cheaper than the in-place form above
__device__ __forceinline__ double fma_sk(double a, double b, double k) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
}

// exp(x) exactly as the device library evaluates it -- k = rint(x log2 e), r = x - k ln 2 (two-part constant), the same
// degree-11 polynomial in the same order, ldexp, the same two range tests: the results are bit-identical -- with the
// polynomial's constants as scalar operands.  Inside a register-limited loop hipcc keeps the library form's nine 64-bit
// coefficients in 18 VGPRs and spends a v_mov_b64 + v_fmac_f64 per Horner step (the accumulating form needs the constant
// in its destination): 9 vector instructions and 18 registers more per call than this form.
__device__ __forceinline__ double exp_sk(double x) {
    const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(k, -0x1.62e42fefa39efp-1, x);
    r = fma(k, -0x1.abc9e3b39803fp-56, r);
    double p = fma(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma_sk(r, p, 0x1.71dee623fde64p-19);
    p = fma_sk(r, p, 0x1.a01997c89e6b0p-16);
    p = fma_sk(r, p, 0x1.a01a014761f6ep-13);
    p = fma_sk(r, p, 0x1.6c16c1852b7b0p-10);
    p = fma_sk(r, p, 0x1.1111111122322p-7);
    p = fma_sk(r, p, 0x1.55555555502a1p-5);
    p = fma_sk(r, p, 0x1.5555555555511p-3);
    p = fma_sk(r, p, 0x1.000000000000bp-1);
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
    double z = __builtin_amdgcn_ldexp(p, static_cast<int>(k));
    z = x > 1024.0 ? __builtin_huge_val() : z;
    z = x < -1075.0 ? 0.0 : z;
    return z;
}

// natural log of a positive, finite, normal x, ~1 ulp: x = 2^e m, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
// log m = 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ...), z = s^2 <= 0.0295 (13 terms: remainder < 1e-21)
__device__ __forceinline__ double log_fast(double x) {
    if (x > 1e-300 && x < 1e300) {
        double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
        int e = __builtin_amdgcn_frexp_exp(x);
        if (m < 0.70710678118654752) {
            m = m + m;
            e -= 1;
        }
        const double f = m - 1.0;
        const double s = f * rcp_nr(2.0 + f);
        const double z = s * s;
        double p = 1.0 / 25.0;
        p = fma(p, z, 1.0 / 23.0);
        p = fma(p, z, 1.0 / 21.0);
        p = fma(p, z, 1.0 / 19.0);
        p = fma(p, z, 1.0 / 17.0);
        p = fma(p, z, 1.0 / 15.0);
        p = fma(p, z, 1.0 / 13.0);
        p = fma(p, z, 1.0 / 11.0);
        p = fma(p, z, 1.0 / 9.0);
        p = fma(p, z, 1.0 / 7.0);
        p = fma(p, z, 1.0 / 5.0);
        p = fma(p, z, 1.0 / 3.0);
        const double ed = (double)e;
        const double s2 = s + s;
        // e ln2_hi + (2 s + (2 s z p + e ln2_lo)); ln2_hi has 11 trailing zero bits: e ln2_hi is exact
        const double lo = fma(s2 * z, p, ed * 1.9082149292705877e-10);
        return fma(ed, 0.69314718036912382, s2 + lo);
    }
    QMRI_COLD_PATH();
    return log(x);
}

// log_fast with every constant as a scalar operand made in place (see exp_sk: nothing stays pinned in registers across
// the caller's loop) and without the device library's log behind it: every positive finite argument, denormals included,
// goes through the frexp form; x < 0 and NaN give NaN, 0 gives -inf, +inf gives +inf like log().
__device__ __forceinline__ double log_sk(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752) {
        m = m + m;
        e -= 1;
    }
    const double f = m - 1.0;
    const double s = f * rcp_nr(2.0 + f);
    const double z = s * s;
    double p = QMRI_FMA_KS(z, 1.0 / 25.0, 1.0 / 23.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 21.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 19.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 17.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 15.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 13.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 11.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 9.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 7.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 5.0);
    p = QMRI_FMA_SK(p, z, 1.0 / 3.0);
    const double ed = (double)e;
    const double s2 = s + s;
    const double lo = fma(s2 * z, p, QMRI_MUL_SK(ed, 1.9082149292705877e-10));
    double r = QMRI_FMA_KS(ed, 0.69314718036912382, s2 + lo);
    // specials: class mask 0x200 = +inf, 0x060 = +-0, 0x01f = NaNs, -inf, negative normals / denormals
    r = __builtin_amdgcn_class(x, 0x200) ? x : r;
    r = __builtin_amdgcn_class(x, 0x060) ? -__builtin_huge_val() : r;
    r = __builtin_amdgcn_class(x, 0x01f) ? __builtin_nan("") : r;
    return r;
}

}  // namespace qmri
#endif
