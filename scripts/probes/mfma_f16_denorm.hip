// probe: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs (needed by the fp16 hi/lo split)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float a_val, float b_val, float *out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // k index 0 only: A[row][0] = a_val for every row, B[0][col] = b_val
    if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float av[] = {1.0f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 3.0e-5f};
    const float bv[] = {1.0f, 1.0f, 1024.0f, 16384.0f, 3.0e-5f};
    for (int i = 0; i < 5; ++i) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, av[i], bv[i], d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g (fp16 %g) b=%g -> mfma %g expected %g\n", av[i], (float)(_Float16)av[i], bv[i], h,
               (float)(_Float16)av[i] * (float)(_Float16)bv[i]);
    }
    return 0;
}
