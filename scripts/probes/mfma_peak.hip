// probe: what does v_mfma_f32_32x32x16_f16 sustain on this part with NOTHING else in the loop -- operands in registers, four
// independent accumulator chains per wave, no memory, no LDS -- on zero operands and on random operands (the data-dependent
// power draw sets the clock the part holds), at 1 and 2 waves per SIMD?  The conv kernels' MFMA issue rate is read against this.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_peak.hip -o scripts/probes/mfma_peak && scripts/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NOPER>
__global__ __launch_bounds__(256) void mfma_loop(const f16x8 *__restrict__ src, int iters, float *out) {
    // NOPER different A and B fragments per lane, cycled through: consecutive MFMAs see different operands (an MFMA fed the
    // same registers every time toggles nothing on its inputs)
    f16x8 a[NOPER], b[NOPER];
    const int lane = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
    for (int i = 0; i < NOPER; ++i) {
        a[i] = src[(size_t)(2 * i) * 65536 + (lane & 65535)];
        b[i] = src[(size_t)(2 * i + 1) * 65536 + (lane & 65535)];
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NOPER; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[i], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + 1) % NOPER], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) % NOPER], b[i], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 2) % NOPER], b[(i + 3) % NOPER], c3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == 12345.678f) out[0] = s;  // (keeps the chains alive)
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const size_t nfrag = (size_t)16 * 65536;
    std::vector<_Float16> h(nfrag * 8);
    f16x8 *d;
    float *out;
    hipMalloc(&d, nfrag * sizeof(f16x8));
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
    for (int data = 0; data < 5; ++data) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            float u = 0.f;
            if (data == 1) {  // ~N(0, 1): activations / weights of a network
                float s = 0.f;
                for (int k = 0; k < 12; ++k) s += (float)rand() / RAND_MAX;
                u = s - 6.f;
            } else if (data == 2) {  // the LOW parts of a hi + lo split: small magnitudes, random mantissas
                u = ((float)rand() / RAND_MAX - 0.5f) * 9.7e-4f;
            } else {  // data 3: ONE operand after a ReLU (the even fragment planes = every a[]: half of its elements zero), the
                      // other N(0,1); data 4: both operands half zeros
                float s = 0.f;
                for (int k = 0; k < 12; ++k) s += (float)rand() / RAND_MAX;
                u = s - 6.f;
                const bool plane_a = ((i / 8 / 65536) & 1) == 0;
                if ((plane_a || data == 4) && u < 0.f) u = 0.f;
            }
            h[i] = (_Float16)u;
        }
        hipMemcpy(d, h.data(), nfrag * sizeof(f16x8), hipMemcpyHostToDevice);
        for (int waves = 1; waves <= 2; ++waves) {
            const int blocks = cus * waves;  // 256 threads = 4 waves per block: `waves` waves per SIMD
            const int iters = 60000;
            auto fn = mfma_loop<8>;
            hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, 2000, out);  // warm-up
            hipDeviceSynchronize();
            float best = 1e30f, last = 0.f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, d, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&last, e0, e1);
                if (last < best) best = last;
            }
            const double mfmas = (double)blocks * 4 * iters * 8 * 4;
            const double tf = mfmas * 32 * 32 * 16 * 2 / (best * 1e-3) / 1e12, tf_last = mfmas * 32 * 32 * 16 * 2 / (last * 1e-3) / 1e12;
            // 8 passes of 4 cycles per 32x32x16 f16 MFMA and SIMD: clock = MFMAs per SIMD * 32 / time (if the pipe never idles)
            const double ghz = (double)iters * 8 * 4 * waves * 32 / (best * 1e-3) / 1e9;
            printf("%-34s %d wave(s)/SIMD: best %8.2f ms = %7.1f TFLOP/s (4th run %7.1f); pipe never idle <=> %.2f GHz\n",
                   data == 0 ? "zeros" : data == 1 ? "N(0,1) operands" : data == 2 ? "low parts of a split (|x|<5e-4)" : data == 3 ? "A = relu(N(0,1)), B = N(0,1)" : "A, B = relu(N(0,1))", waves, best, tf,
                   tf_last, ghz);
        }
    }
    return 0;
}
