// probe: how fast does ONE CU get global stores issued on gfx950?  conv_c4_kernel's epilogue moves 256 KB per work item and CU at
// ~18 B per clock: is that the part's rate for 16-byte-per-lane stores, or a property of the epilogue's write pattern
// (a wave-store = 16 segments of 64 B, 256-512 B apart)?  One block of W waves per CU (256 blocks), every wave stores `iters` x 1 KB
// from registers with nothing else to do; patterns:
//   0 contiguous        lane i -> base + 16 i                               (1 KB contiguous per wave-store)
//   1 c4 epilogue       lane = (pixel = lane >> 2, piece = lane & 3): pixel * stride + piece * 16      (16 x 64 B, `stride` apart)
//   2 full records      lane = (pixel = lane >> 3, piece = lane & 7): pixel * stride + piece * 16      (8 x 128 B)
//   3 full records from split lanes: lanes 0-31 the hi halves (pixel = lane >> 2, piece = lane & 3), lanes 32-63 the lo halves of the SAME 8
//     pixels (what a v_permlane32_swap of the epilogue's hi / lo pieces would give: 8 x 128 B per wave-store, each record from two lane groups)
// nt = 1: __builtin_nontemporal_store.  Output: bytes per clock and CU (s_memtime), GB/s per CU and of the chip.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/store_issue.hip -o scripts/probes/store_issue && scripts/probes/store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool NT>
__global__ void k(unsigned char *out, long long per_block, int iters, int stride, unsigned long long *cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned char *base = out + (long long)blockIdx.x * per_block;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    long long lane_off;
    long long step;  // bytes a wave-store advances
    if (PATTERN == 0) {
        lane_off = lane * 16;
        step = 1024;
    } else if (PATTERN == 1) {
        lane_off = (long long)(lane >> 2) * stride + (lane & 3) * 16;
        step = 16LL * stride;
    } else if (PATTERN == 2) {
        lane_off = (long long)(lane >> 3) * stride + (lane & 7) * 16;
        step = 8LL * stride;
    } else {
        lane_off = (long long)((lane & 31) >> 2) * stride + (lane >> 5) * 64 + (lane & 3) * 16;
        step = 8LL * stride;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // waves interleave their store streams: wave w takes wave-stores w, w + nw, ...
    for (int i = 0; i < iters; ++i) {
        long long off = ((long long)i * nw + wave) * step + lane_off;
        if (PATTERN == 1 && (i & 1)) off += 64;  // (hi plane, then lo plane of the same records)
        off &= (per_block >> 1) - 16;  // (wrap by a power-of-two mask: a 64-bit modulo here was ~190 clocks of ALU per store -- the first version of this probe measured that)
        u32x4 *dst = reinterpret_cast<u32x4 *>(base + off);
        if (NT)
            __builtin_nontemporal_store(v, dst);
        else
            *dst = v;
        v.x += 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// bursts between compute phases, like a persistent convolution kernel: `items` times { spin for `spin` shader clocks without touching
// memory; every wave issues `per_burst` wave-stores and does NOT wait for them }.  What the burst costs = (time - items * spin) / items.
template <int PATTERN>
__global__ void kb(unsigned char *out, long long per_block, int items, int per_burst, int stride, int spin, int stagger) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned char *base = out + (long long)blockIdx.x * per_block;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
    long long lane_off, step;
    if (PATTERN == 0) {
        lane_off = lane * 16;
        step = 1024;
    } else if (PATTERN == 1) {
        lane_off = (long long)(lane >> 2) * stride + (lane & 3) * 16;
        step = 16LL * stride;
    } else if (PATTERN == 2) {
        lane_off = (long long)(lane >> 3) * stride + (lane & 7) * 16;
        step = 8LL * stride;
    } else {
        lane_off = (long long)((lane & 31) >> 2) * stride + (lane >> 5) * 64 + (lane & 3) * 16;
        step = 8LL * stride;
    }
    long long n = 0;
    if (stagger > 0) {  // (out of phase: CU groups start a fraction of a compute phase apart; blockIdx / 8 so that every XCD has all phases)
        const unsigned long long late = (unsigned long long)((blockIdx.x >> 3) % stagger) * (unsigned long long)spin / stagger;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < late) __builtin_amdgcn_s_sleep(2);
    }
    for (int it = 0; it < items; ++it) {
        if (spin > 0) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(2);
        }
        for (int i = 0; i < per_burst; ++i, ++n) {
            long long off = (n * nw + wave) * step + lane_off;
            if (PATTERN == 1 && (i & 1)) off += 64;
            off &= (per_block >> 1) - 16;
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(base + off));
            v.x += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// the same burst with narrower stores (8 and 4 bytes per lane), full-record pattern: is the CU's limit bytes or instructions?
template <int WORDS>
__global__ void kw(unsigned char *out, long long per_block, int items, int per_burst, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    unsigned char *base = out + (long long)blockIdx.x * per_block;
    unsigned v = lane;
    long long n = 0;
    for (int it = 0; it < items; ++it) {
        if (spin > 0) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(2);
        }
        for (int i = 0; i < per_burst; ++i, ++n) {
            long long off = ((n * nw + wave) * 64 + lane) * (4LL * WORDS);   // contiguous: 64 lanes x WORDS dwords
            off &= (per_block >> 1) - 4LL * WORDS;
            if (WORDS == 4) {
                u32x4 t = {v, v, v, v};
                __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(base + off));
            } else if (WORDS == 2) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                u32x2 t = {v, v};
                __builtin_nontemporal_store(t, reinterpret_cast<u32x2 *>(base + off));
            } else {
                __builtin_nontemporal_store(v, reinterpret_cast<unsigned *>(base + off));
            }
            v += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int WORDS>
static double run_width(unsigned char *buf, long long per_block, int waves, int items, int per_burst, int spin) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kw<WORDS>), dim3(256), dim3(waves * 64), 0, 0, buf, per_block, items, per_burst, spin);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kw<WORDS>), dim3(256), dim3(waves * 64), 0, 0, buf, per_block, items, per_burst, spin);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

static int g_blocks = 256;      // CUs that take part
static int g_stagger = 0;       // > 0: block b starts (b % g_stagger) * spin / g_stagger clocks late (CUs out of phase)
template <int PATTERN>
static double run_bursts(unsigned char *buf, long long per_block, int items, int per_burst, int stride, int spin) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((kb<PATTERN>), dim3(g_blocks), dim3(256), 0, 0, buf, per_block, items, per_burst, stride, spin, g_stagger);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kb<PATTERN>), dim3(g_blocks), dim3(256), 0, 0, buf, per_block, items, per_burst, stride, spin, g_stagger);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}

template <int PATTERN, bool NT>
static void run(const char *name, unsigned char *buf, long long per_block, int waves, int iters, int stride, unsigned long long *dcyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256;
    hipLaunchKernelGGL((k<PATTERN, NT>), dim3(blocks), dim3(waves * 64), 0, 0, buf, per_block, iters, stride, dcyc);  // warm
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<PATTERN, NT>), dim3(blocks), dim3(waves * 64), 0, 0, buf, per_block, iters, stride, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += (double)v;
    mean /= blocks;
    const double bytes = (double)waves * iters * 1024.0;
    // s_memtime ticks at 100 MHz on this part (10 ns): convert with the kernel's wall time instead of assuming a clock
    printf("%-34s waves %d  stride %4d  %6.1f KB per CU in %7.1f us -> %6.1f GB/s per CU, %5.2f TB/s chip\n", name, waves, stride, bytes / 1024.0,
           ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    const long long per_block = 64LL << 20;  // 64 MB per block: 16 GB for 256 blocks
    unsigned char *buf;
    if (hipMalloc(&buf, per_block * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    unsigned long long *dcyc;
    hipMalloc(&dcyc, 256 * 8);
    const int iters = 2048;  // per wave: 2 MB; 4 waves: 8 MB per CU (an epilogue moves 0.25 MB per item)
    for (int waves : {1, 4, 8}) {
        run<0, true>("contiguous, nt", buf, per_block, waves, iters, 0, dcyc);
        run<0, false>("contiguous, default policy", buf, per_block, waves, iters, 0, dcyc);
        for (int stride : {128, 256, 512}) {
            run<1, true>("c4 epilogue (16 x 64 B), nt", buf, per_block, waves, iters, stride, dcyc);
            run<2, true>("full records (8 x 128 B), nt", buf, per_block, waves, iters, stride, dcyc);
            run<3, true>("full records, split lanes, nt", buf, per_block, waves, iters, stride, dcyc);
        }
    }
    // short bursts like one epilogue: 64 wave-stores per wave (256 KB per CU with 4 waves)
    run<1, true>("c4 epilogue, burst of 64 per wave", buf, per_block, 4, 64, 256, dcyc);
    run<0, true>("contiguous, burst of 64 per wave", buf, per_block, 4, 64, 0, dcyc);
    // a persistent kernel's epilogues: 40 items of { compute phase, 64 wave-stores per wave = 256 KB per CU, not waited for }
    printf("\nbursts of 256 KB per CU (4 waves x 64 wave-stores) between compute phases of `spin` clocks, 40 items; us per item beyond the spin-only run:\n");
    for (int spin : {20000, 60000, 150000}) {
        const double t_spin = run_bursts<0>(buf, per_block, 40, 0, 256, spin);
        const double t_c = run_bursts<0>(buf, per_block, 40, 64, 256, spin);
        const double t_h = run_bursts<1>(buf, per_block, 40, 64, 256, spin);
        const double t_f = run_bursts<2>(buf, per_block, 40, 64, 256, spin);
        const double t_h5 = run_bursts<1>(buf, per_block, 40, 64, 512, spin);
        const double t_f5 = run_bursts<2>(buf, per_block, 40, 64, 512, spin);
        const double t_s = run_bursts<3>(buf, per_block, 40, 64, 256, spin);
        printf("spin %6d: full records from split lanes stride 256: +%5.2f us\n", spin, (t_s - t_spin) / 40);
        printf("spin %6d clocks (%6.1f us per item): contiguous +%5.2f us | 16 x 64 B (c4 epilogue) stride 256: +%5.2f, 512: +%5.2f | 8 x 128 B (full records) stride 256: +%5.2f, 512: +%5.2f\n",
               spin, t_spin / 40, (t_c - t_spin) / 40, (t_h - t_spin) / 40, (t_h5 - t_spin) / 40, (t_f - t_spin) / 40, (t_f5 - t_spin) / 40);
    }
    // is the burst's cost the CU's own, or the chip's (everyone bursting at once)?  fewer CUs; and all CUs, out of phase
    printf("\nthe same burst (c4 epilogue pattern, stride 256, spin 60000) with fewer CUs taking part, and with the CUs out of phase:\n");
    for (int nb : {256, 128, 64, 32, 8}) {
        g_blocks = nb;
        const double t_spin = run_bursts<0>(buf, per_block, 40, 0, 256, 60000);
        const double t_h = run_bursts<1>(buf, per_block, 40, 64, 256, 60000);
        const double t_f = run_bursts<2>(buf, per_block, 40, 64, 256, 60000);
        printf("%3d CUs: half-records +%5.2f us per burst, full records +%5.2f\n", nb, (t_h - t_spin) / 40, (t_f - t_spin) / 40);
    }
    g_blocks = 256;
    for (int st : {2, 4, 8}) {
        g_stagger = st;
        const double t_spin = run_bursts<0>(buf, per_block, 40, 0, 256, 60000);
        const double t_h = run_bursts<1>(buf, per_block, 40, 64, 256, 60000);
        const double t_f = run_bursts<2>(buf, per_block, 40, 64, 256, 60000);
        printf("256 CUs in %d phases: half-records +%5.2f us per burst, full records +%5.2f\n", st, (t_h - t_spin) / 40, (t_f - t_spin) / 40);
    }
    // bytes or instructions?  64 wave-stores per wave and burst, contiguous, 16 / 8 / 4 bytes per lane; and 8 waves (two per SIMD) of 32
    printf("\n64 wave-stores per wave and burst (4 waves, contiguous, spin 60000): us per burst beyond the spin-only run\n");
    {
        const double t0 = run_width<4>(buf, per_block, 4, 40, 0, 60000);
        printf("16 B per lane (256 KB per CU): +%5.2f us | 8 B per lane (128 KB): +%5.2f | 4 B per lane (64 KB): +%5.2f\n",
               (run_width<4>(buf, per_block, 4, 40, 64, 60000) - t0) / 40, (run_width<2>(buf, per_block, 4, 40, 64, 60000) - t0) / 40,
               (run_width<1>(buf, per_block, 4, 40, 64, 60000) - t0) / 40);
        const double t8 = run_width<4>(buf, per_block, 8, 40, 0, 60000);
        printf("8 waves x 32 wave-stores of 16 B per lane (256 KB per CU): +%5.2f us; 1 wave x 256: +%5.2f; 2 waves x 128: +%5.2f\n",
               (run_width<4>(buf, per_block, 8, 40, 32, 60000) - t8) / 40,
               (run_width<4>(buf, per_block, 1, 40, 256, 60000) - run_width<4>(buf, per_block, 1, 40, 0, 60000)) / 40,
               (run_width<4>(buf, per_block, 2, 40, 128, 60000) - run_width<4>(buf, per_block, 2, 40, 0, 60000)) / 40);
    }
    return 0;
}
