// unet_s3.hip -- the parity-mode ("fp16x3") 3x3 convolution of the 2D U-Net for gfx950 (MI355X).
//
// Same layer as conv_igemm_kernel (unet_kernels.hip): Conv2D(3x3, SAME) + bias + ReLU (+ the BatchNormalization affine
// that follows the 2nd ReLU of a block) of /root/reference/dosma/models/oaiunet2d.py:213-226, 266-279, in the
// precision mode whose logits meet north_star's 1e-3 bar: every operand is a 16-bit hi + lo pair and a product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 (fp32 accumulate).  scripts/unet_precision_sim.py shows why it has
// to be three MFMAs (any two-MFMA split is 7e-3 .. 3e-2 off on the logits) and why the parts are fp16, not bf16
// (2^-22 vs 2^-17 per product: 1.4e-5 instead of 6.7e-4 on a network with realistic BatchNorm statistics).
//
// What is different from the general kernel (which reached 33-37 % MFMA issue in this mode, round 1):
//   * ACTIVATIONS LIVE IN HBM ALREADY SPLIT: per pixel and 32-channel chunk 32 fp16 hi parts then 32 fp16 lo parts
//     (128 B, the bytes of the fp32 values they replace).  The producing epilogue splits once; consumers never convert,
//     so the halo goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass,
//     no VALU work in the main loop at all.
//   * PERSISTENT blocks of 8 waves walk (channel block, pixel tile) work items.  The K loop is a stream of
//     steps (32-channel chunk x tap) that does not stop at a tile boundary: weights run through a 4-slot LDS ring
//     requested three steps ahead, the halo of the next chunk -- possibly the next tile's first -- is requested one
//     piece per step into the other halo buffer, all by DMA with COUNTED s_waitcnt vmcnt (never 0 in the loop) and ONE
//     raw s_barrier per step; the MFMA operands of step s+1 are read from LDS before the barrier that ends step s.
//   * a tile is 8 image rows x 32 pixels (one MFMA row-tile = 32 consecutive pixels of one image row: with the
//     XOR-swizzled 64-byte pixel rows every ds_read_b128 lane group is conflict-free at any tap shift), or, for images
//     narrower than 64 pixels, 256 consecutive positions of the FLATTENED zero-framed image stack (pitch W + 2, one
//     zero row between images): a 3x3 tap is then a constant shift of the flat index, row-tiles are still 32
//     consecutive LDS pixels, and only the 2 frame columns per row are wasted (12 x 12 images: 14 % instead of the
//     44 % a 16 x 16 tile wastes).
//   * the epilogue goes through a wave-private 4 KB LDS window (the finished chunk's halo buffer): bias / ReLU /
//     affine, split, [pixel][hi | lo] image, 16-byte global stores of whole 128-byte pixel-chunks; fused 2x2 max-pool
//     (lane-local in the MFMA C layout) and fused 1x1 head + threshold as in the general kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "qmri_internal.h"

namespace qmri {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) __fp16 h16x2;  // what v_cvt_pkrtz_f16_f32 returns
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ uint4 g_zero16_s3;  // source of halo pieces outside the image (zero padding)

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kMTile = 256;        // output positions per tile: 8 MFMA row-tiles of 32
constexpr int kPitch2D = 34;       // halo row pitch of the 8 x 32 tile
constexpr int kHalo2D = 10 * kPitch2D;
constexpr int kRing = 4;           // weight ring slots

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform base + lane * 16 (M0), global source per lane.
// Issued through inline asm so that hipcc neither counts it (it would drain vmcnt(0) before every ds_read it cannot
// prove disjoint) nor waits for it: every wait in this file is a hand-counted s_waitcnt vmcnt(N).
// (M0 is written and read inside ONE statement and not restored: nothing else in these kernels uses M0 -- gfx9+ LDS
//  instructions do not -- and tests/test_abi.py checks the generated code for any other M0 reader.)
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(lds_dst_wave_base)
                 : "memory");
}

// 16-byte feature-map store with the non-temporal hint (a layer's output is not read again by this kernel).  Measured
// neutral on the whole network (35.4 ms either way): kept as the statement of intent, not as an optimisation
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store16(void *dst, const uint4 &v) {
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(dst));
}

#ifdef QMRI_S3_EXPERIMENTS
__device__ unsigned long long s3_tstat[8];  // QMRI_S3_DBG & 1024: cycles in [0] main loop [1] epilogue [2] tile switch [3] work items [4] affine part
#endif
template <int BN>
struct S3Cfg {
    static constexpr int WN = BN >= 64 ? 2 : 1;          // waves along the channel axis
    static constexpr int WM = kWaves / WN;               // waves along the pixel axis
    static constexpr int RT = 8 / WM;                    // 32-pixel row-tiles per wave
    static constexpr int CT = BN / WN / 32;              // 32-channel column tiles per wave
    // Taps per ring slot = taps between two barriers.  At BN = 32 a tap is only 6 MFMAs per wave -- less than a barrier
    // and a round of requests cost -- so a slot holds a whole tap ROW (3 taps, 12 KB) and the barrier comes once per row.
    static constexpr int TPS = BN == 32 ? 3 : 1;
    static constexpr int TAP_BYTES = BN * 128;           // one tap of one 32-channel chunk: [plane][BN][64 B]
    static constexpr int SLOT_BYTES = TPS * TAP_BYTES;   // one ring slot
    static constexpr int W_INSTR = SLOT_BYTES / 1024;    // DMA wave-instructions per slot: 16 / 8 / 12
    static constexpr int W_PER_WAVE = (W_INSTR + kWaves - 1) / kWaves;
    static constexpr int H_PER_TAP = TPS == 3 ? 2 : 1;   // halo pieces (or repeats) requested per tap
    // DMA instructions per wave between two barriers (the waits are counted): weights of one slot + the halo pieces
    static constexpr int DPS = W_PER_WAVE + TPS * H_PER_TAP;
};

__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)(lds_void *)p; }

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

}  // namespace

#ifdef QMRI_S3_EXPERIMENTS  // timing experiments of DESIGN.md section 6 (QMRI_S3_DBG = 1 no vmcnt wait | 2 no barrier | 4 no requests | 8 no LDS reads | 16 no epilogue global stores | 32 no epilogue at all | 64 no MFMAs | 128 stores into a 4 MB window | 256 one store of four)
#define S3_DBG(bit) (A.dbg & (bit))
#define S3_NOW() __builtin_amdgcn_s_memtime()
#else
#define S3_DBG(bit) 0
#endif

// DECONV: Conv2DTranspose(3x3, strides 2, SAME) (oaiunet2d.py:259-261) on the same machinery.  The tile is a tile of the
// INPUT grid; the 9 taps (packed in phase order by pack_deconv_fused: 4 + 2 + 2 + 1) read the input at (dy, dx) in
// {0, -1}^2 and accumulate into the accumulator set of their output phase (py, px); the epilogue writes the four phases
// to the output pixels (2 y + py, 2 x + px).  The taps are unrolled (the accumulator set must be a compile-time index).
// ONE: the plain-bf16 mode on the same machinery (round 3).  The bf16 NHWC layout already IS the record layout for 64-channel
// chunks -- a pixel's 64 channels are 128 contiguous bytes -- so the halo DMA, the LDS image, the swizzles and the weight ring
// are used unchanged with "plane 0 / plane 1" = channels 0-31 / 32-63 of the chunk: a k-step is ah x bh + al x bl (two MFMAs
// for 64 channels instead of three for 32).  The launcher passes x in 4-byte units (ldx = channels / 2, Cin = channels / 2);
// y / pool_y stay in channels of 2 bytes.  No fused classifier, no saturation flag (bf16 has the fp32 exponent range).
template <int BN, bool FLAT, bool DECONV, bool ONE = false>
__global__ __launch_bounds__(kThreads, 2) void conv_s3_kernel(const ConvS3Args A) {
    using C = S3Cfg<BN>;
    constexpr int RT = C::RT, CT = C::CT;
    constexpr int NPH = DECONV ? 4 : 1;
    static_assert(!DECONV || C::TPS == 1 || BN == 32, "");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NJ = A.nj;                       // DMA instructions (8 pixels x 128 B each) per halo buffer
    const int hbuf_bytes = NJ * 1024;
    unsigned char *halo = smem;                                  // [2 buffers][NJ * 8 pixels][hi 64 B | lo 64 B, swizzled]
    unsigned char *ring = smem + 2 * hbuf_bytes;                 // [kRing][2 planes][BN][64 B]
    int *outpix = reinterpret_cast<int *>(ring + kRing * C::SLOT_BYTES);  // [256] output pixel of a tile position, or -1
    float *prm = reinterpret_cast<float *>(outpix + kMTile);     // bias | scale | shift, [BN] each
    float *hw = prm + 3 * BN;                                    // head weights [32][4] + bias [4] (BN = 32 only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef QMRI_S3_EXPERIMENTS
    if (S3_DBG(512)) {  // experiment: eight phase groups of CUs, started (dbg >> 16) x 1024 cycles apart
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)((blockIdx.x >> 3) & 7) * (unsigned)(A.dbg >> 16) * 1024ull;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
#endif
    const int wn = wave % C::WN, wm = wave / C::WN;
    const int P = FLAT ? A.P : kPitch2D;       // LDS / flat pitch of one image row
    const int hpix = FLAT ? kMTile + 2 * P + 2 : kHalo2D;

    const int steps = A.steps;                 // chunks * 9 per work item
    const int wsteps = steps / C::TPS;         // ring slots per work item
    const int ntiles = A.ntiles;

    // ---- per-lane constants of the halo DMA: this wave's slot i is instruction j = wave + 8 i of the NJ that make up a
    // halo buffer.  One instruction moves 8 WHOLE pixel-chunks: 8 consecutive lanes fetch the 128 contiguous bytes
    // (64 B of hi parts, 64 B of lo parts) of one pixel -- full cache lines on the source side (half lines, fetched by
    // two different instructions, cost the texture-address path twice).  LDS image: pixel p at p * 128; its eight
    // 16-byte pieces (plane P in {hi, lo}, K piece q) sit at position ((P ^ (p >> 1 & 1)) * 4 + (q ^ (p >> 2 & 3))) -- the
    // swizzle lives on the SOURCE address and on the ds_read address; with 2 pixels per 256-byte bank row the 16 lanes of
    // a ds_read_b128 group (16 consecutive pixels, one plane, one q) then hit 16 different 16-byte slots.
    constexpr int kSlots = 6;
    int h_hp[kSlots], h_srcb[kSlots];  // halo pixel (or -1: beyond the halo), byte offset inside the pixel-chunk
    unsigned h_dst[kSlots];            // LDS byte offset of the instruction inside a halo buffer
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
        int j = wave + kWaves * i;
        if (j >= NJ) j = wave;  // no such piece: repeat this wave's first one (same bytes to the same place)
        const int hp = j * 8 + (lane >> 3);
        const int p8 = lane & 7;
        const int plane = (p8 >> 2) ^ ((hp >> 1) & 1);
        const int q = (p8 & 3) ^ ((hp >> 2) & 3);
        h_hp[i] = hp < hpix ? hp : -1;
        h_srcb[i] = plane * 64 + q * 16;
        h_dst[i] = (unsigned)(j * 1024);
    }

    // ---- work item -> tile geometry ----
    // 2D:   tile = (image b, rows y0 .. y0+7, columns x0 .. x0+31)
    // FLAT: tile = flat positions f0 .. f0+255
    int t_b = 0, t_y0 = 0, t_x0 = 0, t_f0 = 0, t_nb = 0;
    auto decode_work = [&](int w, int &nb, int &b, int &y0, int &x0, int &f0) {
        nb = w / ntiles;
        const int t = w - nb * ntiles;
        if (FLAT) {
            f0 = A.P + t * kMTile;
            b = y0 = x0 = 0;
        } else {
            const int per_img = A.tiles_y * A.tiles_x;
            b = t / per_img;
            const int r = t - b * per_img;
            const int ty = r / A.tiles_x;
            y0 = ty * 8;
            x0 = (r - ty * A.tiles_x) * 32;
            f0 = 0;
        }
    };
    // source pixel (index into the NHWC pixel grid) of this lane's piece of slot i for a tile, or -1
    auto halo_src_pix = [&](int i, int b, int y0, int x0, int f0) -> int {
        const int hp = h_hp[i];
        if (hp < 0) return -1;
        if (FLAT) return flat_to_pix(f0 - P - 1 + hp, P, A.H, A.W, A.B);
        const int hy = hp / kPitch2D, hx = hp - hy * kPitch2D;
        const int yy = y0 + hy - 1, xx = x0 + hx - 1;
        if ((unsigned)yy >= (unsigned)A.H || (unsigned)xx >= (unsigned)A.W) return -1;
        return (b * A.H + yy) * A.W + xx;
    };
    const unsigned char *xbase = static_cast<const unsigned char *>(A.x);
    const unsigned halo_lds = lds_off(halo), ring_lds = lds_off(ring);

    const unsigned char *zero_line = reinterpret_cast<const unsigned char *>(&g_zero16_s3);
    // per piece: source of chunk 0 for the tile whose chunks are being REQUESTED (the current tile, or the next one) and
    // the byte step from chunk to chunk (0 for pieces that read the zero line)
    const unsigned char *hsrc[kSlots];
    int hstep[kSlots];
    // 2D tiling: the halo geometry does not depend on the tile -- row / column of the piece's pixel inside the halo and its
    // byte offset relative to the tile's first pixel are per-lane constants; a tile only adds its origin and the border test
    int h_yx[kSlots], h_rel[kSlots];
    if (!FLAT) {
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const int hp = h_hp[i] < 0 ? 0 : h_hp[i];
            const int hy = hp / kPitch2D, hx = hp - hy * kPitch2D;
            h_yx[i] = h_hp[i] < 0 ? -1 : (hy | (hx << 8));
            h_rel[i] = (int)((((long long)(hy - 1) * A.W + (hx - 1)) * A.ldx) * 4) + h_srcb[i];
        }
    }
    auto set_halo_sources = [&](int b, int y0, int x0, int f0) {
        if (!FLAT) {
            const unsigned char *origin = xbase + (((long long)(b * A.H + y0) * A.W + x0) * A.ldx + A.xoff) * 4;
#pragma unroll
            for (int i = 0; i < kSlots; ++i) {
                const int yy = y0 + (h_yx[i] & 0xFF) - 1, xx = x0 + ((h_yx[i] >> 8) & 0xFF) - 1;
                const bool ok = h_yx[i] >= 0 && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W;
                hsrc[i] = ok ? origin + h_rel[i] : zero_line;
                hstep[i] = ok ? 128 : 0;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < kSlots; ++i) {
            const int pix = halo_src_pix(i, b, y0, x0, f0);
            hsrc[i] = pix >= 0 ? xbase + ((long long)pix * A.ldx + A.xoff) * 4 + h_srcb[i] : zero_line;
            hstep[i] = pix >= 0 ? 128 : 0;
        }
    };
    int req_work, req_chunk;  // the next halo chunk to request: work item and chunk index
    int req_buf;

    auto issue_halo_piece = [&](int i) {
        dma16(hsrc[i] + req_chunk * hstep[i], halo_lds + (unsigned)(req_buf * hbuf_bytes) + h_dst[i]);
    };
    // weights of (channel block nb, local step s) -> ring slot
    const unsigned char *wbase = static_cast<const unsigned char *>(A.w);
    auto issue_weights = [&](int nb, int s, int slot) {
#pragma unroll
        for (int r = 0; r < C::W_PER_WAVE; ++r) {
            const int jj = (wave * C::W_PER_WAVE + r) % C::W_INSTR;  // BN = 32: waves 4-7 repeat the pieces of waves 0-3
            const unsigned char *g = wbase + ((long long)nb * wsteps + s) * C::SLOT_BYTES + jj * 1024 + lane * 16;
            dma16(g, ring_lds + (unsigned)(slot * C::SLOT_BYTES + jj * 1024));
        }
    };

    // ---- per-lane LDS read addressing ----
    // A fragment (row-tile i, tap shift sh, k-step kk, plane p): pixel hp = abase[i] + sh, piece (kk*2 + lane>>5) ^ ((hp>>2)&3)
    int abase[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int rt = wm * RT + i;
        abase[i] = FLAT ? rt * 32 + (lane & 31) + P + 1 : (rt + 1) * kPitch2D + (lane & 31) + 1;
    }
    // B fragment (column tile j, k-step kk, plane p): row n = (wn*CT + j)*32 + lane&31 of the slot image
    unsigned boff[CT][2];
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int n = (wn * CT + j) * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) boff[j][kk] = (unsigned)(n * 64 + (((kk * 2 + (lane >> 5)) ^ ((n >> 2) & 3)) * 16));
    }
    const int khalf = lane >> 5;

    struct Frags {
        f16x8 ah[RT], al[RT], bh[CT], bl[CT];
    };
    // LDS byte offsets of the A fragments, per distinct tap shift: the halo geometry is tile-invariant (per-lane pixel
    // abase[i] + shift), so the swizzled offsets are computed ONCE per kernel instead of per read (they were ~320 of the
    // ~670 VALU instructions a 32-channel tile cost per wave).  Convolution: 9 shifts (dy, dx); transposed: 4 ({0,-1} x {0,-1}).
    constexpr bool kUnrollChunk = DECONV || BN < 128;  // BN = 128 keeps a tap-row loop (register budget), see the main loop
    constexpr int NSH = !kUnrollChunk ? 1 : (DECONV ? 4 : 9);
    // (k-step 1 is the k-step-0 offset ^ 32: the piece index q = 2 kk + khalf only flips bit 1; the lo plane is ^ 64)
    int aoff[NSH][RT];
#pragma unroll
    for (int t = 0; t < NSH; ++t) {
        const int shift = DECONV ? -((t >> 1) * P) - (t & 1) : (t / 3 - 1) * P + (t % 3 - 1);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int hp = abase[i] + shift;
            aoff[t][i] = hp * 128 + ((hp >> 1) & 1) * 64 + ((khalf ^ ((hp >> 2) & 3)) * 16);
        }
    }
    // shift id of tap t: convolution t itself; transposed convolution (packed phase order, pack_deconv_fused):
    //   taps (dy, dx) = (0,0) (0,-1) (-1,0) (-1,-1) | (0,0) (-1,0) | (0,0) (0,-1) | (0,0)  ->  id = 2 * (dy == -1) + (dx == -1)
#define S3_SHIFT_ID(T) (DECONV ? ((T) == 1 || (T) == 7 ? 1 : (T) == 2 || (T) == 5 ? 2 : (T) == 3 ? 3 : 0) : (T))
    auto load_frags = [&](Frags &f, int buf, int slot, int wtap, const int (&ao)[RT], int kk) {
        const unsigned char *hb = halo + buf * hbuf_bytes;
        const unsigned char *wb = ring + slot * C::SLOT_BYTES + wtap * C::TAP_BYTES;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int o = ao[i] ^ (kk * 32);
            f.ah[i] = *reinterpret_cast<const f16x8 *>(hb + o);
            f.al[i] = *reinterpret_cast<const f16x8 *>(hb + (o ^ 64));
        }
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            f.bh[j] = *reinterpret_cast<const f16x8 *>(wb + boff[j][kk]);
            f.bl[j] = *reinterpret_cast<const f16x8 *>(wb + BN * 64 + boff[j][kk]);
        }
    };

    float amax = 0.f;  // max |v| of everything this lane has stored in the split layout (saturation flag, ConvS3Args::sat)
    f32x16 acc[NPH][RT][CT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ph][i][j][e] = 0.f;
    };
    // Work distribution.  Blocks are placed on XCD (blockIdx % 8) -- observed, used for speed only -- and every XCD has its
    // own L2: with the plain round-robin walk two tiles that share halo rows run on different XCDs at the same time and the
    // 1.33x halo over-fetch goes to the fabric.  With a grid that is a multiple of 8, XCD x walks the contiguous range
    // [x, x + 1) * ceil(nwork / 8) of work items instead, its blocks interleaved inside it: neighbouring tiles (and one
    // channel block's weights) stay in one L2.
    int wstride = gridDim.x, w_end = A.nwork;
    int work = blockIdx.x;
    if ((gridDim.x & 7) == 0 && A.nwork >= 64) {
        const int per_xcd = (A.nwork + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        wstride = gridDim.x >> 3;
        work = xcd * per_xcd + (blockIdx.x >> 3);
        w_end = (xcd + 1) * per_xcd < A.nwork ? (xcd + 1) * per_xcd : A.nwork;
    }
    const int nwork = w_end;  // (everything below tests "is there such a work item" against this block's range)
    if (work >= nwork) return;
    decode_work(work, t_nb, t_b, t_y0, t_x0, t_f0);

    // ---- prologue: halo chunk 0 of the first tile (all 6 pieces), weights of steps 0, 1, 2 ----
    req_work = work;
    req_chunk = 0;
    req_buf = 0;
    set_halo_sources(t_b, t_y0, t_x0, t_f0);
#pragma unroll
    for (int i = 0; i < kSlots; ++i) issue_halo_piece(i);
    // weight requests run 3 steps ahead of the computation: (w_work, w_nb, w_s) is the NEXT step to request
    int w_work = work, w_nb = t_nb, w_s = 0, w_slot = 0;
    auto advance_w = [&]() {
        if (++w_s == wsteps) {
            w_s = 0;
            w_work += wstride;
            w_nb = w_work < nwork ? w_work / ntiles : w_nb;
        }
        w_slot = (w_slot + 1) & (kRing - 1);
    };
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        issue_weights(w_nb, w_s, w_slot);
        advance_w();
    }
    // the request pointer of the halo moves to chunk 1 (or the next tile's chunk 0)
    auto advance_req = [&]() {
        req_buf ^= 1;
        if (++req_chunk == A.chunks) {
            req_chunk = 0;
            req_work += wstride;
        }
    };
    advance_req();
    bool req_tile_ready = req_chunk != 0;  // src_pix belongs to the tile of req_work?
    for (int i = tid; i < 3 * BN; i += kThreads) {  // epilogue parameters of this block's FIRST channel block
        const int c = i % BN, which = i / BN;
        const int n = t_nb * BN + c;
        prm[i] = which == 0 ? (A.bias ? A.bias[n] : 0.f) : which == 1 ? (A.scale ? A.scale[n] : 1.f) : (A.shift ? A.shift[n] : 0.f);
    }
    if (BN == 32 && A.head_w)
        for (int i = tid; i < 32 * 4 + 4; i += kThreads) {
            const int NC = A.head_nc;
            float v = 0.f;
            if (i < 128) {
                const int co = i >> 2, c = i & 3;
                v = c < NC ? A.head_w[co * NC + c] : 0.f;
            } else {
                v = (i - 128) < NC ? A.head_b[i - 128] : 0.f;
            }
            hw[i] = v;
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    zero_acc();
    int chunk = 0, cbuf = 0, slot = 0;  // the chunk being computed, its halo buffer, the ring slot of the current step
    int row = 0;                        // (row-loop variant only) tap row dy = row - 1 being computed
    Frags f0, f1;
    load_frags(f0, cbuf, slot, 0, aoff[0], 0);  // operands of the very first tap (k-step 0)

    // MFMAs of one k-step in two halves (by row-tile), so that the requests and the address arithmetic of a step can be
    // placed BETWEEN them: an in-order wave hides ~6 ALU / DMA instructions behind each 32-cycle MFMA, and the two waves of
    // a SIMD run in lock-step between barriers -- ~100 bookkeeping instructions in one lump idle the matrix pipe for both.
    auto mma_part = [&](const Frags &f, int part, auto phc) {
        constexpr int PH = decltype(phc)::value;
        if (S3_DBG(64)) return;  // (experiments: no MFMAs)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if ((RT == 1 ? 0 : i) != part) continue;
            if constexpr (ONE) {
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[PH][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.bh[j]), __builtin_bit_cast(bf16x8, f.ah[i]), acc[PH][i][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[PH][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.bl[j]), __builtin_bit_cast(bf16x8, f.al[i]), acc[PH][i][j], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[PH][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], acc[PH][i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[PH][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[PH][i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[PH][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[PH][i][j], 0, 0, 0);
        }
    };
    // this wave's first weight piece of the slot image to request next, as a running per-lane pointer
    const int wjj0 = (wave * C::W_PER_WAVE) % C::W_INSTR;
    auto wptr_of = [&](int nb, int st) -> const unsigned char * {
        return wbase + ((long long)nb * wsteps + st) * C::SLOT_BYTES + wjj0 * 1024 + lane * 16;
    };
    const unsigned char *wp = wptr_of(w_work < nwork ? w_nb : t_nb, w_work < nwork ? w_s : 0);

    // pins: the LDS reads may not sink to their first use (hipcc's default, which exposes the LDS latency); ALU, MFMA and the
    // request statements float, so that the scheduler can hide them behind each other
#define S3_PIN() __builtin_amdgcn_sched_barrier(0x07F)
    // one step = one tap: DXI = dx + 1 is a compile-time constant, the tap row is not
    // Instruction order of a step, spelled out for the scheduler (sched_group_barrier pipelines): the 2 (RT + CT) LDS
    // reads of a k-step ride in the gaps of the first MFMAs of the PREVIOUS k-step, two per MFMA pair, so that no
    // s_waitcnt lgkmcnt sits directly in front of an MFMA; everything else (address arithmetic, the request statements)
    // is free to fill the remaining gaps.
#define S3_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define S3_PIPE()                                                                                                 \
    if constexpr (BN == 128) {                                                                                    \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { S3_SGB(0x100, 2); S3_SGB(0x008, 2); }                  \
        S3_SGB(0x008, 4);                                                                                         \
    } else if constexpr (BN == 64) {                                                                              \
        _Pragma("unroll") for (int g_ = 0; g_ < 3; ++g_) { S3_SGB(0x100, 2); S3_SGB(0x008, 1); }                  \
        S3_SGB(0x008, 3);                                                                                         \
    } else {                                                                                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) { S3_SGB(0x100, 2); S3_SGB(0x008, 1); }                  \
        S3_SGB(0x008, 1);                                                                                         \
    }
    // one step = one tap: DXI = dx + 1 is a compile-time constant, the tap row is not
#define S3_STEP(ROW, DXI, AOC, AON, PH, NBUF)                                                                               \
    {                                                                                                             \
        constexpr bool kSlotStart = C::TPS == 1 || (DXI) == 0; /* first tap of a ring slot: request the slot 3 ahead */ \
        constexpr bool kSlotEnd = C::TPS == 1 || (DXI) == 2;   /* last tap of a ring slot: wait + barrier */      \
        constexpr int kWTap = C::TPS == 3 ? (DXI) : 0;                                                            \
        using PhC_ = std::integral_constant<int, (PH)>;                                                           \
        if (!S3_DBG(8)) load_frags(f1, cbuf, slot, kWTap, (AOC), 1);                            \
        mma_part(f0, 0, PhC_{});                                                                                  \
        mma_part(f0, 1, PhC_{});                                                                                  \
        if constexpr (kSlotStart) { /* weights of the slot 3 ahead -> ring slot w_slot */                         \
            const unsigned wdst = ring_lds + (unsigned)(w_slot * C::SLOT_BYTES + wjj0 * 1024);                    \
            if (!S3_DBG(4)) {                                                                                     \
                dma16(wp, wdst);                                                                                  \
                if (C::W_PER_WAVE == 2) dma16(wp + 1024, wdst + 1024);                                            \
            }                                                                                                     \
            wp += C::SLOT_BYTES;                                                                                  \
            ++w_s;                                                                                                \
            w_slot = (w_slot + 1) & (kRing - 1);                                                                  \
        }                                                                                                         \
        /* halo pieces of the NEXT chunk.  One tap per slot: piece row * 3 + DXI in tap rows 0, 1.  A tap row per slot: all */ \
        /* six pieces in tap row 0 (two per tap), so that they have landed -- the waits come once per row -- before the last */ \
        /* tap of row 2 reads the next chunk's first operands.  Tap rows without pieces issue nothing: the counted wait at */ \
        /* the end of the interval uses the matching count (requesting identical bytes again to keep ONE count cost 29 % of */ \
        /* a 32-channel layer in DMA issue slots). */                                                               \
        const bool due_ = C::TPS == 3 ? (ROW) == 0 : (ROW) < 2;                                                   \
        if (due_ && !S3_DBG(4)) {                                                                                 \
            _Pragma("unroll") for (int h_ = 0; h_ < C::H_PER_TAP; ++h_) {                                         \
                constexpr int kP0 = C::TPS == 3 ? 2 * (DXI) : (DXI);                                              \
                const int i0_ = kP0 + h_, i1_ = C::TPS == 3 ? i0_ : 3 + (DXI);                                    \
                const unsigned char *hs_ = (ROW) == 0 ? hsrc[i0_] : hsrc[i1_];                                    \
                const int st_ = (ROW) == 0 ? hstep[i0_] : hstep[i1_];                                             \
                const unsigned hd_ = (ROW) == 0 ? h_dst[i0_] : h_dst[i1_];                                        \
                dma16(hs_ + req_chunk * st_, (unsigned)__builtin_amdgcn_readfirstlane(                           \
                                                 (int)(halo_lds + (unsigned)(req_buf * hbuf_bytes) + hd_)));      \
            }                                                                                                     \
        }                                                                                                         \
        const int n_slot_ = kSlotEnd ? (slot + 1) & (kRing - 1) : slot;                                           \
        constexpr int kNextWTap = C::TPS == 3 ? ((DXI) + 1) % 3 : 0;                                              \
        if (!S3_DBG(8)) load_frags(f0, (NBUF), n_slot_, kNextWTap, (AON), 0); /* next step, k-step 0 */ \
        mma_part(f1, 0, PhC_{});                                                                                  \
        mma_part(f1, 1, PhC_{});                                                                                  \
        S3_PIPE()                                                                                                 \
        S3_PIPE()                                                                                                 \
        if constexpr (kSlotEnd) {                                                                                 \
            /* everything requested before this barrier interval has landed (this wave's part); then everyone's */ \
            if (S3_DBG(3)) {                                                                                      \
                if (!S3_DBG(1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::DPS) : "memory");                     \
                if (!S3_DBG(2)) asm volatile("s_barrier" ::: "memory");                                           \
            } else if (due_) {                                                                                    \
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(C::DPS) : "memory");                       \
            } else {                                                                                              \
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(C::W_PER_WAVE) : "memory");                \
            }                                                                                                     \
        }                                                                                                         \
        slot = n_slot_;                                                                                           \
    }

    // the request pointer (three slots ahead) has finished a work item's weights: on to the next one's.  (The steps of a work
    // item are a multiple of 3 and the pointer starts 3 ahead: it can only run out after the third step of a tap row.)
#define S3_WRAP()                                                        \
    if (w_s == wsteps) {                                                 \
        w_s = 0;                                                         \
        w_work += wstride;                                             \
        w_nb = w_work < nwork ? w_work / ntiles : w_nb;                  \
        wp = wptr_of(w_work < nwork ? w_nb : t_nb, 0);                   \
    }

#ifdef QMRI_S3_EXPERIMENTS
    unsigned long long ts_loop = S3_DBG(1024) ? S3_NOW() : 0ull;
#endif
    while (true) {
        // ---- one tap row (dy = row - 1) of one 32-channel chunk: three steps ----
        if ((kUnrollChunk || row == 0) && !req_tile_ready && !S3_DBG(4)) {
            // first request for a new tile: where do its halo pixels come from
            int nb_, b_, y0_, x0_, f0_;
            const int rw = req_work < nwork ? req_work : work;  // past the end: re-request this tile (harmless)
            decode_work(rw, nb_, b_, y0_, x0_, f0_);
            set_halo_sources(b_, y0_, x0_, f0_);
            req_tile_ready = true;
        }
        int n_row, n_chunk, n_cbuf;
        bool last;
        if constexpr (kUnrollChunk) {
            // one whole 32-channel chunk, taps unrolled (tap-dependent addresses, halo pieces and -- for the transposed
            // convolution -- the accumulator set are compile-time):
            //   convolution           t = 3 (dy + 1) + (dx + 1), one accumulator set
            //   transposed (phase)    (0,0,0) (0,-1,0) (-1,0,0) (-1,-1,0) | (0,0,1) (-1,0,1) | (0,0,2) (0,-1,2) | (0,0,3)
            n_row = 0;
            n_chunk = chunk + 1;
            n_cbuf = cbuf ^ 1;
            last = n_chunk == A.chunks;
#define S3_T(T, PH, NBUF) S3_STEP((T) / 3, (T) % 3, aoff[S3_SHIFT_ID(T)], aoff[S3_SHIFT_ID(((T) + 1) % 9)], PH, NBUF)
            S3_T(0, 0, cbuf)
            S3_T(1, 0, cbuf)
            S3_T(2, 0, cbuf)
            S3_WRAP()
            S3_T(3, 0, cbuf)
            S3_T(4, DECONV ? 1 : 0, cbuf)
            S3_T(5, DECONV ? 1 : 0, cbuf)
            S3_WRAP()
            S3_T(6, DECONV ? 2 : 0, cbuf)
            S3_T(7, DECONV ? 2 : 0, cbuf)
            S3_T(8, DECONV ? 3 : 0, n_cbuf)
#undef S3_T
        } else {
            // 128-channel blocks: one tap ROW per trip (the unrolled chunk needs > 256 VGPRs there); the row's three
            // A offsets (+ the first of the next row) are recomputed per row -- 24 MFMAs per tap amortise it
            n_row = row + 1;
            n_chunk = chunk;
            n_cbuf = cbuf;
            if (n_row == 3) {
                n_row = 0;
                n_chunk = chunk + 1;
                n_cbuf = cbuf ^ 1;
            }
            last = n_row == 0 && n_chunk == A.chunks;
            int arow[4][RT];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int shift = t < 3 ? (row - 1) * P + (t - 1) : (n_row - 1) * P - 1;
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const int hp = abase[i] + shift;
                    arow[t][i] = hp * 128 + ((hp >> 1) & 1) * 64 + ((khalf ^ ((hp >> 2) & 3)) * 16);
                }
            }
            S3_STEP(row, 0, arow[0], arow[1], 0, cbuf)
            S3_STEP(row, 1, arow[1], arow[2], 0, cbuf)
            S3_STEP(row, 2, arow[2], arow[3], 0, n_cbuf)
            row = n_row;
        }
        S3_WRAP()
        if (n_row == 0) {  // the chunk is finished: the request pointer moves on to the chunk after the next
            advance_req();
            if (req_chunk == 0) req_tile_ready = false;
        }
        chunk = n_chunk;
        cbuf = n_cbuf;
        if (!last) continue;
#ifdef QMRI_S3_EXPERIMENTS
        const unsigned long long ts_main = S3_DBG(1024) ? S3_NOW() : 0ull;
        unsigned long long ts_aff = ts_main;
#endif

        // ======================= epilogue of this work item =======================
        // staging: the halo buffer of the chunk that just finished (cbuf ^ 1 after the advance above), 4 KB + per wave
        if (!S3_DBG(32)) {
            unsigned char *stage = halo + (cbuf ^ 1) * hbuf_bytes + wave * 4096;
            // output pixel of every tile position (2D: computed inline; FLAT: table filled by the first 256 threads)
            if (FLAT) {
                if (tid < kMTile) {
                    int pix = flat_to_pix(t_f0 + tid, P, A.H, A.W, A.B);
                    if (DECONV && pix >= 0) {  // input pixel (b, y, x) -> output pixel (b, 2 y, 2 x) of the 2H x 2W grid
                        const int x = pix % A.W, r = pix / A.W;  // r = b * H + y
                        pix = (2 * r) * (2 * A.W) + 2 * x;
                    }
                    outpix[tid] = pix;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            // output pixel of this lane's store t of row-tile i (lane >> 3 = pixel within a group of 8).  FLAT: from the table, once
            // per work item; 2D: a row base (tile origin: scalar registers) + the column, recomputed per use -- eight VGPRs held across
            // the epilogue were enough to push the halo source pointers of the main loop into scratch
            int opix_flat[FLAT ? RT : 1][4];
            if (FLAT) {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) opix_flat[i][t] = outpix[(wm * RT + i) * 32 + t * 8 + (lane >> 3)];
            }
            auto out_pixel = [&](int i, int t) -> int {
                if (FLAT) return opix_flat[i][t];
                const int yy = t_y0 + wm * RT + i, xx = t_x0 + t * 8 + (lane >> 3);
                if (yy >= A.H) return -1;
                return DECONV ? (2 * (t_b * A.H + yy)) * (2 * A.W) + 2 * xx : (t_b * A.H + yy) * A.W + xx;
            };
            const float winv = A.winv;
            // bias, ReLU, BatchNorm affine in place, one accumulator tile at a time (interleaving the tiles keeps two versions of
            // every 16-register tile alive: 256 VGPRs + spills); the parameters of four channels are one float4 broadcast
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            // bias, ReLU, BatchNorm affine of four neighbouring channels (one float4 broadcast each from LDS).  The accumulators
            // are never modified: every consumer (staging, classifier, pooling) applies the affine to the values it takes.  (Updating
            // the 16-register tiles in place, four elements at a time, kept two versions of every tile alive: 256 VGPRs, the main
            // loop's halo pointers in scratch -- and every scratch reload in the MFMA loop is a vmcnt(0) among the counted waits.)
            struct Prm4 {
                f32x4 b, s, t;
            };
            auto load_prm = [&](int j, int q) -> Prm4 {
                Prm4 p;
                const float *pp = prm + (wn * CT + j) * 32 + 8 * q + 4 * khalf;
                p.b = *reinterpret_cast<const f32x4 *>(pp);
                p.s = *reinterpret_cast<const f32x4 *>(pp + BN);
                p.t = *reinterpret_cast<const f32x4 *>(pp + 2 * BN);
                return p;
            };
            auto affine = [&](float a, const Prm4 &p, int r) -> float {
                float v = fmaf(a, winv, p.b[r]);
                if (A.relu) v = fmaxf(v, 0.f);
                return fmaf(v, p.s[r], p.t[r]);
            };
            const int n0 = t_nb * BN;
            const int px_l = lane & 31;
            constexpr int kYB = ONE ? 2 : 4;  // bytes per output channel: bf16, or fp16 hi + lo
            // staging window: pixel px, 16-byte piece p8 = plane * 4 + q at position p8 ^ ((px >> 1) & 7) (the 8-byte writes of
            // 32 lanes then spread over 16 bank groups instead of 2)
            auto stage_piece = [&](int px, int p8) { return px * 128 + ((p8 ^ ((px >> 1) & 7)) * 16); };
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
            const int ph_off = DECONV ? (ph >> 1) * 2 * A.W + (ph & 1) : 0;  // output phase (py, px) = (ph >> 1, ph & 1)
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int cbase = n0 + (wn * CT + j) * 32;  // first output channel of this column tile
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    // ---- [32 pixels][32 hi | 32 lo] image of the tile in the wave's window ----
                    float z[4] = {0.f, 0.f, 0.f, 0.f};  // (classifier partial sums, BN = 32 with a fused head only)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const Prm4 p = load_prm(j, q);
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = affine(acc[ph][i][j][4 * q + r], p, r);
                        if (!ONE && !DECONV && BN == 32 && A.head_w) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float4 w4 = *reinterpret_cast<const float4 *>(hw + (8 * q + 4 * khalf + r) * 4);
                                z[0] = fmaf(v[r], w4.x, z[0]);
                                z[1] = fmaf(v[r], w4.y, z[1]);
                                z[2] = fmaf(v[r], w4.z, z[2]);
                                z[3] = fmaf(v[r], w4.w, z[3]);
                            }
                        }
                        if (ONE) {
                            if (A.y) {  // 4 bf16 channels = the 8-byte half of piece q: only the "hi" pieces 0-3 of the window are used
                                const bf16x2 b0 = {(__bf16)v[0], (__bf16)v[1]}, b1 = {(__bf16)v[2], (__bf16)v[3]};
                                *reinterpret_cast<uint2 *>(stage + stage_piece(px_l, q) + 8 * khalf) =
                                    make_uint2(__builtin_bit_cast(unsigned, b0), __builtin_bit_cast(unsigned, b1));
                            }
                        } else if (A.y) {
#ifndef QMRI_NO_SAT_TRACK  // (timing experiment: what the saturation tracking costs)
                            amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));  // (two v_max3_f32 with |.| modifiers)
#endif
#ifndef QMRI_NO_SAT_TRACK  // (timing experiment: what the saturation tracking costs)
                            amax = fmaxf(fmaxf(amax, fabsf(v[2])), fabsf(v[3]));
#endif
                            const h16x2 h0 = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), h1 = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
                            const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(v[0] - (float)h0[0], v[1] - (float)h0[1]);
                            const h16x2 l1 = __builtin_amdgcn_cvt_pkrtz(v[2] - (float)h1[0], v[3] - (float)h1[1]);
                            *reinterpret_cast<uint2 *>(stage + stage_piece(px_l, q) + 8 * khalf) =
                                make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                            *reinterpret_cast<uint2 *>(stage + stage_piece(px_l, 4 + q) + 8 * khalf) =
                                make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (A.y) {
                        // four 16-byte stores per lane: 8 lanes = one 128-byte pixel-chunk (pieces permuted by the window
                        // swizzle); the LDS reads are issued together (ONE wait), only the global stores are predicated
#pragma unroll
                        for (int t0 = 0; t0 < 4; t0 += 2) {
                        uint4 v[4];
#pragma unroll
                        for (int t = t0; t < t0 + 2; ++t) v[t] = *reinterpret_cast<const uint4 *>(stage + (t * 8 + (lane >> 3)) * 128 + (lane & 7) * 16);
#pragma unroll
                        for (int t = t0; t < t0 + 2; ++t) {
                            const int pix = out_pixel(i, t);
                            const int p8 = (lane & 7) ^ (((t * 8 + (lane >> 3)) >> 1) & 7);
                            if (pix >= 0 && !S3_DBG(16) && (!ONE || p8 < 4)) {  // (bf16: the four "hi" pieces are the pixel's 64 bytes)
                                long long doff = ((long long)(pix + ph_off) * A.ldy + A.yoff + cbase) * kYB + p8 * 16;
                                if (S3_DBG(128)) doff &= (1ll << 22) - 16;  // experiment: every store lands in one 4 MB window
                                if (S3_DBG(256) && t) continue;              // experiment: one store of four
                                nt_store16(static_cast<unsigned char *>(A.y) + doff, v[t]);
                            }
                        }
                        }
                    }
                    if (!ONE && !DECONV && BN == 32 && A.head_w) {
                        // 1x1 classifier: this lane summed its 16 channels of its pixel above, the partner lane (+32) adds the rest
#pragma unroll
                        for (int c = 0; c < 4; ++c) z[c] += __shfl_xor(z[c], 32, 64);
                        const int pixh = FLAT ? outpix[(wm * RT + i) * 32 + px_l] : (t_y0 + wm * RT + i < A.H ? (t_b * A.H + t_y0 + wm * RT + i) * A.W + t_x0 + px_l : -1);
                        if (khalf == 0 && pixh >= 0) {
                            const long long pix = pixh;
                            const int NC = A.head_nc;
                            float zz[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) zz[c] = z[c] + hw[128 + c];
                            if (NC == 4) {  // one 16-byte store of the pixel's logits, one 4-byte store of its mask bytes
                                if (A.logits) *reinterpret_cast<float4 *>(A.logits + pix * 4) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                                if (A.mask)
                                    *reinterpret_cast<unsigned *>(A.mask + pix * 4) =
                                        (zz[0] > 0.f ? 1u : 0u) | (zz[1] > 0.f ? 0x100u : 0u) | (zz[2] > 0.f ? 0x10000u : 0u) |
                                        (zz[3] > 0.f ? 0x1000000u : 0u);
                            } else {
                                for (int c = 0; c < NC; ++c) {
                                    if (A.logits) A.logits[pix * NC + c] = zz[c];
                                    if (A.mask) A.mask[pix * NC + c] = zz[c] > 0.f ? 1 : 0;
                                }
                            }
                        }
                    }
                }
                // ---- fused MaxPooling2D(2x2): rows rt, rt+1 of this wave (RT = 2) -> 16 pooled pixels x 32 channels ----
                if (!DECONV && !FLAT && RT == 2 && A.pool_y) {
                    // vertical: the two row-tiles of this lane; horizontal: the neighbouring lane (pixel ^ 1); even lanes keep the result
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const Prm4 p = load_prm(j, q);
                        float m[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float vmax = fmaxf(affine(acc[0][0][j][4 * q + r], p, r), affine(acc[0][RT - 1][j][4 * q + r], p, r));
                            const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, vmax), 0xB1, 0xF, 0xF, true));  // quad_perm [1, 0, 3, 2]
                            m[r] = fmaxf(vmax, other);
                        }
                        if (ONE) {
                            const bf16x2 b0 = {(__bf16)m[0], (__bf16)m[1]}, b1 = {(__bf16)m[2], (__bf16)m[3]};
                            if (!(px_l & 1))
                                *reinterpret_cast<uint2 *>(stage + stage_piece(px_l >> 1, q) + 8 * khalf) =
                                    make_uint2(__builtin_bit_cast(unsigned, b0), __builtin_bit_cast(unsigned, b1));
                            continue;
                        }
                        const h16x2 h0 = __builtin_amdgcn_cvt_pkrtz(m[0], m[1]), h1 = __builtin_amdgcn_cvt_pkrtz(m[2], m[3]);
                        const h16x2 l0 = __builtin_amdgcn_cvt_pkrtz(m[0] - (float)h0[0], m[1] - (float)h0[1]);
                        const h16x2 l1 = __builtin_amdgcn_cvt_pkrtz(m[2] - (float)h1[0], m[3] - (float)h1[1]);
                        if (!(px_l & 1)) {
                            const int pp = px_l >> 1;  // pooled column 0..15
                            *reinterpret_cast<uint2 *>(stage + stage_piece(pp, q) + 8 * khalf) =
                                make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
                            *reinterpret_cast<uint2 *>(stage + stage_piece(pp, 4 + q) + 8 * khalf) =
                                make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const int Hp = A.H >> 1, Wp = A.W >> 1;
                    const int yy = (t_y0 >> 1) + wm;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int id = t * 64 + lane, px = id >> 3, pos = id & 7;
                        if (yy < Hp) {
                            const uint4 v = *reinterpret_cast<const uint4 *>(stage + px * 128 + pos * 16);
                            const int p8 = pos ^ ((px >> 1) & 7);
                            const long long pix = (long long)(t_b * Hp + yy) * Wp + (t_x0 >> 1) + px;
                            unsigned char *dst = static_cast<unsigned char *>(A.pool_y) + (pix * A.pool_ld + cbase) * kYB + p8 * 16;
                            if (!ONE || p8 < 4) nt_store16(dst, v);
                        }
                    }
                }
            }
            }  // phases
        }
        // ---- next work item ----
#ifdef QMRI_S3_EXPERIMENTS
        const unsigned long long ts_ep = S3_DBG(1024) ? S3_NOW() : 0ull;
#endif
        work += wstride;
        if (work >= nwork) break;
        const int prev_nb = t_nb;
        decode_work(work, t_nb, t_b, t_y0, t_x0, t_f0);
        zero_acc();
        chunk = 0;
        // every wave is done with its staging window (the next chunk's DMA lands there) and with the parameters
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t_nb != prev_nb) {         // (only with more than one channel block per launch)
            for (int i = tid; i < 3 * BN; i += kThreads) {
                const int c = i % BN, which = i / BN;
                const int n = t_nb * BN + c;
                prm[i] = which == 0 ? (A.bias ? A.bias[n] : 0.f) : which == 1 ? (A.scale ? A.scale[n] : 1.f) : (A.shift ? A.shift[n] : 0.f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#ifdef QMRI_S3_EXPERIMENTS
        if (S3_DBG(1024)) {
            const unsigned long long ts_next = S3_NOW();
            if (tid == 0) {
                atomicAdd(&s3_tstat[0], ts_main - ts_loop);
                atomicAdd(&s3_tstat[1], ts_ep - ts_main);
                atomicAdd(&s3_tstat[2], ts_next - ts_ep);
                atomicAdd(&s3_tstat[3], 1ull);
                atomicAdd(&s3_tstat[4], ts_aff - ts_main);
            }
            ts_loop = ts_next;
        }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this block's DMA may land after it has exited
    if (A.sat && amax > 65504.f) *A.sat = 1;  // (the pooled values are maxima of stored ones: covered)
}

static size_t s3_lds_bytes(int bn, int nj) {
    const size_t slot = (size_t)(bn == 32 ? 3 : 1) * bn * 128;  // S3Cfg::SLOT_BYTES
    return (size_t)2 * nj * 1024 + (size_t)kRing * slot + kMTile * 4 + (size_t)3 * bn * 4 + (32 * 4 + 4) * 4;
}

bool conv_s3_supported(const ConvS3Args &k) {
    if (k.Cin % 32 || k.Cout % 32) return false;
    if (k.W % 32 == 0) return true;      // 8 x 32 tiles
    return k.W + 2 <= 50;                // flattened zero-framed stack (LDS: 256 + 2 (W + 2) + 2 halo pixels)
}

template <int BN, bool FLAT, bool DECONV, bool ONE = false>
static hipError_t s3_launch_t(ConvS3Args &k, int num_cu, hipStream_t stream) {
    auto fn = conv_s3_kernel<BN, FLAT, DECONV, ONE>;
    const size_t lds = s3_lds_bytes(BN, k.nj);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    int grid = k.nwork < num_cu ? k.nwork : num_cu;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, stream, k);
    return hipGetLastError();
}

// channel-block size the kernel uses for a layer (the weight packing depends on it).  The transposed convolution keeps
// four accumulator sets: 32 channels per block (64: 4 x 32 + 72 operand registers -> 256 VGPRs and 170 B of scratch).
int conv_s3_block_channels(int Cout, int deconv) {
    if (deconv) return 32;
    return Cout % 128 == 0 ? 128 : (Cout % 64 == 0 ? 64 : 32);
}

// Which kernel for a layer both support?  conv_c4_kernel (unet_c4.hip: one wave per SIMD, 4 x 4 or 6 x 2 register tiles) issues
// MFMAs faster but works in bigger items (512 positions x 128 channels / 768 x 64); QMRI_C4 = 0 never / 1 by the rule below
// (default) / 2 wherever it is supported.
bool conv_s3_takes_c4(const ConvS3Args &k, int num_cu) {
    static const int mode = [] {
        const char *e = std::getenv("QMRI_C4");
        return e ? std::atoi(e) : 1;
    }();
    if (!k.w_c4 || k.c4_mode < 0 || !conv_c4_supported(k)) return false;
    if (k.c4_mode > 0 || mode >= 2) return true;
    if (mode <= 0) return false;
    // The choice depends on the LAYER only (level geometry and channel counts), never on the batch: the two kernels add the same
    // products in different orders (tap-major against half-chunk-major), and a slice's logits must not depend on how many slices
    // travel with it (tests/test_unet_fullsize_gpu.py::test_forward_is_bitwise_repeatable runs one volume through engines of two
    // batch sizes).  Since conv_c4_kernel's epilogue writes the accumulators to LDS as they are (round 4, profiles/r04_c4_ab.txt)
    // it wins on every layer it supports -- 128-channel blocks at any depth (2 chunks: 567 against 620 us; flattened, 4 chunks: 533
    // against 584) and, on 24-row image tiles, the 64-channel layers of the 192 x 192 level (701 / 1165 / 2015 / 1153 us against
    // 709 / 1234 / 2127 / 1248 for conv_s3_kernel<64>); 16 / 48 slices per forward and the 512 x 512 network: +-0 / +1.3 % / +1.4 %.
    // Before that its per-item costs tied or lost wherever an item had fewer than 4 (flattened: 8) input chunks.
    // (64-channel blocks on the FLATTENED levels -- 4 x 2 tiles, 24 MFMAs per k-step -- were 5-10 % slower than conv_s3_kernel<64>
    //  when last measured and appear in no network of the bench: kept for tests, not picked)
    (void)num_cu;
    return !(conv_c4_block_channels(k.Cout) == 64 && conv_tiles_flat(k.W));
}

// The transposed convolutions: deconv_d4_kernel (unet_d4.hip) wherever it supports the layer -- by layer shape only, like
// conv_s3_takes_c4.  QMRI_D4 = 0: conv_s3_kernel<32, *, DECONV> (round 2's kernel; the A/B switch of profiles/r05_d4_ab.txt).
bool conv_s3_takes_d4(const ConvS3Args &k) {
    static const int mode = [] {
        const char *e = std::getenv("QMRI_D4");
        return e ? std::atoi(e) : 1;
    }();
    if (!k.deconv || !k.w_c4 || k.c4_mode < 0 || !conv_d4_supported(k)) return false;
    return k.c4_mode > 0 || mode > 0;
}

hipError_t conv_s3_launch(const ConvS3Args &k0, int num_cu, hipStream_t stream) {
    ConvS3Args k = k0;
    if (conv_s3_takes_d4(k)) return conv_d4_launch(k, num_cu, stream);
    if (conv_s3_takes_c4(k, num_cu)) return conv_c4_launch(k, num_cu, stream);
    if (k.c4_mode > 0) return hipErrorInvalidValue;
    if (!conv_s3_supported(k)) return hipErrorInvalidValue;
    const int bn = conv_s3_block_channels(k.Cout, k.deconv);
    const bool flat = k.W % 32 != 0;
    if (k.head_w && (bn != 32 || k.Cout != 32 || flat || k.deconv || k.head_nc < 1 || k.head_nc > 4)) return hipErrorInvalidValue;
    if (k.pool_y && (flat || k.deconv || (k.H & 1) || (k.W & 1) || bn == 32)) return hipErrorInvalidValue;
    k.chunks = k.Cin / 32;
    k.steps = k.chunks * 9;
    k.nb = k.Cout / bn;
    if (flat) {
        k.P = k.W + 2;
        const long long span = (long long)k.B * (k.H + 1) * k.P - k.P;  // flat positions [P, B (H+1) P)
        k.ntiles = (int)((span + kMTile - 1) / kMTile);
        k.tiles_x = k.tiles_y = 0;
        k.nj = (kMTile + 2 * k.P + 2 + 7) / 8;
    } else {
        k.P = kPitch2D;
        k.tiles_x = k.W / 32;
        k.tiles_y = (k.H + 7) / 8;
        k.ntiles = k.B * k.tiles_x * k.tiles_y;
        k.nj = (kHalo2D + 7) / 8;
    }
    k.nwork = k.nb * k.ntiles;
    static const int dbg = [] {
        const char *e = std::getenv("QMRI_S3_DBG"), *d = std::getenv("QMRI_S3_DELAY");
        return (e ? std::atoi(e) : 0) | (d ? std::atoi(d) << 16 : 0);
    }();
    k.dbg = dbg;
    (void)hipGetLastError();
    if (k.one) {  // plain bf16 (k.Cin, k.ldx, k.xoff are in 4-byte units: see the kernel's header comment)
        if (k.head_w) return hipErrorInvalidValue;
        if (k.deconv) return flat ? s3_launch_t<32, true, true, true>(k, num_cu, stream) : s3_launch_t<32, false, true, true>(k, num_cu, stream);
        if (flat) {
            if (bn == 128) return s3_launch_t<128, true, false, true>(k, num_cu, stream);
            if (bn == 64) return s3_launch_t<64, true, false, true>(k, num_cu, stream);
            return s3_launch_t<32, true, false, true>(k, num_cu, stream);
        }
        if (bn == 128) return s3_launch_t<128, false, false, true>(k, num_cu, stream);
        if (bn == 64) return s3_launch_t<64, false, false, true>(k, num_cu, stream);
        return s3_launch_t<32, false, false, true>(k, num_cu, stream);
    }
    if (k.deconv) return flat ? s3_launch_t<32, true, true>(k, num_cu, stream) : s3_launch_t<32, false, true>(k, num_cu, stream);
    if (flat) {
        if (bn == 128) return s3_launch_t<128, true, false>(k, num_cu, stream);
        if (bn == 64) return s3_launch_t<64, true, false>(k, num_cu, stream);
        return s3_launch_t<32, true, false>(k, num_cu, stream);
    }
    if (bn == 128) return s3_launch_t<128, false, false>(k, num_cu, stream);
    if (bn == 64) return s3_launch_t<64, false, false>(k, num_cu, stream);
    return s3_launch_t<32, false, false>(k, num_cu, stream);
}

// =====================================================================================================================
// Small streaming kernels on the split layout (all HBM-bound): first layer, 2x2 max-pool, 1x1 head, layout casts.
// One thread = one (pixel, 8-channel group): 16 bytes of hi parts + 16 bytes of lo parts.
// =====================================================================================================================
namespace {

__device__ __forceinline__ void split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h16x2 hh = __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]);
        const h16x2 ll = __builtin_amdgcn_cvt_pkrtz(v[2 * q] - (float)hh[0], v[2 * q + 1] - (float)hh[1]);
        h[q] = __builtin_bit_cast(unsigned, hh);
        l[q] = __builtin_bit_cast(unsigned, ll);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void join8(const uint4 &hi, const uint4 &lo, float (&v)[8]) {
    const unsigned h[4] = {hi.x, hi.y, hi.z, hi.w}, l[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const h16x2 hh = __builtin_bit_cast(h16x2, h[q]), ll = __builtin_bit_cast(h16x2, l[q]);
        v[2 * q] = (float)hh[0] + (float)ll[0];
        v[2 * q + 1] = (float)hh[1] + (float)ll[1];
    }
}
// byte address of the 8-channel group g (channels 8 g .. 8 g + 7) of a pixel's split run starting at `base`
__device__ __forceinline__ long long split_group_off(int g) { return (long long)(g >> 2) * 128 + (g & 3) * 16; }

// Conv2D(C, 3x3, SAME) on ONE input channel + bias + ReLU (oaiunet2d.py:213-219 on the image), fp32 VALU, split output
// A thread owns 8 output channels of one image column over a strip of kC1Rows rows: its 72 weights stay in registers and the
// 3x3 window slides down (3 loads per pixel instead of 9 + 18 weight loads -- the first version was load-issue bound).
constexpr int kC1Rows = 16;
__global__ __launch_bounds__(256) void c1_split_kernel(const float *__restrict__ x, int B, int H, int W,
                                                       const float *__restrict__ w /*[9][C]*/, const float *__restrict__ bias,
                                                       int Cout, unsigned char *__restrict__ y, long long ldy, int yoff, int *sat) {
    const int groups = Cout / 8;
    const int strips = (H + kC1Rows - 1) / kC1Rows;
    float amax = 0.f;  // saturation flag of the split layout (ConvS3Args::sat)
    const long long total = (long long)B * strips * W * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        long long t = idx / groups;
        const int xw = (int)(t % W);
        t /= W;
        const int y0 = (int)(t % strips) * kC1Rows;
        const long long b = t / strips;
        const float *img = x + b * (long long)H * W;
        float wt[9][8], bs[8];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) wt[k][c] = w[k * Cout + g * 8 + c];
#pragma unroll
        for (int c = 0; c < 8; ++c) bs[c] = bias[g * 8 + c];
        const bool xl = xw > 0, xr = xw + 1 < W;
        auto load_row = [&](int yy, float (&r)[3]) {
            const bool in = yy >= 0 && yy < H;
            const float *row = img + (long long)yy * W + xw;
            r[0] = in && xl ? row[-1] : 0.f;
            r[1] = in ? row[0] : 0.f;
            r[2] = in && xr ? row[1] : 0.f;
        };
        float win[3][3];
        load_row(y0 - 1, win[0]);
        load_row(y0, win[1]);
#pragma unroll 4
        for (int r = 0; r < kC1Rows; ++r) {
            const int yh = y0 + r;
            if (yh >= H) break;
            load_row(yh + 1, win[2]);
            float acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = bs[c];
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(win[k / 3][k % 3], wt[k][c], acc[c]);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[c] = fmaxf(acc[c], 0.f);
#ifndef QMRI_NO_SAT_TRACK  // (timing experiment: what the saturation tracking costs)
                amax = fmaxf(amax, acc[c]);
#endif
            }
            uint4 hi, lo;
            split8(acc, hi, lo);
            const long long pix = (b * H + yh) * W + xw;
            unsigned char *dst = y + (pix * ldy + yoff) * 4 + split_group_off(g);
            *reinterpret_cast<uint4 *>(dst) = hi;
            *reinterpret_cast<uint4 *>(dst + 64) = lo;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                win[0][q] = win[1][q];
                win[1][q] = win[2][q];
            }
        }
    }
    if (sat && amax > 65504.f) *sat = 1;
}

// MaxPooling2D(k x k), k = 2 or 3 (oaiunet2d.py:234-243), split in (pixel stride ldx, offset xoff) -> compact split out
__global__ __launch_bounds__(256) void maxpool2_split_kernel(const unsigned char *__restrict__ x, long long ldx, int xoff, int B,
                                                             int H, int W, int C, int K, unsigned char *__restrict__ y) {
    const int Ho = H / K, Wo = W / K, groups = C / 8;
    const long long total = (long long)B * Ho * Wo * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        const long long p = idx / groups;
        const int xo = (int)(p % Wo);
        const long long t = p / Wo;
        const int yo = (int)(t % Ho);
        const long long b = t / Ho;
        const long long p00 = (b * H + K * yo) * W + K * xo;
        float m[8];
        for (int q = 0; q < K * K; ++q) {
            const long long pp = p00 + (q / K) * W + (q % K);
            const unsigned char *src = x + (pp * ldx + xoff) * 4 + split_group_off(g);
            float v[8];
            join8(*reinterpret_cast<const uint4 *>(src), *reinterpret_cast<const uint4 *>(src + 64), v);
#pragma unroll
            for (int c = 0; c < 8; ++c) m[c] = q == 0 ? v[c] : fmaxf(m[c], v[c]);
        }
        uint4 hi, lo;
        split8(m, hi, lo);
        unsigned char *dst = y + p * C * 4 + split_group_off(g);
        *reinterpret_cast<uint4 *>(dst) = hi;
        *reinterpret_cast<uint4 *>(dst + 64) = lo;
    }
}

// Conv2D(n_classes <= 4, 1x1) head on a split feature map: Cin / 8 lanes per pixel, wave-shuffle reduction
__global__ __launch_bounds__(256) void head_split_kernel(const unsigned char *__restrict__ x, long long npix, int Cin,
                                                         const float *__restrict__ w /*[Cin][NC]*/, const float *__restrict__ bias,
                                                         int NC, float *__restrict__ logits, unsigned char *__restrict__ mask) {
    const int lpp = Cin / 8;  // lanes per pixel: 4 (Cin = 32) .. 32 (Cin = 256), a power of two
    const int sub = threadIdx.x % lpp;
    const long long ppb = 256 / lpp;
    for (long long p = (long long)blockIdx.x * ppb + threadIdx.x / lpp; p < npix; p += (long long)gridDim.x * ppb) {
        const unsigned char *src = x + p * Cin * 4 + split_group_off(sub);
        float v[8];
        join8(*reinterpret_cast<const uint4 *>(src), *reinterpret_cast<const uint4 *>(src + 64), v);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q)
            for (int c = 0; c < NC; ++c) acc[c] = fmaf(v[q], w[(sub * 8 + q) * NC + c], acc[c]);
        for (int o = lpp / 2; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
        if (sub < NC) {
            const float z = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3]) + bias[sub];
            if (logits) logits[p * NC + sub] = z;
            if (mask) mask[p * NC + sub] = z > 0.f ? 1 : 0;
        }
    }
}

// fp32 NHWC [pix][C] <-> split [pix][C] (operator-level host entry and tests)
__global__ void f32_to_split_kernel(const float *__restrict__ x, long long npix, int C, unsigned char *__restrict__ y) {
    const int groups = C / 8;
    const long long total = npix * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        const long long p = idx / groups;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = x[p * C + g * 8 + q];
        uint4 hi, lo;
        split8(v, hi, lo);
        unsigned char *dst = y + p * C * 4 + split_group_off(g);
        *reinterpret_cast<uint4 *>(dst) = hi;
        *reinterpret_cast<uint4 *>(dst + 64) = lo;
    }
}
__global__ void split_to_f32_kernel(const unsigned char *__restrict__ x, long long npix, int C, float *__restrict__ y) {
    const int groups = C / 8;
    const long long total = npix * groups;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups);
        const long long p = idx / groups;
        const unsigned char *src = x + p * C * 4 + split_group_off(g);
        float v[8];
        join8(*reinterpret_cast<const uint4 *>(src), *reinterpret_cast<const uint4 *>(src + 64), v);
#pragma unroll
        for (int q = 0; q < 8; ++q) y[p * C + g * 8 + q] = v[q];
    }
}

unsigned grid_for(long long total) {
    long long blocks = (total + 255) / 256;
    if (blocks > 65535 * 4) blocks = 65535 * 4;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

hipError_t c1_split_launch(const float *x, int B, int H, int W, const float *w, const float *bias, int Cout, void *y,
                           long long ldy, int yoff, int *sat, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(c1_split_kernel, dim3(grid_for((long long)B * ((H + kC1Rows - 1) / kC1Rows) * W * (Cout / 8))), dim3(256), 0, stream, x, B, H, W, w, bias,
                       Cout, static_cast<unsigned char *>(y), ldy, yoff, sat);
    return hipGetLastError();
}
hipError_t maxpoolk_split_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, int K, void *y,
                                 hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(maxpool2_split_kernel, dim3(grid_for((long long)B * (H / K) * (W / K) * (C / 8))), dim3(256), 0, stream,
                       static_cast<const unsigned char *>(x), ldx, xoff, B, H, W, C, K, static_cast<unsigned char *>(y));
    return hipGetLastError();
}
hipError_t maxpool2_split_launch(const void *x, long long ldx, int xoff, int B, int H, int W, int C, void *y, hipStream_t stream) {
    return maxpoolk_split_launch(x, ldx, xoff, B, H, W, C, 2, y, stream);
}
hipError_t head_split_launch(const void *x, long long npix, int Cin, const float *w, const float *bias, int NC, float *logits,
                             unsigned char *mask, hipStream_t stream) {
    if (NC > 4 || Cin % 32 || Cin > 256 || (Cin & (Cin - 1))) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(head_split_kernel, dim3(grid_for(npix * (Cin / 8))), dim3(256), 0, stream,
                       static_cast<const unsigned char *>(x), npix, Cin, w, bias, NC, logits, mask);
    return hipGetLastError();
}
hipError_t split_cast_launch(const void *x, long long npix, int C, void *y, int to_split, hipStream_t stream) {
    (void)hipGetLastError();
    const unsigned grid = grid_for(npix * (C / 8)) > 8192 ? 8192 : grid_for(npix * (C / 8));
    if (to_split)
        hipLaunchKernelGGL(f32_to_split_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const float *>(x), npix, C,
                           static_cast<unsigned char *>(y));
    else
        hipLaunchKernelGGL(split_to_f32_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const unsigned char *>(x), npix, C,
                           static_cast<float *>(y));
    return hipGetLastError();
}

}  // namespace qmri

#ifdef QMRI_S3_EXPERIMENTS
extern "C" int qmri_s3_debug_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(qmri::s3_tstat), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(qmri::s3_tstat), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
