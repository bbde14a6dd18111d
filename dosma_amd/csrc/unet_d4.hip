// unet_d4.hip -- the parity-mode ("fp16x3") TRANSPOSED convolution on the one-wave-per-SIMD register budget, round 5.
//
// Layer: Conv2DTranspose(3x3, strides 2, SAME) + bias of /root/reference/dosma/models/oaiunet2d.py:259-261 on SPLIT activations
// (fp16 hi + lo parts, a product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate):
//     out[2 y + ky, 2 x + kx] += in[y, x] * w[ky, kx]      (TF's SAME alignment for stride 2: pad_before = 0)
// i.e. output phase (py, px) of input position (y, x) -- the output pixel (2 y + py, 2 x + px) -- sums
//     (0,0): in[y,x] w00 + in[y,x-1] w02 + in[y-1,x] w20 + in[y-1,x-1] w22      (0,1): in[y,x] w01 + in[y-1,x] w21
//     (1,0): in[y,x] w10 + in[y,x-1] w12                                        (1,1): in[y,x] w11
// Nine tap products per input position, four accumulators: 2.25 taps per output value -- a quarter of a 3 x 3 convolution's
// arithmetic per stored value, which is why this layer's epilogue weighs four times what conv_c4_kernel's does.
//
// Why another kernel.  conv_s3_kernel<32, *, DECONV> (unet_s3.hip) runs it on 8 waves of 2 row-tiles x 4 phases (round 2):
// MfmaUtil 0.36-0.62, 3-10 % LDS bank conflicts, 281-333 TF (profiles/r04g_*).  This is the same layer in conv_c4_kernel's form
// (unet_c4.hip -- read that file's header first):
//   * block = 4 waves, __launch_bounds__(256, 1); a wave's register tile = 4 row-tiles (32 input positions each) x 4 phases x 32
//     output channels = 256 AccVGPRs.  Block tile = 512 input positions (16 image rows x 32 pixels, or 512 positions of the
//     flattened zero-framed stack) x 32 output channels -> 32 x 64 output pixels.
//   * K runs in k-steps of 16 input channels (a half-chunk), and a k-step in four SHIFT GROUPS -- the taps that read the same
//     shifted input fragment share it:
//         A: shift ( 0, 0): taps (0,0) (0,1) (1,0) (1,1) -> phases 0 1 2 3      48 MFMAs per wave   16 fragment reads
//         B: shift ( 0,-1): taps (0,2) (1,2)             -> phases 0 2          24                  12
//         C: shift (-1, 0): taps (2,0) (2,1)             -> phases 0 1          24                  12
//         D: shift (-1,-1): tap  (2,2)                   -> phase  0            12                  10
//     108 MFMAs from 50 ds_read_b128 per wave and k-step; the operands of a group are read while the previous group multiplies
//     (two fragment sets of 16, like conv_c4_kernel).
//   * halo of a half-chunk: conv_c4_kernel's image (64 B per pixel, 39 KB, same swizzle, same tile origin; the row / column
//     beyond the tile that only a 3 x 3 convolution needs is not fetched), two buffers.  Weights: a ring of THREE k-step slots of
//     18 KB = nine taps x [plane][32 channels][2 x 16 B] in group order.  All by buffer_load_dwordx4 ... lds.
//   * ring protocol: ONE s_barrier per k-step, between groups C and D.  When a wave passes the barrier of k-step u every wave has
//     issued its last operand reads of k-step u (group D's, during C), so from there on halo buffer u & 1 takes the halo of k-step
//     u + 2 and the weight slot of k-step u - 1 ... has long been free: requests per wave and k-step
//         D(u-1): 2 halo pieces of k-step u + 1 | A(u): 5 | B(u): 3 | C(u): 3 weight pieces of k-step u + 2 | barrier | D(u): 2
//     and the counted wait in front of the barrier lets exactly C's three stay in flight: a halo piece has at least group C, a
//     weight piece a whole k-step, to land.  (The LAST k-step of a work item requests no halo in D: buffer 1 is the epilogue's
//     staging area; next_item issues those two pieces.)
//   * epilogue: conv_c4_kernel's -- accumulators leave the register file raw (ds_write_b128 from AccVGPRs into two wave-private
//     4 KB windows), come back pixel-major, bias (+ ReLU / affine where a caller asks), split, two 16-byte stores per pixel and
//     channel octet -- with the phase where that kernel has the column tile: tile (row-tile i, phase ph) goes to output row
//     2 y + py, pixels 2 x + px.
//   * which layers run here: conv_d4_supported -- by layer shape only (a slice's bits must not depend on the batch).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "qmri_internal.h"
#include "unet_c4_common.h"

namespace qmri {

namespace {

using namespace c4;

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kPitch2D = 34;
constexpr int kRT = 4;                      // 32-position row-tiles per wave
constexpr int kRows = 4 * kRT;              // image rows of a block tile
constexpr int kMTile = kRows * 32;          // input positions per block tile
constexpr int kHalo2D = (kRows + 2) * kPitch2D;  // conv_c4_kernel's halo geometry (tile origin at halo (1, 1)); row kRows + 1 and column 33 stay unfetched
constexpr int kNJ = (kHalo2D + 15) / 16;    // DMA instructions (16 pixels x 64 B) per halo buffer: 39
constexpr int kHBuf = kNJ * 1024;
constexpr int kHSlots = (kNJ + 3) / 4;      // halo pieces per wave: 10
constexpr int kStage = 8192;                // epilogue staging per wave: two 4 KB windows
constexpr int kTapBytes = 2048;             // one tap of a k-step: [plane][32 channels][2 x 16 B]
constexpr int kWSlot = 9 * kTapBytes;       // weights of a k-step
constexpr int kWRing = 3;                   // k-step slots in LDS
constexpr int kPrmBlocks = 4;               // channel blocks whose epilogue parameters the LDS holds at once (tile-major order: Cout <= 128)
constexpr int kWPieces = 5;                 // weight DMA instructions per wave and k-step (18 pieces of 1 KB: waves 0, 1 five, waves 2, 3 four + a repeat)
// requests per group (see the header): halo pieces of the NEXT k-step (D starts the one after), weight pieces two k-steps ahead.
// Within a group the halo pieces are issued first.  QMRI_D4_SCHED picks a schedule at compile time (A/B builds).
#ifndef QMRI_D4_SCHED
#define QMRI_D4_SCHED 1
#endif
#if QMRI_D4_SCHED == 0     // the first version: the last halo pieces have only group C (0.45 us) to land
constexpr int kHaloD = 2, kHaloA = 5, kHaloB = 3;
constexpr int kWgtA = 0, kWgtB = 0, kWgtC = 3, kWgtD = 2;
#elif QMRI_D4_SCHED == 1   // halo as early as the two buffers allow: D (right behind the barrier that frees the buffer) and A
constexpr int kHaloD = 4, kHaloA = 6, kHaloB = 0;
constexpr int kWgtA = 0, kWgtB = 2, kWgtC = 3, kWgtD = 0;
#elif QMRI_D4_SCHED == 2
constexpr int kHaloD = 3, kHaloA = 7, kHaloB = 0;
constexpr int kWgtA = 0, kWgtB = 2, kWgtC = 3, kWgtD = 0;
#else                      // 3: everything early
constexpr int kHaloD = 4, kHaloA = 6, kHaloB = 0;
constexpr int kWgtA = 0, kWgtB = 3, kWgtC = 2, kWgtD = 0;
#endif
static_assert(kHaloD + kHaloA + kHaloB == kHSlots && kWgtA + kWgtB + kWgtC + kWgtD == kWPieces, "request schedule");
// what the counted wait in front of a k-step's barrier leaves in flight: the weight pieces (needed a k-step later) issued behind the
// k-step's last halo piece
constexpr int kWaitN = kHaloB > 0 ? kWgtB + kWgtC : kWgtA + kWgtB + kWgtC;

// taps of a k-step slot in group order; phase of tap slot s; first slot / number of taps of group g (A B C D)
__host__ __device__ constexpr int d4_phase(int s) { return s < 4 ? s : (s == 4 ? 0 : s == 5 ? 2 : s == 6 ? 0 : s == 7 ? 1 : 0); }
__host__ __device__ constexpr int d4_first(int g) { return g == 0 ? 0 : g == 1 ? 4 : g == 2 ? 6 : 8; }
__host__ __device__ constexpr int d4_ntaps(int g) { return g == 0 ? 4 : g == 3 ? 1 : 2; }

struct Frags {
    f16x8 ah[kRT], al[kRT];  // input fragments of the group's shift, one per row-tile (hi / lo plane)
    f16x8 wh[4], wl[4];      // weight fragments of the group's taps
};

// scheduling hints of a group of kRT row-tiles with NM MFMAs each: every row-tile's MFMAs open with an MFMA, the R<r> reads of part r of
// the NEXT group's operands ride one per MFMA behind it
template <int NM, int NR>
__device__ __forceinline__ void d4_sched_row() {
    constexpr int NP = NR < NM ? NR : NM;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if constexpr (NR > NP) __builtin_amdgcn_sched_group_barrier(0x100, NR - NP, 0);
    if constexpr (NM > NP) __builtin_amdgcn_sched_group_barrier(0x008, NM - NP, 0);
}
template <int NM, int R0, int R1, int R2, int R3>
__device__ __forceinline__ void d4_sched_rows() {
    d4_sched_row<NM, R0>();
    d4_sched_row<NM, R1>();
    d4_sched_row<NM, R2>();
    d4_sched_row<NM, R3>();
}

}  // namespace

#ifdef QMRI_D4_EXPERIMENTS  // timing experiments (results wrong by construction): QMRI_D4_DBG = 1 no epilogue | 4 no MFMAs | 8 no LDS operand reads | 16 no DMA requests | 32 no global stores | 64 no barrier in the k loop | 128 no counted wait
#define D4_DBG(bit) (A.dbg & (bit))
#else
#define D4_DBG(bit) 0
#endif

template <bool FLAT>
__global__ __launch_bounds__(kThreads, 1) void deconv_d4_kernel(const ConvS3Args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *halo = smem;                                             // [2][kNJ * 16 pixels][64 B swizzled]
    unsigned char *ring = smem + 2 * kHBuf;                                 // [kWRing][9 taps][2 planes][32][32 B swizzled]
    int *outpix = reinterpret_cast<int *>(ring + kWRing * kWSlot);          // [kMTile] output pixel (phase 0) of a tile position, or -1 (FLAT only)
    float *prm = reinterpret_cast<float *>(outpix + (FLAT ? kMTile : 0));   // bias | scale | shift, [32] each; tile-major order: of every channel block
    unsigned *hofft = reinterpret_cast<unsigned *>(prm + 3 * 32 * kPrmBlocks);  // [kHSlots][256] per-lane halo source offsets

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5;
    const int P = FLAT ? A.P : kPitch2D;
    const int ntiles = A.ntiles;
    const int ksteps = A.steps;  // 2 * chunks k-steps per work item

    i32x4 xr = make_rsrc(A.x);   // (base moves with the work item: see conv_c4_kernel)
    const i32x4 wr = make_rsrc(A.w_c4);
    const long long img_bytes = (long long)A.H * A.W * A.ldx * 4;
    const unsigned halo_lds = lds_off(halo), ring_lds = lds_off(ring);

    // ---- work distribution: XCD x walks a contiguous eighth of the work items (one channel block's weights and neighbouring tiles
    // stay in one L2), its blocks striding through it -- conv_c4_kernel's, without the channel split (a block is one column tile) ----
    int lo = 0, hi = A.nwork, lb = blockIdx.x, nblk = gridDim.x;
    if ((gridDim.x & 7) == 0 && A.nwork >= 64) {
        const int per_xcd = (A.nwork + 7) >> 3;
        const int xcd = blockIdx.x & 7;
        lo = xcd * per_xcd;
        hi = lo + per_xcd < A.nwork ? lo + per_xcd : A.nwork;
        lb = blockIdx.x >> 3;
        nblk = gridDim.x >> 3;
    }
    const int n_range = hi > lo ? hi - lo : 0;
    const int my_items = n_range > lb ? (n_range - lb + nblk - 1) / nblk : 0;
    if (my_items == 0) return;
    auto item_at = [&](int k) -> int {
        if (k >= my_items) k = my_items - 1;  // (past the end: the last one again -- requests made for it are never read)
        return lo + lb + k * nblk;
    };

    int t_nb = 0, t_b = 0, t_y0 = 0, t_x0 = 0, t_f0 = 0;
    // work item -> (channel block, tile).  Channel-major (an XCD's blocks share one channel block's weights, tiles re-read per block)
    // where the layer's weights are bigger than an L2 can keep; TILE-MAJOR (A.tile_group, set by the launcher: a layer's whole
    // weight group <= 2.4 MB, the launcher's budget) where they fit: the channel blocks of a tile run side by side on one XCD, the tile's halo comes from HBM
    // once and from that L2 for the other blocks -- the transposed convolutions of the 96 x 96 and 48 x 48 levels moved 2.1-4.2 x
    // their input through the fabric (profiles/r04g_unet_reads_by_layer.txt) at 3.6-4.1 TB/s of total traffic.
    const int nbk = A.nb;
    // G = A.tile_group: channel blocks per GROUP (0 / 1: channel-major).  Items run group by group; inside a group tile by tile, the
    // group's G channel blocks of a tile side by side: w = (group * ntiles + tile) * G + j, channel block = group * G + j.  The launcher
    // picks the largest G (<= kPrmBlocks) whose G blocks of weights stay in an XCD's L2 (<= 2.4 MB): the layer's input is then read
    // nb / G times instead of nb times (up3: 2 instead of 8, up4: 8 instead of 16; up2 / up1: once).
    const int G = A.tile_group;
    const bool tile_major = G > 1;
    const int per_group = ntiles * (G > 1 ? G : 1);
    auto decode_work = [&](int w, int &nb, int &b, int &y0, int &x0, int &f0) {
        int t;
        if (tile_major) {
            const int cg = w / per_group, r = w - cg * per_group;
            t = __builtin_amdgcn_readfirstlane(r / G);  // (wave-uniform; the divisions run on the vector ALU)
            nb = __builtin_amdgcn_readfirstlane(cg * G + (r - t * G));
        } else {
            nb = w / ntiles;
            t = w - nb * ntiles;
        }
        if (FLAT) {
            f0 = A.P + t * kMTile;
            b = y0 = x0 = 0;
        } else {
            const int per_img = A.tiles_y * A.tiles_x;
            b = t / per_img;
            const int r = t - b * per_img;
            const int ty = r / A.tiles_x;
            y0 = ty * kRows;
            x0 = (r - ty * A.tiles_x) * 32;
            f0 = 0;
        }
    };
    int cur = 0;  // ordinal of the item being computed
    decode_work(item_at(0), t_nb, t_b, t_y0, t_x0, t_f0);

    // ---- halo requests: piece i of this wave is DMA instruction j = wave + 4 i (16 pixels x 64 B) of a halo buffer; lane -> halo
    // pixel hp = 16 j + (lane >> 2), LDS position lane & 3 holds piece c = pos ^ ((hp >> 2) & 3) = plane * 2 + g (conv_c4_kernel's
    // image).  hofft[i][tid]: the lane's source offset for the tile being REQUESTED (chunk 0, half 0), or kPadOff (zeros). ----
    auto set_halo_sources = [&](int b, int y0, int x0, int f0) {
        int lane_h = lane;
        asm volatile("" : "+v"(lane_h));  // (opaque: the per-piece pixel coordinates are otherwise computed before the main loop and spilled)
        const int lane = lane_h;
        int b0 = b;
        if (FLAT) {
            const int r1 = (f0 - P - 1) / P - 1;
            b0 = r1 > 0 ? r1 / (A.H + 1) : 0;
            if (b0 > A.B - 1) b0 = A.B - 1;
        }
        xr = make_rsrc(static_cast<const unsigned char *>(A.x) + (long long)b0 * img_bytes);
        const int pix0 = b0 * A.H * A.W;
#pragma unroll
        for (int i = 0; i < kHSlots; ++i) {
            int j = wave + kWaves * i;
            if (j >= kNJ) j = wave;
            const int hp = j * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((hp >> 2) & 3);
            const unsigned srcb = (unsigned)((c >> 1) * 64 + (c & 1) * 16);
            int pix = -1;
            if (FLAT) {
                // halo pixel hp = flat position f0 - P - 1 + hp; the shifts are 0, -1, -P, -P - 1: nothing beyond the tile's last position
                if (hp < kMTile + P + 1) pix = flat_to_pix(f0 - P - 1 + hp, P, A.H, A.W, A.B);
            } else {
                const int hy = hp / kPitch2D, hx = hp - hy * kPitch2D;
                const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                if (hy <= kRows && hx <= 32 && (unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W) pix = (b * A.H + yy) * A.W + xx;
            }
            hofft[i * kThreads + tid] = pix >= 0 ? (unsigned)(pix - pix0) * (unsigned)A.ldx * 4u + srcb : kPadOff;
        }
    };
    // the k-step (of the item being requested) whose halo is being requested; its buffer = req_u & 1
    int req_k = 0, req_u = 0;
    auto halo_soff = [&](int u) -> unsigned { return (unsigned)(A.xoff * 4 + (u >> 1) * 128 + (u & 1) * 32); };
    auto issue_halo_at = [&](int i, int u, unsigned voff) {
        int j = wave + kWaves * i;
        if (j >= kNJ) j = wave;
        if (D4_DBG(16)) return;
        dma_buf16(voff, xr, halo_soff(u), halo_lds + (unsigned)((u & 1) * kHBuf + j * 1024));
    };
    auto issue_halo = [&](int i, int u) { issue_halo_at(i, u, hofft[i * kThreads + tid]); };
    // the request pointer moves on to the next k-step's halo (group D; the prologue; next_item): past the item's last k-step it
    // enters the next item, whose halo sources are computed right there
    auto advance_halo_req = [&]() -> bool {
        ++req_u;
        if (req_u != ksteps) return false;
        req_u = 0;
        ++req_k;
        int nb_, b_, y0_, x0_, f0_;
        decode_work(item_at(req_k), nb_, b_, y0_, x0_, f0_);
        set_halo_sources(b_, y0_, x0_, f0_);
        return true;
    };

    // ---- weight requests: piece p of this wave's share of a k-step slot is DMA instruction q = wave + 4 p (p < 4), 16 + wave (p = 4;
    // waves 2, 3 repeat their first piece: same bytes to the same place) ----
    const unsigned wlane = (unsigned)lane * 16u;
    auto first_slot_of = [&](int nb) -> unsigned { return (unsigned)nb * (unsigned)ksteps * (unsigned)kWSlot; };
    unsigned w_so = first_slot_of(t_nb);   // scalar offset of the k-step slot being requested
    unsigned w_next = w_so;
    int w_left = ksteps;                   // k-steps of the current item still to request (incl. the one being requested)
    int w_slot = 0;                        // ring slot being requested into
    auto issue_weight_piece = [&](int p) {
        int q = p < 4 ? wave + 4 * p : 16 + wave;
        if (q >= 18) q = wave;
        if (D4_DBG(16)) return;
        dma_buf16(wlane, wr, w_so + (unsigned)(q * 1024), ring_lds + (unsigned)(w_slot * kWSlot + q * 1024));
    };
    auto advance_weight_req = [&]() {
        w_slot = w_slot == kWRing - 1 ? 0 : w_slot + 1;
        --w_left;
        const bool wrap = w_left == 0;
        w_so = wrap ? w_next : w_so + (unsigned)kWSlot;
        w_left = wrap ? ksteps : w_left;
    };
    auto nb_of = [&](int w) -> int {
        if (!tile_major) return w / ntiles;
        const int cg = w / per_group, r = w - cg * per_group;
        return __builtin_amdgcn_readfirstlane(cg * G + r % G);
    };
    auto refresh_w_next = [&](int k) { w_next = first_slot_of(nb_of(item_at(k + 1))); };
    refresh_w_next(0);

    // ---- per-lane LDS read offsets (conv_c4_kernel's image: tile origin at halo pixel (1, 1)) ----
    // input fragment of row-tile r at group g's shift: halo pixel hp = base(r) + shift(g), piece khalf at position
    // khalf ^ ((hp >> 2) & 3); the lo plane is ^ 32, the second halo buffer + kHBuf (an immediate).  The offsets are loop
    // invariants held in registers (16 on flattened levels, 10 on image tiles: see kReuse) -- computed per read they were 7 vector
    // instructions per fragment pair, 112 per k-step against its 108 MFMAs.
    // IMAGE TILES REUSE ROWS: a row-tile is an image row, so the fragment of row-tile r shifted one row up IS row-tile r - 1's
    // unshifted fragment.  Group C (shift (-1, 0)) multiplies group A's fragments of row-tiles 0..2 for its row-tiles 1..3 and reads
    // only the row above the wave's first (into the register set of A's row-tile 3, dead by then); group D does the same with
    // group B's.  38 instead of 50 reads per k-step.  (On flattened levels a row-tile is 32 flat positions, the pitch is not.)
    constexpr bool kReuse = !FLAT;
    auto read_off = [&](int r, int g) -> unsigned {
        const int rt = wave * kRT + r;
        const int base = FLAT ? rt * 32 + (lane & 31) + P + 1 : (rt + 1) * kPitch2D + (lane & 31) + 1;
        const int hp = base - (g >> 1) * P - (g & 1);
        return halo_lds + (unsigned)(hp * 64 + ((khalf ^ ((hp >> 2) & 3)) * 16));
    };
    unsigned aoff[4][kRT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < kRT; ++r)
            if (g < 2 || !kReuse || r == 0) {
                aoff[g][r] = read_off(r, g);
                asm volatile("" : "+v"(aoff[g][r]));  // (a register, not an expression to re-evaluate in the loop)
            }
    // weights: tap slot s, plane p: row n = lane & 31, piece khalf at position khalf ^ ((n >> 3) & 1); s and p are immediates
    const unsigned boff = ring_lds + (unsigned)((lane & 31) * 32 + ((khalf ^ (((lane & 31) >> 3) & 1)) * 16));

    auto lds16 = [](unsigned off) -> f16x8 { return *reinterpret_cast<const lds_f16x8 *>((size_t)off); };
    // register slot of row-tile i's input fragment in group g's set: on image tiles groups C, D find row-tile i's fragment where
    // group A, B left row-tile i - 1's, and their one new row (above row-tile 0) in slot 3
    auto px_slot = [](int g, int i) constexpr -> int { return (kReuse && g >= 2) ? (i + 3) & 3 : i; };
    // part r of group g's operands: the input fragments of row-tile r at the group's shift (image tiles, groups C / D: only r = 0),
    // and the weight fragments of its tap r
    auto load_part = [&](Frags &f, int g, int buf_imm, int slot_, int r) {
        if (D4_DBG(8)) return;
        if (!(kReuse && g >= 2 && r > 0)) {
            const unsigned ao = aoff[g][r];
            f.ah[px_slot(g, r)] = lds16(ao + (unsigned)buf_imm);
            f.al[px_slot(g, r)] = lds16((ao ^ 32u) + (unsigned)buf_imm);
        }
        if (r < d4_ntaps(g)) {
            const unsigned wb = boff + (unsigned)(slot_ * kWSlot + (d4_first(g) + r) * kTapBytes);
            f.wh[r] = lds16(wb);
            f.wl[r] = lds16(wb + 1024u);
        }
    };
    auto load_group = [&](Frags &f, int g, int buf_imm, int slot_) {
#pragma unroll
        for (int r = 0; r < kRT; ++r) load_part(f, g, buf_imm, slot_, r);
    };

    f32x16 acc[kRT][4];  // [row-tile][phase]
    // the 3 NT MFMAs of row-tile i in group g: lo x hi, hi x lo, hi x hi over the group's taps (a different accumulator each).
    // `first` (group A of an item's first k-step): the accumulators start from the MFMA's constant-zero C operand
    auto mma_row = [&](const Frags &f, int i, auto gsel, auto first) {
        constexpr int g = decltype(gsel)::value;
        constexpr bool kFirst = decltype(first)::value;
        constexpr int s0 = d4_first(g), nt = d4_ntaps(g);
        if (D4_DBG(4)) return;
        const int pi = px_slot(g, i);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < nt; ++t) acc[i][d4_phase(s0 + t)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[t], f.al[pi], kFirst ? zero : acc[i][d4_phase(s0 + t)], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < nt; ++t) acc[i][d4_phase(s0 + t)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl[t], f.ah[pi], acc[i][d4_phase(s0 + t)], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < nt; ++t) acc[i][d4_phase(s0 + t)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[t], f.ah[pi], acc[i][d4_phase(s0 + t)], 0, 0, 0);
    };
    // group D has ONE tap: its three products of a row-tile would be back-to-back MFMAs on one accumulator -- run product-major
    // over the four row-tiles instead (a different accumulator every instruction)
    auto mma_d = [&](const Frags &f) {
        if (D4_DBG(4)) return;
#pragma unroll
        for (int i = 0; i < kRT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[0], f.al[px_slot(3, i)], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < kRT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wl[0], f.ah[px_slot(3, i)], acc[i][0], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < kRT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.wh[0], f.ah[px_slot(3, i)], acc[i][0], 0, 0, 0);
    };

    // epilogue parameters: [block slot][bias | scale | shift][32]; channel-major: slot 0 = the current block (reloaded when it
    // changes), tile-major: the G blocks of the current group (reloaded when the group changes)
    auto load_prm = [&](int nb) {
        const int nslots = tile_major ? G : 1;
        for (int i = tid; i < 3 * 32 * nslots; i += kThreads) {
            const int sl = i / 96, j = i - sl * 96;
            const int c = j & 31, which = j >> 5;
            const int n = (tile_major ? (nb / G) * G + sl : nb) * 32 + c;
            prm[i] = which == 0 ? (A.bias ? A.bias[n] : 0.f) : which == 1 ? (A.scale ? A.scale[n] : 1.f) : (A.shift ? A.shift[n] : 0.f);
        }
    };

    // ---- prologue: halo of k-step 0 and the first two pieces of k-step 1's (what group D of a k-step "-1" would have requested),
    // weights of k-steps 0 and 1, epilogue parameters ----
    set_halo_sources(t_b, t_y0, t_x0, t_f0);
#pragma unroll
    for (int i = 0; i < kHSlots; ++i) issue_halo(i, 0);
    advance_halo_req();
#pragma unroll
    for (int i = 0; i < kHaloD; ++i) issue_halo(i, req_u);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int p = 0; p < kWPieces; ++p) issue_weight_piece(p);
        advance_weight_req();
    }
    load_prm(t_nb);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    float amax = 0.f;
    int slot = 0;  // ring slot of the k-step being computed
    int u = 0;     // k-step of the item being computed
    Frags f0, f1;
    using GA = std::integral_constant<int, 0>;
    using GB = std::integral_constant<int, 1>;
    using GC = std::integral_constant<int, 2>;
    using GD = std::integral_constant<int, 3>;

    // One k-step (half-chunk u of the current item; halo buffer u & 1 -- the macro is instantiated per parity so that the buffer
    // is an immediate): groups A B C | counted wait + barrier | D.  CUR / NXT fragment sets alternate per group; group D reads
    // group A's operands of k-step u + 1 (other buffer, next ring slot) unless the item ends.
    // halo source offsets leave the LDS table at the HEAD of the group that issues them (D's: in front of the barrier): read right
    // in front of the request they put an lgkmcnt wait -- which also drains the operand reads queued before it -- into the MFMA stream
#define D4_HV_READ(HV_, FIRST_, N_)                                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < (N_); ++q_) HV_[q_] = hofft[((FIRST_) + q_) * kThreads + tid];
#define D4_HV_ISSUE(HV_, OFF_, FIRST_, N_)                                                                \
    _Pragma("unroll") for (int q_ = 0; q_ < (N_); ++q_) issue_halo_at((FIRST_) + q_, req_u, HV_[(OFF_) + q_]);
#define D4_REQ_HALO(FIRST_, N_)                                                                           \
    {                                                                                                     \
        unsigned hv_[(N_) > 0 ? (N_) : 1];                                                                \
        D4_HV_READ(hv_, FIRST_, N_)                                                                       \
        D4_HV_ISSUE(hv_, 0, FIRST_, N_)                                                                   \
    }
#define D4_REQ_WGT(FIRST_, N_)                                                                            \
    {                                                                                                     \
        _Pragma("unroll") for (int p_ = (FIRST_); p_ < (FIRST_) + (N_); ++p_) issue_weight_piece(p_);     \
    }
#define D4_KSTEP(PAR, FIRST)                                                                              \
    {                                                                                                     \
        constexpr int buf_ = (PAR) * kHBuf, nbuf_ = (1 - (PAR)) * kHBuf;                                  \
        constexpr int kRB_ = kReuse ? 0 : 2;   /* input-fragment reads of parts 1..3 of groups C, D */    \
        const int nslot_ = slot == kWRing - 1 ? 0 : slot + 1;                                             \
        const bool last_ = u + 1 == ksteps;                                                               \
        unsigned hva_[kHaloA + kHaloB > 0 ? kHaloA + kHaloB : 1], hvd_[kHaloD > 0 ? kHaloD : 1];          \
        /* ---- A (48 MFMAs): reads B's operands ---- */                                                   \
        D4_HV_READ(hva_, kHaloD, kHaloA + kHaloB)                                                         \
        _Pragma("unroll") for (int r_ = 0; r_ < kRT; ++r_) {                                              \
            load_part(f1, 1, buf_, slot, r_);                                                             \
            mma_row(f0, r_, GA{}, FIRST{});                                                               \
            if constexpr (kHaloA > 0) if (r_ == 0) D4_HV_ISSUE(hva_, 0, kHaloD, (kHaloA + 1) / 2)         \
            if constexpr (kHaloA > 1) if (r_ == 1) D4_HV_ISSUE(hva_, (kHaloA + 1) / 2, kHaloD + (kHaloA + 1) / 2, kHaloA / 2) \
            if constexpr (kWgtA > 0) if (r_ == 2) D4_REQ_WGT(0, kWgtA)                                    \
        }                                                                                                 \
        d4_sched_rows<12, 4, 4, 2, 2>();                                                                  \
        /* ---- B (24): reads C's operands ---- */                                                         \
        _Pragma("unroll") for (int r_ = 0; r_ < kRT; ++r_) {                                              \
            load_part(f0, 2, buf_, slot, r_);                                                             \
            mma_row(f1, r_, GB{}, std::false_type{});                                                     \
            if constexpr (kHaloB > 0) if (r_ == 1) D4_HV_ISSUE(hva_, kHaloA, kHaloD + kHaloA, kHaloB)     \
            if constexpr (kWgtB > 0) if (r_ == 2) D4_REQ_WGT(kWgtA, kWgtB)                                \
        }                                                                                                 \
        d4_sched_rows<6, 4, 2 + kRB_, kRB_, kRB_>();                                                      \
        /* ---- C (24): reads D's operands (and D's halo source offsets) ---- */                           \
        D4_HV_READ(hvd_, 0, kHaloD)                                                                       \
        _Pragma("unroll") for (int r_ = 0; r_ < kRT; ++r_) {                                              \
            load_part(f1, 3, buf_, slot, r_);                                                             \
            mma_row(f0, r_, GC{}, std::false_type{});                                                     \
            if constexpr (kWgtC > 0) if (r_ == 1) D4_REQ_WGT(kWgtA + kWgtB, kWgtC)                        \
        }                                                                                                 \
        d4_sched_rows<6, 4, kRB_, kRB_, kRB_>();                                                          \
        /* every halo piece of the next k-step and every weight piece of earlier k-steps this wave requested has landed; */ \
        /* every wave has issued its last operand reads of this k-step */                                 \
        /* (lgkmcnt(0): group D's operand reads, issued during C, have RETURNED before any wave may request into the buffer they read) */ \
        if (!D4_DBG(128)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kWaitN) : "memory");          \
        if (!D4_DBG(64)) asm volatile("s_barrier" ::: "memory");                                           \
        /* ---- D (12): reads A's operands of the next k-step; the first halo pieces of k-step u + 2 (the buffer is free now) ---- */ \
        if (!last_) {                                                                                     \
            if (advance_halo_req()) D4_HV_READ(hvd_, 0, kHaloD)   /* (a new item: the table has just been rewritten) */ \
            load_group(f0, 0, nbuf_, nslot_);                                                             \
        }                                                                                                 \
        mma_d(f1);                                                                                        \
        if (!last_) D4_HV_ISSUE(hvd_, 0, 0, kHaloD)                                                       \
        if constexpr (kWgtD > 0) D4_REQ_WGT(kWgtA + kWgtB + kWgtC, kWgtD)                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) {                                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                            \
        }                                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                \
        advance_weight_req();                                                                             \
        slot = nslot_;                                                                                    \
    }
    // all k-steps of the current item (an even number: two per 32-channel chunk)
#define D4_ITEM()                                                                                         \
    D4_KSTEP(0, std::true_type)                                                                           \
    ++u;                                                                                                  \
    D4_KSTEP(1, std::false_type)                                                                          \
    ++u;                                                                                                  \
    while (u < ksteps) {                                                                                  \
        D4_KSTEP(0, std::false_type)                                                                      \
        ++u;                                                                                              \
        D4_KSTEP(1, std::false_type)                                                                      \
        ++u;                                                                                              \
    }

    // ======================= epilogue of a work item =======================
    // staging: buffer 1 (the k-step that just finished; buffer 0 holds the next item's first), 8 KB per wave
    auto epilogue = [&]() {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));  // (opaque: see conv_c4_kernel -- per-lane addresses computed before the main loop get spilled)
        const int lane = lane_e, khalf = lane_e >> 5;
        int ey0 = t_y0, ex0 = t_x0, eb = t_b, ef0 = t_f0, enb = t_nb;
        asm volatile("" : "+s"(ey0), "+s"(ex0), "+s"(eb), "+s"(ef0), "+s"(enb));
        const int t_y0 = ey0, t_x0 = ex0, t_b = eb, t_f0 = ef0, t_nb = enb;
        unsigned char *stage = halo + kHBuf + wave * kStage;
        if (FLAT) {
            // input position -> output pixel of phase (0, 0): (b, 2 y, 2 x) of the 2H x 2W grid
            for (int i = tid; i < kMTile; i += kThreads) {
                int pix = flat_to_pix(t_f0 + i, P, A.H, A.W, A.B);
                if (pix >= 0) {
                    const int x = pix % A.W, r = pix / A.W;  // r = b * H + y
                    pix = (2 * r) * (2 * A.W) + 2 * x;
                }
                outpix[i] = pix;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const float winv = A.winv;
        const float floor_ = A.relu ? 0.f : -__builtin_inff();
        const int cbase = t_nb * 32;
        const int px_l = lane & 31;
        auto piece_off = [&](int px, int g) -> int { return px * 128 + ((g ^ (((px & 1) << 2) | ((px >> 1) & 3))) * 16); };
        struct Prm8 {
            f32x4 b[2], s[2], t[2];
        };
        auto finish8 = [&](const f32x4 &r0, const f32x4 &r1, const Prm8 &p, float (&v)[8]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = fmaf(fmaxf(fmaf(r0[r], winv, p.b[0][r]), floor_), p.s[0][r], p.t[0][r]);
                v[4 + r] = fmaf(fmaxf(fmaf(r1[r], winv, p.b[1][r]), floor_), p.s[1][r], p.t[1][r]);
            }
        };
        auto split8 = [&](const float (&v)[8], uint4 &hi, uint4 &lo) {
            uint2 h0, l0, h1, l1;
            const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
            split4(a, h0, l0);
            split4(b, h1, l1);
            hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
            lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
        };
        // pairs of row-tiles of one phase form one software pipeline over (phase, pair): 4 x 2 = 8 pairs
        constexpr int kNP = kRT / 2, kU = 4 * kNP;
        auto put_tile = [&](int i, int ph, int win) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x16 &a = acc[i][ph];
                const f32x4 piece = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                *reinterpret_cast<f32x4 *>(stage + win * 4096 + piece_off(px_l, 2 * q + khalf)) = piece;
            }
        };
        auto put_pair = [&](int w) {
            put_tile(2 * (w % kNP), w / kNP, 0);
            put_tile(2 * (w % kNP) + 1, w / kNP, 1);
        };
        put_pair(0);
        Prm8 p;
        {
            const float *pp = prm + (tile_major ? (t_nb % G) * 96 : 0) + 8 * (lane & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                p.b[h] = *reinterpret_cast<const f32x4 *>(pp + 4 * h);
                p.s[h] = *reinterpret_cast<const f32x4 *>(pp + 32 + 4 * h);
                p.t[h] = *reinterpret_cast<const f32x4 *>(pp + 64 + 4 * h);
            }
        }
        const unsigned pstep = (unsigned)A.ldy * 4u;  // bytes per output pixel
#pragma unroll
        for (int w = 0; w < kU; ++w) {
            const int ph = w / kNP, pr = w % kNP;
            const int py = ph >> 1, pxo = ph & 1;
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));  // (addresses rebuilt per pair from a lane index made opaque HERE: as loop invariants they get spilled)
            const int oc_t = lane_t & 3, opx_t = lane_t >> 2;
            f32x4 r[2][2][2];
#pragma unroll
            for (int win = 0; win < 2; ++win) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) r[win][t][h] = *reinterpret_cast<const f32x4 *>(stage + win * 4096 + piece_off(t * 16 + opx_t, 2 * oc_t + h));
                }
            }
            if (w + 1 < kU) put_pair(w + 1);
#pragma unroll
            for (int win = 0; win < 2; ++win) {
                const int i = 2 * pr + win;
                uint4 hi[2], lo[2];
                float tmax[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v[8];
                    finish8(r[win][t][0], r[win][t][1], p, v);
                    float tm = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; k += 2) tm = fmaxf(fmaxf(tm, fabsf(v[k])), fabsf(v[k + 1]));
                    tmax[t] = tm;
                    split8(v, hi[t], lo[t]);
                }
                if (FLAT) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int pix = outpix[(wave * kRT + i) * 32 + t * 16 + opx_t];
                        amax = fmaxf(amax, pix >= 0 ? tmax[t] : 0.f);  // (saturation tracking over the STORED values only)
                        asm volatile("" : "+v"(amax));
                        if (pix >= 0 && !D4_DBG(32)) {
                            unsigned char *dst = static_cast<unsigned char *>(A.y) +
                                                 ((long long)(pix + py * 2 * A.W + pxo) * A.ldy + A.yoff + cbase) * 4 + oc_t * 16;
                            nt_store16(dst, hi[t]);
                            nt_store16(dst + 64, lo[t]);
                        }
                    }
                } else {
                    const int yy = t_y0 + wave * kRT + i;  // input row (scalar)
                    if (yy < A.H) {
                        unsigned char *rowp = static_cast<unsigned char *>(A.y) +
                                              (((long long)(2 * (t_b * A.H + yy) + py) * (2 * A.W) + 2 * t_x0 + pxo) * A.ldy + A.yoff + cbase) * 4;  // scalar pointer
                        const int xlim = A.W - t_x0;  // (scalar) input positions of this tile inside the image: < 32 only in the last column tile of a level with W % 32 != 0
                        if (xlim >= 32) {
                            amax = fmaxf(fmaxf(amax, tmax[0]), tmax[1]);
                            asm volatile("" : "+v"(amax));
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const unsigned off = (unsigned)(2 * (t * 16 + opx_t)) * pstep + (unsigned)oc_t * 16u;
                                if (!D4_DBG(32)) {
                                    nt_store16(rowp + (size_t)off, hi[t]);
                                    nt_store16(rowp + (size_t)(off + 64u), lo[t]);
                                }
                            }
                        } else {  // ragged tile: positions at or beyond W are neither stored nor tracked
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const bool in = t * 16 + opx_t < xlim;
                                amax = fmaxf(amax, in ? tmax[t] : 0.f);
                                asm volatile("" : "+v"(amax));
                                const unsigned off = (unsigned)(2 * (t * 16 + opx_t)) * pstep + (unsigned)oc_t * 16u;
                                if (in && !D4_DBG(32)) {
                                    nt_store16(rowp + (size_t)off, hi[t]);
                                    nt_store16(rowp + (size_t)(off + 64u), lo[t]);
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one pair at a time (left alone, the scheduler piles up window traffic and the addresses spill)
        }
    };
    // between two items: geometry and epilogue parameters of the next one; everyone done with the staging windows (halo buffer 1),
    // then the two halo pieces of its k-step 1 that the previous item's last group D left out.  Returns false after the last item.
    auto next_item = [&]() -> bool {
        if (++cur >= my_items) return false;
        const int prev_nb = t_nb;
        decode_work(item_at(cur), t_nb, t_b, t_y0, t_x0, t_f0);
        refresh_w_next(cur);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tile_major ? t_nb / G != prev_nb / G : t_nb != prev_nb) {
            load_prm(t_nb);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        advance_halo_req();
        D4_REQ_HALO(0, kHaloD)
        return true;
    };
    // (output-only: the fragment sets are dead across the epilogue, but their conditional refill keeps the old contents live to
    //  the compiler -- 128 registers the epilogue does not have; see conv_c4_kernel)
    auto kill_frags = [&](Frags &f) {
#pragma unroll
        for (int i = 0; i < kRT; ++i) {
            asm volatile("" : "=v"(f.ah[i]));
            asm volatile("" : "=v"(f.al[i]));
            asm volatile("" : "=v"(f.wh[i]));
            asm volatile("" : "=v"(f.wl[i]));
        }
    };

    while (true) {
        load_group(f0, 0, 0, slot);  // group A's operands of the item's first k-step (landed: the previous item's last waits / the prologue)
        u = 0;
        D4_ITEM()
        kill_frags(f0);
        kill_frags(f1);
        if (!D4_DBG(1)) epilogue();
        if (!next_item()) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this block's DMA may land after it has exited
    if (A.sat && amax > 65504.f) *A.sat = 1;
}

template <bool FLAT>
static constexpr size_t d4_lds_bytes() {
    return (size_t)2 * kHBuf + (size_t)kWRing * kWSlot + (FLAT ? kMTile * 4 : 0) + (size_t)3 * 32 * 4 * kPrmBlocks + (size_t)kHSlots * kThreads * 4;
}
static_assert(d4_lds_bytes<true>() <= 160 * 1024, "LDS");
static_assert(kWaves * kStage <= kHBuf, "the staging windows live in halo buffer 1");

// which layers the kernel takes: a transposed convolution with 32-channel input chunks and output blocks, on a level it tiles
bool conv_d4_supported(const ConvS3Args &k) {
    if (!k.deconv || k.one || k.head_w || k.pool_y) return false;
    if (k.Cin % 32 || k.Cout % 32) return false;
    const unsigned long long img = (unsigned long long)k.H * k.W * (unsigned long long)k.ldx * 4ull;  // (see conv_c4_supported)
    if (!conv_tiles_flat(k.W)) return img < (unsigned long long)kPadOff;  // image tiles (W % 32 != 0: a ragged last column tile)
    const unsigned long long span = (624ull + (unsigned long long)(k.H + 1) * (k.W + 2) - 1) / ((unsigned long long)(k.H + 1) * (k.W + 2)) + 1;
    return span * img < (unsigned long long)kPadOff;
}

template <bool FLAT>
static hipError_t d4_launch_t(ConvS3Args &k, int num_cu, hipStream_t stream) {
    auto fn = deconv_d4_kernel<FLAT>;
    constexpr size_t lds = d4_lds_bytes<FLAT>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = k.nwork < num_cu ? k.nwork : num_cu;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, stream, k);
    return hipGetLastError();
}

hipError_t conv_d4_launch(const ConvS3Args &k0, int num_cu, hipStream_t stream) {
    ConvS3Args k = k0;
    if (!conv_d4_supported(k) || !k.w_c4) return hipErrorInvalidValue;
    const bool flat = conv_tiles_flat(k.W);
    k.chunks = k.Cin / 32;
    k.steps = 2 * k.chunks;
    k.nb = k.Cout / 32;
    if (flat) {
        k.P = k.W + 2;
        const long long span = (long long)k.B * (k.H + 1) * k.P - k.P;
        k.ntiles = (int)((span + kMTile - 1) / kMTile);
        k.tiles_x = k.tiles_y = 0;
    } else {
        k.P = kPitch2D;
        k.tiles_x = (k.W + 31) / 32;
        k.tiles_y = (k.H + kRows - 1) / kRows;
        k.ntiles = k.B * k.tiles_x * k.tiles_y;
    }
    k.nj = 0;
    k.nwork = k.nb * k.ntiles;
    // item order (see the kernel): tile-major where the layer's weight image stays in an XCD's L2 beside the activations -- a
    // property of the layer (QMRI_D4_ORDER = 0: channel-major everywhere, 1: the largest group whatever the weights' size; the A/B switch)
    static const int order = [] {
        const char *e = std::getenv("QMRI_D4_ORDER");
        return e ? std::atoi(e) : -1;
    }();
    const size_t block_bytes = (size_t)k.Cin * 9 * 32 * 4;  // weights of one 32-channel block
    int G = 0;
    for (int g = 2; g <= kPrmBlocks && g <= k.nb; ++g)
        if (k.nb % g == 0 && (order > 0 || g * block_bytes <= (size_t)2400 << 10)) G = g;
    k.tile_group = order == 0 ? 0 : G;
    static const int dbg = [] {
        const char *e = std::getenv("QMRI_D4_DBG");
        return e ? std::atoi(e) : 0;
    }();
    k.dbg = dbg;
    (void)hipGetLastError();
    return flat ? d4_launch_t<true>(k, num_cu, stream) : d4_launch_t<false>(k, num_cu, stream);
}

}  // namespace qmri
