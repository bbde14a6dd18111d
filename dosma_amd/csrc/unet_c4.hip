// unet_c4.hip -- the parity-mode ("fp16x3") 3x3 convolution for layers with >= 64 output channels, round 4:
// ONE WAVE PER SIMD, 128 x 128 (or 192 x 64) register tiles.
//
// Same layer as conv_s3_kernel<128> (unet_s3.hip): Conv2D(3x3, SAME) + bias + ReLU (+ the BatchNormalization affine after
// the block's second ReLU, + fused MaxPooling2D) of /root/reference/dosma/models/oaiunet2d.py:213-226, 266-279 on SPLIT
// activations (fp16 hi + lo parts, a product = hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate).
//
// Why another kernel.  conv_s3_kernel<128> runs 8 waves (two per SIMD) of 2 x 2 MFMA tiles on a 256-pixel x 128-channel
// block tile and issues 1.10-1.22 PFLOP/s (MfmaUtil 0.68-0.70).  scripts/probes/mfma_lds.hip (profiles/r04_mfma_lds.txt) rebuilds
// that step as a skeleton with its LDS reads, its three LDS-DMA requests and its bookkeeping instructions: 1.34 PF of the 1.64 PF
// the part sustains on MFMAs alone -- and 1.47 PF for FOUR waves (one per SIMD) of 4 x 4 tiles stepping K by 16.  What the bigger
// tile buys is not LDS bandwidth (ds_read_b128 runs at 256 B/clk; the reads are a third of the pipe) but fewer DMA requests,
// reads, barriers and bookkeeping instructions per MFMA: per k-step of 16 a wave issues 48 MFMAs against 16 reads and 2-4
// requests (conv_s3_kernel<128>: 12 MFMAs against 8 reads and 1.5 requests).
//
// Structure
//   * block = 4 waves, __launch_bounds__(256, 1): up to 512 registers per lane -- the 256 accumulators of a wave's
//     128-pixel x 128-channel tile (AccVGPRs) + two sets of operand fragments (128) + addressing.
//   * block tile = 512 output positions x 128 channels: 16 image rows x 32 pixels (halo 18 x 34 = 1.20 x the tile; the 8-row
//     tile of conv_s3_kernel reads 1.33 x), or 512 consecutive positions of the flattened zero-framed image stack for levels
//     narrower than 64 pixels (same construction as unet_s3.hip).  Wave w owns rows 4 w .. 4 w + 3 (positions 128 w ..).
//   * K runs in steps of ONE k-step: (32-channel chunk, 16-channel half, tap).  The halo of a half-chunk is 64 bytes per pixel
//     (16 hi + 16 lo parts), 39 KB per buffer, two buffers -- the 77 KB halo of a whole chunk would not fit twice beside the
//     weight ring.  The source layout stays the network's (128-byte pixel-chunks): a half-chunk request fetches two 32-byte
//     runs per pixel.  LDS image: pixel p at p * 64, piece c = plane * 2 + kgroup at position c ^ ((p >> 2) & 3): every
//     ds_read_b128 lane group is conflict-free at any tap shift (brute-forced: scripts/lds_bank_check.py).
//   * weights: ring of 8 slots of 8 KB ([plane][128 channels][2 x 16 B], piece g of row n at position g ^ ((n >> 3) & 1)),
//     requested SIX steps ahead; halo pieces of the next half-chunk are requested during taps 0-3.  All by LDS-DMA through
//     buffer_load_dwordx4 ... lds: the address is an SGPR descriptor + a per-lane 32-bit offset + a scalar offset, so a
//     request costs no vector ALU, and a lane outside the image reads beyond num_records = ZERO (the SAME padding; no zero
//     line, no select; scripts/probes/buffer_lds.hip).  Waits are counted (s_waitcnt vmcnt(N): everything issued two or
//     more steps ago has landed), one s_barrier per TWO steps = 96 MFMAs per wave (the ring protocol and round 6's item-start hole: see kBarEvery).
//   * the operand fragments of step s + 1 are read while step s multiplies (two register sets); the second halo buffer is an
//     immediate offset; the first k-step of an item starts from the MFMA's constant-zero C operand (no zeroing pass).
//   * work items are big, so the last, partial round of a launch is split by CHANNELS: each leftover item becomes four
//     sub-items of one 32-channel column tile, one per block (the 24 x 24 level: 3.3 instead of 4 rounds).
//   * epilogue: THE ACCUMULATORS LEAVE THE REGISTER FILE RAW -- ds_write_b128 straight from AccVGPRs into two wave-private
//     4 KB windows in the finished half-chunk's halo buffer -- and come back pixel-major, a lane = 8 channels of one pixel:
//     bias, ReLU, BatchNormalization affine, split, two 16-byte stores (hi and lo plane) into the pixel's 128-byte record;
//     fused 2x2 max-pool on the staged row pair; saturation tracking.  (The first version did the arithmetic in the MFMA
//     layout like unet_s3.hip: VALU instructions cannot read AccVGPRs, the register allocator moved the epilogue's whole live
//     range into ArchVGPRs at its entry -- 160 v_accvgpr_read in a row and six tiles to scratch, every reload behind
//     s_waitcnt vmcnt(0), i.e. behind the acknowledgement of all stores so far.  -3.7 % on the layers of this kernel.)
//   * 64-channel layers (CT = 2) on image tiles take 6 x 2 register tiles per wave: 24 rows x 32 pixels per block, 36 MFMAs
//     per k-step, halo 1.15 x the tile (C4Geo).  With 4 x 2 tiles (24 MFMAs per k-step; still the flattened form) the kernel
//     lost to conv_s3_kernel<64>.
//   * which layers run here is conv_s3_takes_c4's decision (unet_s3.hip): by layer shape only, so that a slice's bits do not
//     depend on the batch.
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
// Tile geometry.  A wave's register tile is RT row-tiles (32 pixels each) x CT column tiles (32 channels each; the template parameter):
//   CT = 4: 128-channel blocks, RT = 4 -> 256 accumulators; block tile 16 rows x 32 pixels (or 512 flattened positions)
//   CT = 2: 64-channel blocks (the Cout = 64 layers).  With RT = 4 a k-step is 24 MFMAs per wave and the kernel loses to
//           conv_s3_kernel<64>; on image tiles it therefore takes RT = 6: 24 rows x 32 pixels, 192 accumulators, 36 MFMAs per
//           k-step, halo 26 x 34 = 1.15 x the tile -- what the LDS holds twice beside the ring (2 x 56 KB + 32 KB).
template <bool FLAT, int CT>
struct C4Geo {
    static constexpr int RT = (CT == 2 && !FLAT) ? 6 : 4;   // 32-pixel row-tiles per wave
    static constexpr int ROWS = 4 * RT;                      // image rows of a block tile (one per row-tile)
    static constexpr int MTILE = ROWS * 32;                  // output positions per block tile
    static constexpr int HALO2D = (ROWS + 2) * kPitch2D;     // 612 / 884 halo pixels
    // DMA instructions (16 pixels x 64 B) per halo buffer: 39 (624 >= 612 (2D), >= 512 + 2 * 50 + 2 (flat)) / 56 (896 >= 884)
    static constexpr int NJ = (HALO2D + 15) / 16;
    static constexpr int HBUF = NJ * 1024;                   // bytes per halo buffer
    static constexpr int HSLOTS = (NJ + 3) / 4;              // halo pieces per wave: 10 (the 40th repeats the wave's first) / 14
    static constexpr int STAGE = 8192;                       // epilogue staging per wave: two 4 KB windows (the tiles of a row pair)
};
constexpr int kPrmBlocksC4 = 2;          // channel blocks whose epilogue parameters the LDS holds at once (tile-major item order: two-block layers)
constexpr int kRing = 8;                 // weight ring slots of [2 planes][32 CT rows][32 B] = 2048 CT bytes
// One s_barrier per kBarEvery steps.  What the ring allows: the operands of step u are read during step u - 1, i.e. after the last
// barrier at or before step u - 2; a wave's counted wait in front of a barrier covers everything it issued kInFlight or more steps
// earlier, so slots must be requested kAhead >= kBarEvery + kInFlight + 1 steps ahead, and a slot is only rewritten after a barrier
// that follows its last read: kRing >= kAhead + kBarEvery - 1.  Measured on the nine c4 layers of the 384 x 384 network, same box,
// alternating (profiles/r04_c4_ab.txt): every step 10.84 ms, every 2nd 10.72, every 3rd (taps 2, 5, 8 of a half-chunk) 10.68.
// ROUND 6: THE DEFAULT IS EVERY SECOND STEP -- the every-third cadence of rounds 4-5 had a hole AT THE START OF A WORK ITEM.  The ring
// inequality above assumes that the operands of step u are read during step u - 1.  That is false for step 0 of every item: the last
// step of the previous item reads nothing ahead (the epilogue needs the registers), so step 0's operands are read by load_frags behind
// next_item's barrier -- in the SAME barrier interval as steps 0, 1, 2 when the first barrier of the item stands at tap 2.  The request of
// step 2 (the weights of step 8) lands in ring slot (2 + kAhead) mod kRing = step 0's slot: a wave that runs two steps ahead rewrites it while
// a late wave has not read step 0's weight fragments yet, and the late wave multiplies step 0's pixels by step 8's weights for the column
// tile(s) it had not read -- one wave's 4 rows x 32 pixels x 32 or 64 channels of one item off by one k-step's contribution.  Measured:
// 2-12 % of the forwards of the 512 x 512 network (its up3.conv1 first in 18 of 20), 0.07 % at 384 x 384; the damaged unit attributed by
// scripts/c4_keep_model.py to "k-step 0 multiplied by the weights of step 8" with correlation 0.996; and ONE bare s_barrier behind step 1 of
// every item (-DQMRI_C4_BAR3F) removes it (0 of 1 500 forwards against 184 of 1 500 on the same box), as does any cadence whose first barrier
// stands at tap <= kRing - kAhead - 1 = 1: every second step 0 of 5 600, every step 0 of 1 000 (profiles/r06_c4_race.txt).  The
// condition is a static_assert now.  Every second step costs 0.3 % against every third on the forward and nothing against third + fence.
#if defined(QMRI_C4_BAR1)                // (A/B switches)
constexpr int kBarEvery = 1, kInFlight = 3, kAhead = 5;
#elif defined(QMRI_C4_BAR3) || defined(QMRI_C4_BAR3F)  // (rounds 4-5's default: see above; BAR3F = with the item fence)
constexpr int kBarEvery = 3, kInFlight = 2, kAhead = 6;
#else
constexpr int kBarEvery = 2, kInFlight = 3, kAhead = 6;  // weights are requested kAhead steps ahead
#endif
static_assert(kAhead >= kBarEvery + kInFlight + 1 && 8 >= kAhead + kBarEvery - 1, "ring protocol");
// item start: step 0's operands are read behind next_item's barrier; the first request into step 0's slot is issued at step 8 - kAhead, so
// a barrier must stand behind one of the steps 0 .. 7 - kAhead (the first barrier of an item: tap kBarEvery - 1; BAR3F adds its own)
#if !defined(QMRI_C4_BAR3)
static_assert(kBarEvery - 1 <= 8 - kAhead - 1
#if defined(QMRI_C4_BAR3F)
              || true
#endif
              , "the first barrier of a work item must separate load_frags from the request that rewrites step 0's ring slot");
#endif
// requests a wave issues in tap t: the step's weight pieces + the halo pieces of c4_halo_pieces(t, n) (n = pieces per wave and
// half-chunk, spread over the first kHaloTaps taps: 3 3 2 2 for 10, 4 4 3 3 for 14); the counted wait at the end of tap t lets the
// requests of the last kInFlight steps (taps wrap: every half-chunk has the same pattern) stay in flight
constexpr int kHaloTaps = kBarEvery == 1 ? 5 : 4;
constexpr int c4_halo_pieces(int t, int n) { return t >= kHaloTaps ? 0 : n / kHaloTaps + (t < n % kHaloTaps ? 1 : 0); }
constexpr int c4_halo_first(int t, int n) { return t * (n / kHaloTaps) + (t < n % kHaloTaps ? t : n % kHaloTaps); }
constexpr int c4_issued(int t, int w, int n) { return w + c4_halo_pieces(t, n); }
constexpr int c4_in_flight(int t, int w, int n) {
    return c4_issued(t, w, n) + c4_issued((t + 8) % 9, w, n) + (kInFlight == 2 ? 0 : c4_issued((t + 7) % 9, w, n));
}
static_assert(c4_halo_first(kHaloTaps - 1, 10) + c4_halo_pieces(kHaloTaps - 1, 10) == 10 && c4_halo_pieces(0, 10) <= 4, "halo schedule");
static_assert(c4_halo_first(kHaloTaps - 1, 14) + c4_halo_pieces(kHaloTaps - 1, 14) == 14 && c4_halo_pieces(0, 14) <= 4, "halo schedule");

template <int RT, int CT>
struct Frags {
    f16x8 ah[RT], al[RT], bh[CT], bl[CT];
};

// scheduling hints of one k-step (sched_group_barrier wants literal counts): group g = the 3 * CT MFMAs of row-tile g with the
// NR reads of the next step's part g (its two pixel fragments, + two weight fragments while g < CT) one per MFMA behind the first
template <int G, int RT, int CT>
__device__ __forceinline__ void c4_sched_step() {
    if constexpr (G < RT) {
        constexpr int NM = 3 * CT, NR = G < CT ? 4 : 2, NP = NR < NM ? NR : NM;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (NR > NP) __builtin_amdgcn_sched_group_barrier(0x100, NR - NP, 0);
        if constexpr (NM > NP) __builtin_amdgcn_sched_group_barrier(0x008, NM - NP, 0);
        c4_sched_step<G + 1, RT, CT>();
    }
}

}  // namespace

#ifdef QMRI_C4_EXPERIMENTS  // timing experiments (results wrong by construction): QMRI_C4_DBG = 1 no epilogue | 2 halo sources computed once | 4 no MFMAs | 8 no LDS reads | 16 no DMA requests | 32 no global stores in the epilogue | 64 odd blocks start half an item late
#define C4_DBG(bit) (A.dbg & (bit))
#else
#define C4_DBG(bit) 0
#endif
#ifdef QMRI_C4_TIMELINE  // (experiment build: block 8 records where its second and third work items spend their time, in 10 ns ticks;
                         //  read back with qmri_debug_c4_timeline -- scripts/c4_timeline.py)
__device__ unsigned long long g_c4_tl[256 * 16];
__device__ unsigned int g_c4_tl_n;
#define C4_TS(k)                                                                                     \
    do {                                                                                             \
        if (blockIdx.x == 8 && tid == 0 && cur >= 1 && cur <= 2) tsbuf[(cur - 1) * 8 + (k)] = wall_clock64(); \
    } while (0)
#if QMRI_C4_TIMELINE >= 2  // (timestamps inside the first barrier interval too: they disturb the steps they sit in)
#define C4_TS_STEP(k) C4_TS(k)
#else
#define C4_TS_STEP(k)
#endif
#else
#define C4_TS(k)
#define C4_TS_STEP(k)
#endif

template <bool FLAT, int CT>
__global__ __launch_bounds__(kThreads, 1) void conv_c4_kernel(const ConvS3Args A) {
    using G = C4Geo<FLAT, CT>;
    constexpr int kRT = G::RT, kMTile = G::MTILE, kHalo2D = G::HALO2D, kNJ = G::NJ, kHBuf = G::HBUF, kHSlots = G::HSLOTS;
    constexpr int kCT = CT, kBN = 32 * CT, kSlot = 2048 * CT;
    constexpr int kWPieces = kSlot / (kWaves * 1024);  // weight DMA instructions per wave and step: 2 (CT = 4) or 1 (CT = 2)
    using Frags = Frags<kRT, CT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *halo = smem;                                        // [2][kNJ * 16 pixels][64 B swizzled]
    unsigned char *ring = smem + 2 * kHBuf;                            // [kRing][2 planes][128][32 B swizzled]
    int *outpix = reinterpret_cast<int *>(ring + kRing * kSlot);       // [kMTile] output pixel of a tile position, or -1 (FLAT only)
    float *prm = reinterpret_cast<float *>(outpix + (FLAT ? kMTile : 0));  // bias | scale | shift, [128] each
    unsigned *hofft = reinterpret_cast<unsigned *>(prm + 3 * kBN * kPrmBlocksC4);  // [kHSlots][256] per-lane halo source offsets (see below)
#ifdef QMRI_C4_TIMELINE
    unsigned long long *tsbuf = reinterpret_cast<unsigned long long *>(hofft + kHSlots * kThreads);
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5;
    const int P = FLAT ? A.P : kPitch2D;
    const int hpix = FLAT ? kMTile + 2 * P + 2 : kHalo2D;
    const int ntiles = A.ntiles;
    const bool tile_major = A.tile_group != 0;  // (groups of TWO channel blocks: see decode_work)
    const int per_group = 2 * ntiles;
    const int wsteps = A.steps;  // chunks * 18 steps per work item

    // the activation descriptor's base is the first IMAGE the requested tile's halo touches (set_halo_sources): per-lane offsets
    // are 32 bits and stay inside a few images' bytes, so which layers this kernel can address does not depend on the batch
    i32x4 xr = make_rsrc(A.x);
    const i32x4 wr = make_rsrc(A.w_c4);
    const long long img_bytes = (long long)A.H * A.W * A.ldx * 4;
    const unsigned halo_lds = lds_off(halo), ring_lds = lds_off(ring);

    // ---- work distribution ----
    // XCD x (= blockIdx % 8, observed; speed only) walks a contiguous eighth of the work items, its blocks striding through it
    // (neighbouring tiles and one channel block's weights stay in one L2) -- as conv_s3_kernel.  New here: THE LAST, PARTIAL
    // ROUND IS SPLIT BY CHANNELS.  Items are big (512 positions x 128 channels) and a range of n items over nblk blocks takes
    // ceil(n / nblk) rounds: the 24 x 24 level (816 items on 256 CUs) would run 4 rounds for 3.19 rounds of work.  When the
    // leftover L = n mod nblk satisfies CT * L <= nblk, each leftover item becomes CT sub-items of one 32-channel column tile
    // (same tile, same halo, same weight slots -- only the column tile differs), one per block: a quarter-length last round.
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
    const int r_full = n_range / nblk, l_over = n_range - r_full * nblk;
    const bool split = A.c4_split && l_over > 0 && kCT * l_over <= nblk;
    const int my_full = r_full + ((!split && lb < l_over) ? 1 : 0);       // whole items of this block
    const bool has_sub = split && lb < kCT * l_over;                      // + one sub-item (its last)
    const int my_items = my_full + (has_sub ? 1 : 0);
    if (my_items == 0) return;
    const int sub_cq = lb % kCT;                                          // the sub-item's column tile
    // k-th item of this block -> work item index (a sub-item is its parent; past the end: the last one again -- requests made
    // for it are never read)
    auto item_at = [&](int k) -> int {
        if (k >= my_items) k = my_items - 1;
        return k < my_full ? lo + lb + k * nblk : lo + r_full * nblk + lb / kCT;
    };

    int t_nb = 0, t_b = 0, t_y0 = 0, t_x0 = 0, t_f0 = 0;
    auto decode_work = [&](int w, int &nb, int &b, int &y0, int &x0, int &f0) {
        // channel-major (an XCD's blocks share one channel block's weights; every block re-reads the tiles) or -- where the launcher
        // found two channel blocks' weights small enough to stay in an XCD's L2 beside the activations (A.tile_group) -- TILE-MAJOR in
        // groups of two channel blocks: the pair runs side by side on one XCD and the tile's halo comes from HBM once per pair
        int t;
        if (tile_major) {  // w = (group * ntiles + tile) * 2 + j, channel block = 2 group + j (two-block layers: one group)
            const int cg = w / per_group, r = w - cg * per_group;
            t = r >> 1;
            nb = 2 * cg + (r & 1);
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
            y0 = ty * G::ROWS;
            x0 = (r - ty * A.tiles_x) * 32;
            f0 = 0;
        }
    };
    int cur = 0;  // ordinal of the item being computed
    decode_work(item_at(0), t_nb, t_b, t_y0, t_x0, t_f0);

    // ---- halo requests: piece i of this wave is DMA instruction j = wave + 4 i (16 pixels x 64 B) of a halo buffer ----
    // lane -> halo pixel hp = 16 j + (lane >> 2), LDS position pos = lane & 3 holds piece c = pos ^ ((hp >> 2) & 3) = plane * 2 + g:
    // source bytes [plane * 64 + (2 half + g) * 16, + 16) of the pixel's 128-byte chunk record (half and chunk go into soffset)
    // hofft[i][tid]: per-lane source offset of piece i for the tile being REQUESTED (chunk 0, half 0), or kPadOff.  A table in LDS,
    // not ten registers: each value is used once per half-chunk, the register allocator spills such values to scratch, and a
    // scratch reload is a vector-memory load -- hipcc waits for it with vmcnt(0), i.e. for every DMA request in flight.
    auto set_halo_sources = [&](int b, int y0, int x0, int f0) {
        int lane_h = lane;
        asm volatile("" : "+v"(lane_h));  // (opaque: the per-piece pixel coordinates are otherwise computed before the main loop and spilled)
        const int lane = lane_h;
        // first image of the halo: the tile's own (image tiles) / the one holding the first halo position of the stack (flattened)
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
            if (hp < hpix) {
                if (FLAT) {
                    pix = flat_to_pix(f0 - P - 1 + hp, P, A.H, A.W, A.B);
                } else {
                    const int hy = hp / kPitch2D, hx = hp - hy * kPitch2D;
                    const int yy = y0 + hy - 1, xx = x0 + hx - 1;
                    if ((unsigned)yy < (unsigned)A.H && (unsigned)xx < (unsigned)A.W) pix = (b * A.H + yy) * A.W + xx;
                }
            }
            hofft[i * kThreads + tid] = pix >= 0 ? (unsigned)(pix - pix0) * (unsigned)A.ldx * 4u + srcb : kPadOff;
        }
    };
    // the half-chunk being requested: item ordinal, half-chunk index hc = 2 chunk + half, destination buffer = hc & 1
    int req_k = 0, req_hc = 0;
    const int nhc = 2 * A.chunks;
    auto halo_soff = [&](int hc) -> unsigned { return (unsigned)(A.xoff * 4 + (hc >> 1) * 128 + (hc & 1) * 32); };
    auto issue_halo_at = [&](int i, int hc, unsigned voff) {
        int j = wave + kWaves * i;
        if (j >= kNJ) j = wave;
        dma_buf16(voff, xr, halo_soff(hc), halo_lds + (unsigned)((hc & 1) * kHBuf + j * 1024));
    };
    auto issue_halo = [&](int i, int hc) { issue_halo_at(i, hc, hofft[i * kThreads + tid]); };

    // ---- weight requests: this wave's share (kSlot / 4 bytes) of the slot of the step kAhead ahead ----
    // The request pointer crosses into the NEXT item's weights during the last steps of the current one: w_next = scalar offset
    // of that item's first slot, refreshed once per item.  No branch per step.  (A sub-item requests whole slots like everyone:
    // the 6 KB it does not read are L2 hits of a last round.)
    const unsigned wlane = (unsigned)lane * 16u;
    auto first_slot_of = [&](int nb) -> unsigned { return (unsigned)nb * (unsigned)wsteps * (unsigned)kSlot + (unsigned)wave * (unsigned)(kSlot / kWaves); };
    unsigned w_so = first_slot_of(t_nb);   // scalar offset of the next slot to request (this wave's share of it)
    unsigned w_next = w_so;
    int w_left = wsteps;                   // slots of the current stretch still to request
    int w_slot = 0;
    auto issue_weights = [&]() {
        const unsigned dst = ring_lds + (unsigned)(w_slot * kSlot) + (unsigned)wave * (unsigned)(kSlot / kWaves);
        dma_buf16(wlane, wr, w_so, dst);
        if constexpr (kWPieces == 2) dma_buf16(wlane, wr, w_so + 1024u, dst + 1024u);
        w_slot = (w_slot + 1) & (kRing - 1);
        --w_left;
        const bool wrap = w_left == 0;
        w_so = wrap ? w_next : w_so + (unsigned)kSlot;
        w_left = wrap ? wsteps : w_left;
    };
    auto refresh_w_next = [&](int k) {
        const int w = item_at(k + 1);
        w_next = first_slot_of(tile_major ? 2 * (w / per_group) + ((w % per_group) & 1) : w / ntiles);
    };
    refresh_w_next(0);

    // ---- per-lane LDS read offsets ----
    // B operand (pixels): row-tile i, tap t, plane 0: halo pixel hp = abase[i] + shift(t) (this lane's pixel of the row-tile),
    // piece khalf at position khalf ^ ((hp >> 2) & 3); plane 1 is ^ 32; the second halo buffer is + kHBuf (an immediate).
    // The four offsets of a step are computed when its operands are read (5 vector instructions each against the step's 48
    // MFMAs): a table of all nine shifts is 36 registers the 256 ArchVGPRs beside the accumulators do not have.
    int abase[kRT];
#pragma unroll
    for (int i = 0; i < kRT; ++i) {
        const int rt = wave * kRT + i;
        abase[i] = FLAT ? rt * 32 + (lane & 31) + P + 1 : (rt + 1) * kPitch2D + (lane & 31) + 1;
    }
    auto a_offsets = [&](unsigned (&ao)[kRT], int t) {
        const int shift = (t / 3 - 1) * P + (t % 3 - 1);
#pragma unroll
        for (int i = 0; i < kRT; ++i) {
            int ab = abase[i];
            asm volatile("" : "+v"(ab));  // (opaque: otherwise all 36 offsets are hoisted out of the loop -- and spilled)
            const int hp = ab + shift;
            ao[i] = halo_lds + (unsigned)(hp * 64 + ((khalf ^ ((hp >> 2) & 3)) * 16));
        }
    };
    // A operand (weights): column tile j, plane p: row n = 32 j + (lane & 31) -> j and p are immediates (1024 j + kSlot / 2 p);
    // a sub-item reads its one column tile (sub_cq) into fragment 0
    const unsigned boff = ring_lds + (unsigned)((lane & 31) * 32 + ((khalf ^ (((lane & 31) >> 3) & 1)) * 16));
    const unsigned boff_sub = boff + (unsigned)(sub_cq * 1024);

    auto lds16 = [](unsigned off) -> f16x8 { return *reinterpret_cast<const lds_f16x8 *>((size_t)off); };
    auto load_frags = [&](Frags &f, int tap, int buf_imm, int slot_, auto sub) {
        constexpr bool kSub = decltype(sub)::value;
        unsigned ao[kRT];
        a_offsets(ao, tap);
#pragma unroll
        for (int i = 0; i < kRT; ++i) {
            f.ah[i] = lds16(ao[i] + (unsigned)buf_imm);
            f.al[i] = lds16((ao[i] ^ 32u) + (unsigned)buf_imm);
        }
        const unsigned wb = (kSub ? boff_sub : boff) + (unsigned)(slot_ * kSlot);
#pragma unroll
        for (int j = 0; j < (kSub ? 1 : kCT); ++j) {
            f.bh[j] = lds16(wb + (unsigned)(j * 1024));
            f.bl[j] = lds16(wb + (unsigned)(j * 1024 + kSlot / 2));
        }
    };

    // the same in four parts -- part r: the pixel fragments of row-tile r and the weight fragments of column tile r -- so that the
    // 16 reads of a step can sit four at a time in front of the four 12-MFMA groups instead of in one block at the step's head
    auto load_part = [&](Frags &f, int tap, int buf_imm, int slot_, auto sub, int r) {
        constexpr bool kSub = decltype(sub)::value;
        const int shift = (tap / 3 - 1) * P + (tap % 3 - 1);
        int ab = abase[r];
        asm volatile("" : "+v"(ab));
        const int hp = ab + shift;
        const unsigned ao = halo_lds + (unsigned)(hp * 64 + ((khalf ^ ((hp >> 2) & 3)) * 16));
        f.ah[r] = lds16(ao + (unsigned)buf_imm);
        f.al[r] = lds16((ao ^ 32u) + (unsigned)buf_imm);
        const unsigned wb = (kSub ? boff_sub : boff) + (unsigned)(slot_ * kSlot);
        if (r < (kSub ? 1 : kCT)) {
            f.bh[r] = lds16(wb + (unsigned)(r * 1024));
            f.bl[r] = lds16(wb + (unsigned)(r * 1024 + kSlot / 2));
        }
    };

    f32x16 acc[kRT][kCT];
    // the 3 CT MFMAs of row-tile i: lo x hi, hi x lo, hi x hi over the column tiles (same accumulator every CT-th instruction).
    // `first`: the first k-step of a work item starts the accumulators from the MFMA's constant-zero C operand -- 256 registers
    // are never zeroed by hand (hipcc materialises the zeros in 256 ArchVGPRs first and spills everything else around them)
    auto mma_row = [&](const Frags &f, int i, auto first, auto sub) {
        constexpr bool kFirst = decltype(first)::value;
        constexpr int kJ = decltype(sub)::value ? 1 : kCT;
        if (C4_DBG(4)) return;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < kJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[i], kFirst ? zero : acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < kJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < kJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[i], acc[i][j], 0, 0, 0);
    };

    // epilogue parameters: [block slot][bias | scale | shift][kBN]; channel-major: slot 0 = the current block (reloaded when it
    // changes), tile-major: both blocks of the layer, once
    auto load_prm = [&](int nb) {
        const int nslots = tile_major ? 2 : 1;
        for (int i = tid; i < 3 * kBN * nslots; i += kThreads) {
            const int sl = i / (3 * kBN), j = i - sl * 3 * kBN;
            const int c = j % kBN, which = j / kBN;
            const int n = (tile_major ? (nb & ~1) + sl : nb) * kBN + c;
            prm[i] = which == 0 ? (A.bias ? A.bias[n] : 0.f) : which == 1 ? (A.scale ? A.scale[n] : 1.f) : (A.shift ? A.shift[n] : 0.f);
        }
    };

    // ---- prologue: halo (chunk 0, half 0) of the first tile, weights of steps 0 .. kAhead - 1, epilogue parameters ----
    set_halo_sources(t_b, t_y0, t_x0, t_f0);
#pragma unroll
    for (int i = 0; i < kHSlots; ++i) issue_halo(i, 0);
    req_hc = 1;
#pragma unroll
    for (int r = 0; r < kAhead; ++r) issue_weights();
    load_prm(t_nb);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    float amax = 0.f;
    int slot = 0;   // ring slot of the step being computed
    int chunk = 0;  // chunk being computed
    Frags f0, f1;
    using Full = std::false_type;
    using Sub = std::true_type;

    // One step = one k-step of 16: (half H, tap T) of the current chunk.  CUR holds its operands; the operands of the next step
    // (half NH, tap NT: buffer NH, next ring slot) are read into NXT while it multiplies.  Requests: the weights of the step
    // kAhead ahead, and in the first taps the halo pieces of the half-chunk after this one.  The counted wait at the end lets the
    // requests of this step and the two before it stay in flight (c4_in_flight).
#define C4_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory")
// -DQMRI_C4_BAR3F (experiment: the proof of DESIGN 6.6's mechanism): the barrier every third step PLUS one bare s_barrier behind step 1 of every
// work item -- the operands of an item's step 0 are read behind next_item's barrier, i.e. in the same barrier interval as the request of
// step 2, which lands in step 0's ring slot.  (The hunt's other builds -- every wait widened to vmcnt(0) lgkmcnt(0), one wave put to sleep behind
// every loop barrier, the epilogue's stores drained in next_item, idle cycles behind every request -- are in git history: round 6.)
#if defined(QMRI_C4_BAR3F)
#define C4_ITEM_FENCE(H, T) if constexpr ((H) == 0 && (T) == 1) { if (chunk == 0) asm volatile("s_barrier" ::: "memory"); }
#else
#define C4_ITEM_FENCE(H, T)
#endif
#define C4_STEP_(H, T, CUR, NXT, FIRST, SUB)                                                                       \
    {                                                                                                              \
        constexpr int NH_ = (T) == 8 ? 1 - (H) : (H);                                                              \
        constexpr int NT_ = (T) == 8 ? 0 : (T) + 1;                                                                \
        constexpr int kHP_ = c4_halo_pieces((T), kHSlots), kHF_ = c4_halo_first((T), kHSlots);                     \
        /* (the last step of a work item reads nothing ahead: the epilogue does not need 64 live operand registers) */ \
        const bool pf_ = !((H) == 1 && (T) == 8 && chunk + 1 == A.chunks) && !C4_DBG(8);                           \
        const int nslot_ = (slot + 1) & (kRing - 1);                                                               \
        /* this step's halo offsets leave the LDS table FIRST: read right in front of the request they would put an lgkmcnt wait -- */ \
        /* which also drains the operand reads queued before it -- in the middle of the MFMA stream */           \
        unsigned hv_[4] = {0u, 0u, 0u, 0u};                                                                        \
        _Pragma("unroll") for (int hp_ = 0; hp_ < kHP_; ++hp_) hv_[hp_] = hofft[(kHF_ + hp_) * kThreads + tid];    \
        /* part r: the reads of the next step's row-tile r (+ column tile r), the MFMAs of this step's row-tile r; the weight */ \
        /* request rides behind the first group, the halo requests behind the second */                           \
        _Pragma("unroll") for (int r_ = 0; r_ < kRT; ++r_) {                                                       \
            if (pf_) load_part(NXT, NT_, NH_ * kHBuf, nslot_, SUB{}, r_);                                          \
            mma_row(CUR, r_, FIRST{}, SUB{});                                                                      \
            if (r_ == 0 && !C4_DBG(16)) issue_weights();                                                           \
            if constexpr (kHP_ > 0) if (r_ == 1 && !C4_DBG(16)) {                                                  \
                _Pragma("unroll") for (int hp_ = 0; hp_ < kHP_; ++hp_) issue_halo_at(kHF_ + hp_, req_hc, hv_[hp_]); \
            }                                                                                                      \
        }                                                                                                          \
        /* every MFMA group opens with an MFMA; its reads (4 while column tiles are left, 2 after) ride one per MFMA behind it */ \
        c4_sched_step<0, kRT, (SUB::value ? 1 : kCT)>();                                                           \
        constexpr int kN_ = c4_in_flight((T), kWPieces, kHSlots);                                                  \
        if constexpr ((H) == 0 && (T) == 2) { if (chunk == 0) C4_TS_STEP(1); }                                     \
        if constexpr (kBarEvery == 1 || (kBarEvery == 2 && ((9 * (H) + (T)) & 1)) || (kBarEvery == 3 && (T) % 3 == 2)) C4_WAIT(kN_); \
        C4_ITEM_FENCE(H, T)                                                                                         \
        if constexpr ((H) == 0 && (T) == 2) { if (chunk == 0) C4_TS_STEP(2); }                                     \
        slot = (slot + 1) & (kRing - 1);                                                                           \
    }
#define C4_STEP(H, T, CUR, NXT, SUB) C4_STEP_(H, T, CUR, NXT, std::false_type, SUB)
#define C4_HALF(H, SUB)                                 \
    if (chunk == 0)                                     \
        C4_STEP_(H, 0, f0, f1, std::true_type, SUB)     \
    else                                                \
        C4_STEP(H, 0, f0, f1, SUB)                      \
    C4_STEP(H, 1, f1, f0, SUB)                          \
    C4_STEP(H, 2, f0, f1, SUB)                          \
    C4_STEP(H, 3, f1, f0, SUB)                          \
    C4_STEP(H, 4, f0, f1, SUB)                          \
    C4_STEP(H, 5, f1, f0, SUB)                          \
    C4_STEP(H, 6, f0, f1, SUB)                          \
    C4_STEP(H, 7, f1, f0, SUB)                          \
    C4_STEP(H, 8, f0, f1, SUB)
    // (nine steps swap the roles of f0 / f1: the second half of a chunk runs with them exchanged)
#define C4_HALF_X(H, SUB)                               \
    C4_STEP(H, 0, f1, f0, SUB)                          \
    C4_STEP(H, 1, f0, f1, SUB)                          \
    C4_STEP(H, 2, f1, f0, SUB)                          \
    C4_STEP(H, 3, f0, f1, SUB)                          \
    C4_STEP(H, 4, f1, f0, SUB)                          \
    C4_STEP(H, 5, f0, f1, SUB)                          \
    C4_STEP(H, 6, f1, f0, SUB)                          \
    C4_STEP(H, 7, f0, f1, SUB)                          \
    C4_STEP(H, 8, f1, f0, SUB)
    // all chunks of the current item.  Requests during half 0: (chunk, half 1) -> buffer 1; during half 1: (chunk + 1, half 0),
    // or the next item's first half-chunk -> buffer 0 (its halo sources are computed when the request pointer gets there)
#define C4_CHUNKS(SUB)                                                                                             \
    for (chunk = 0; chunk < A.chunks; ++chunk) {                                                                   \
        C4_HALF(0, SUB)                                                                                            \
        ++req_hc;                                                                                                  \
        if (req_hc == nhc) {                                                                                       \
            req_hc = 0;                                                                                            \
            ++req_k;                                                                                               \
            int nb_, b_, y0_, x0_, f0_;                                                                            \
            decode_work(item_at(req_k), nb_, b_, y0_, x0_, f0_);                                                   \
            if (!C4_DBG(2)) set_halo_sources(b_, y0_, x0_, f0_);                                                   \
        }                                                                                                          \
        C4_HALF_X(1, SUB)                                                                                          \
        ++req_hc;                                                                                                  \
    }

    // ======================= epilogue of a work item =======================
    // staging: buffer 1 (the half-chunk that just finished; buffer 0 already holds the next item's first), 4 KB per wave
    auto epilogue = [&](auto sub) {
        constexpr bool kSub = decltype(sub)::value;
        // (the lane index is made opaque here: hipcc otherwise computes every per-lane address of the epilogue before the main
        //  loop, cannot keep them in registers through it, and reloads them from scratch for every tile)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int lane = lane_e, khalf = lane_e >> 5;
        // (and the item's geometry: the output addresses only depend on it, so hipcc computes them BEFORE the chunk loop, spills
        //  them across it and reloads one per global store -- each reload a vmcnt(0), i.e. every store behind the previous one)
        int ey0 = t_y0, ex0 = t_x0, eb = t_b, ef0 = t_f0, enb = t_nb;
        asm volatile("" : "+s"(ey0), "+s"(ex0), "+s"(eb), "+s"(ef0), "+s"(enb));
        const int t_y0 = ey0, t_x0 = ex0, t_b = eb, t_f0 = ef0, t_nb = enb;
        unsigned char *stage = halo + kHBuf + wave * G::STAGE;
        if (FLAT) {
            for (int i = tid; i < kMTile; i += kThreads) outpix[i] = flat_to_pix(t_f0 + i, P, A.H, A.W, A.B);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const float winv = A.winv;
        const float floor_ = A.relu ? 0.f : -__builtin_inff();  // ReLU as a lower bound: one v_max, no select on the flag
        const int n0 = t_nb * kBN;
        const int px_l = lane & 31;
        // THE ACCUMULATORS LEAVE THE REGISTER FILE RAW.  An MFMA tile has this lane's pixel (lane & 31) and 16 channels
        // (8 q + 4 khalf + r); VALU instructions cannot read AccVGPRs, and an epilogue that starts with arithmetic on them makes the
        // register allocator move its whole live range into ArchVGPRs at the epilogue's entry (160 v_accvgpr_read in a row, six
        // tiles to SCRATCH, every reload behind s_waitcnt vmcnt(0) = behind the acknowledgement of all stores issued so far).
        // ds_write takes AccVGPR data: four 16-byte pieces (4 channels, fp32) per lane go to a wave-private 4 KB window
        // [32 pixels][128 B], piece g = 2 q + khalf of pixel p at a swizzled position (conflict-free both ways), and come
        // back pixel-major: task t = 0, 1 of a lane = pixel 16 t + (lane >> 2), channel octet lane & 3 -- 8 values -> bias, ReLU,
        // BatchNormalization affine, split -> one 16-byte piece of the hi plane and one of the lo plane of the pixel's 128-byte
        // record.  A lane's octet is the same for every tile of a column tile: its 24 parameters are fetched once.
        // Two windows per wave: the tiles of a row pair are written back to back, so the second write and both read-backs overlap
        // the first tile's arithmetic, and the fused 2 x 2 max-pool finds both rows staged.
        const int oc = lane & 3, opx = lane >> 2;
        // (ds_write_b128 is served in groups of 8 contiguous lanes over 8 slots of 16 B, ds_read_b128 in the guide's groups of 16 over
        //  16 slots: this swizzle is conflict-free for the writes, the read-back and -- with the pixel parity swapped in every other
        //  group of 16 lanes -- the pooling reads; scripts/lds_bank_check.py)
        auto piece_off = [&](int px, int g) -> int { return px * 128 + ((g ^ (((px & 1) << 2) | ((px >> 1) & 3))) * 16); };
        struct Prm8 {
            f32x4 b[2], s[2], t[2];
        };
        // (v_pk_fma_f32 for the two fused multiply-adds of a value pair: measured, -0.3 % here and +6 % on enc0_kernel -- not used)
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
        // The row pairs of all column tiles form ONE software pipeline: pair u + 1 is written to the windows right after pair u's
        // read-back has been issued (LDS instructions of a wave execute in order), so its write and the read-back's latency run
        // under pair u's arithmetic and stores instead of in front of every pair.
        constexpr int kJ = kSub ? 1 : kCT, kNP = kRT / 2, kU = kJ * kNP;
        auto put_tile = [&](int i, int j, int win) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x16 &a = acc[i][j];
                const f32x4 piece = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                *reinterpret_cast<f32x4 *>(stage + win * 4096 + piece_off(px_l, 2 * q + khalf)) = piece;
            }
        };
        auto put_pair = [&](int u) {
            put_tile(2 * (u % kNP), u / kNP, 0);
            put_tile(2 * (u % kNP) + 1, u / kNP, 1);
        };
        put_pair(0);
        Prm8 p;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int j = u / kNP, pr = u % kNP;
            const int jc = kSub ? sub_cq : j;
            const int cbase = n0 + jc * 32;
            if (pr == 0) {
                const float *pp = prm + (tile_major ? (t_nb & 1) * 3 * kBN : 0) + jc * 32 + 8 * oc;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    p.b[h] = *reinterpret_cast<const f32x4 *>(pp + 4 * h);
                    p.s[h] = *reinterpret_cast<const f32x4 *>(pp + kBN + 4 * h);
                    p.t[h] = *reinterpret_cast<const f32x4 *>(pp + 2 * kBN + 4 * h);
                }
            }
            // (addresses are rebuilt per pair from a scalar row pointer + a small per-lane offset of a lane index made opaque HERE:
            //  as loop invariants of the epilogue they get spilled and every store waits for a scratch reload)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int oc_t = lane_t & 3, opx_t = lane_t >> 2;
            const bool pool = !FLAT && A.pool_y;
            // ---- read-back of the pair (+ the pooling reads: a lane = one pooled pixel's octet, the four source pixels' pieces) ----
            f32x4 r[2][2][2], rp[2][2][2];
#pragma unroll
            for (int win = 0; win < 2; ++win) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) r[win][t][h] = *reinterpret_cast<const f32x4 *>(stage + win * 4096 + piece_off(t * 16 + opx_t, 2 * oc_t + h));
                }
            }
            if (pool) {
                const int dsw = (lane_t >> 4) & 1;  // (which of the two columns a lane reads first: see piece_off)
#pragma unroll
                for (int win = 0; win < 2; ++win) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) rp[win][d][h] = *reinterpret_cast<const f32x4 *>(stage + win * 4096 + piece_off(2 * opx_t + (d ^ dsw), 2 * oc_t + h));
                    }
                }
            }
            if (u + 1 < kU) put_pair(u + 1);
            // ---- the pair's two tiles: bias, ReLU, affine, saturation tracking, split, stores ----
#pragma unroll
            for (int win = 0; win < 2; ++win) {
                const int i = 2 * pr + win;
                uint4 hi[2], lo[2];
                float tmax[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v[8];
                    finish8(r[win][t][0], r[win][t][1], p, v);
                    // saturation tracking over the values that are STORED: a discarded position (a frame position of the flattened
                    // stack, a row below the image in a partial tile) above 65504 would re-run the whole forward scaled
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
                        amax = fmaxf(amax, pix >= 0 ? tmax[t] : 0.f);
                        asm volatile("" : "+v"(amax));  // (pinned here: hipcc otherwise sinks the whole max chain behind the epilogue and keeps every value alive until then)
                        if (pix >= 0 && !C4_DBG(32)) {
                            unsigned char *dst = static_cast<unsigned char *>(A.y) + ((long long)pix * A.ldy + A.yoff + cbase) * 4 + oc_t * 16;
                            nt_store16(dst, hi[t]);
                            nt_store16(dst + 64, lo[t]);
                        }
                    }
                } else {
                    const int yy = t_y0 + wave * kRT + i;  // (scalar)
                    if (yy < A.H) {
                        unsigned char *rowp = static_cast<unsigned char *>(A.y) +
                                              (((long long)(t_b * A.H + yy) * A.W + t_x0) * A.ldy + A.yoff + cbase) * 4;  // scalar pointer
                        const unsigned pstep = (unsigned)A.ldy * 4u;  // bytes per output pixel
                        const int xlim = A.W - t_x0;  // (scalar) pixels of this tile inside the image: < 32 only in the last column tile of a level with W % 32 != 0
                        if (xlim >= 32) {
                            amax = fmaxf(fmaxf(amax, tmax[0]), tmax[1]);
                            asm volatile("" : "+v"(amax));  // (pinned: see the flattened branch)
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const unsigned off = (unsigned)(t * 16 + opx_t) * pstep + (unsigned)oc_t * 16u;
                                if (!C4_DBG(32)) {
                                    nt_store16(rowp + (size_t)off, hi[t]);
                                    nt_store16(rowp + (size_t)(off + 64u), lo[t]);
                                }
                            }
                        } else {  // ragged tile: pixels at or beyond W are neither stored nor tracked
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const bool in = t * 16 + opx_t < xlim;
                                amax = fmaxf(amax, in ? tmax[t] : 0.f);
                                asm volatile("" : "+v"(amax));
                                const unsigned off = (unsigned)(t * 16 + opx_t) * pstep + (unsigned)oc_t * 16u;
                                if (in && !C4_DBG(32)) {
                                    nt_store16(rowp + (size_t)off, hi[t]);
                                    nt_store16(rowp + (size_t)(off + 64u), lo[t]);
                                }
                            }
                        }
                    }
                }
            }
            // ---- fused MaxPooling2D(2x2): rows 2 pr, 2 pr + 1 of this wave -> 16 pooled pixels x 32 channels ----
            if (pool) {
                float m[8];
#pragma unroll
                for (int win = 0; win < 2; ++win) {
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float v[8];
                        finish8(rp[win][d][0], rp[win][d][1], p, v);
#pragma unroll
                        for (int k = 0; k < 8; ++k) m[k] = (win == 0 && d == 0) ? v[k] : fmaxf(m[k], v[k]);
                    }
                }
                uint4 hi, lo;
                split8(m, hi, lo);
                const int Hp = A.H >> 1, Wp = A.W >> 1;
                const int yy = (t_y0 >> 1) + wave * (kRT / 2) + pr;
                if (yy < Hp && (t_x0 >> 1) + opx_t < Wp) {  // (the column test only bites in the ragged last column tile)
                    const long long pix = (long long)(t_b * Hp + yy) * Wp + (t_x0 >> 1) + opx_t;
                    unsigned char *dst = static_cast<unsigned char *>(A.pool_y) + (pix * A.pool_ld + cbase) * 4 + oc_t * 16;
                    nt_store16(dst, hi);
                    nt_store16(dst + 64, lo);
                }
            }
            // one row pair at a time (left alone, the scheduler piles up the next pairs' window traffic and the addresses spill)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // between two items: geometry and epilogue parameters of the next one, and everyone done with the staging windows (the next
    // half-chunk's DMA lands there).  Returns false after the block's last item.
    auto next_item = [&]() -> bool {
        if (++cur >= my_items) return false;
        const int prev_nb = t_nb;
        decode_work(item_at(cur), t_nb, t_b, t_y0, t_x0, t_f0);
        refresh_w_next(cur);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tile_major ? (t_nb >> 1) != (prev_nb >> 1) : t_nb != prev_nb) {
            load_prm(t_nb);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        return true;
    };

#ifdef QMRI_C4_EXPERIMENTS
    if (C4_DBG(64) && (blockIdx.x & 8)) {  // (experiment: every other group of 8 blocks -- one per XCD -- starts half an item late)
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)A.chunks * 9ull * 2300ull;
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
    // (output-only: the fragment sets are dead across the epilogue -- the next item's first step reads f0 after load_frags and fills
    //  f1 before its second -- but the fill is conditional (pf_), so to the compiler f1's old contents stay live: 64 registers the
    //  epilogue does not have, paid for with accumulator tiles in scratch and a vmcnt(0) -- every store acknowledged -- per reload)
    auto kill_frags = [&](Frags &f) {
#pragma unroll
        for (int i = 0; i < kRT; ++i) {
            asm volatile("" : "=v"(f.ah[i]));
            asm volatile("" : "=v"(f.al[i]));
        }
#pragma unroll
        for (int j = 0; j < kCT; ++j) {
            asm volatile("" : "=v"(f.bh[j]));
            asm volatile("" : "=v"(f.bl[j]));
        }
    };
    bool more = true;
    if (my_full > 0) {
        load_frags(f0, 0, 0, slot, Full{});  // operands of the very first step
        while (true) {
            C4_TS(0);
            C4_CHUNKS(Full)
            C4_TS(3);
            kill_frags(f0);
            kill_frags(f1);
            if (!C4_DBG(1)) epilogue(Full{});
            C4_TS(4);
            more = next_item();
            if (more) { --cur; C4_TS(5); ++cur; }
            if (!more || cur >= my_full) break;
            load_frags(f0, 0, 0, slot, Full{});  // operands of the new item's first step (landed: see C4_STEP_)
        }
    }
    if (more && has_sub) {  // the block's last item: one 32-channel column tile of a leftover item
        load_frags(f0, 0, 0, slot, Sub{});
        C4_CHUNKS(Sub)
        kill_frags(f0);
        kill_frags(f1);
        epilogue(Sub{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this block's DMA may land after it has exited
#ifdef QMRI_C4_TIMELINE
    if (blockIdx.x == 8 && tid == 0 && my_items >= 4) {
        const unsigned r = atomicAdd(&g_c4_tl_n, 1u);
        if (r < 256u) {
            unsigned long long *o = g_c4_tl + r * 16;
            o[0] = (unsigned long long)CT;
            o[1] = (unsigned long long)FLAT;
            o[2] = (unsigned long long)A.chunks;
            o[3] = (unsigned long long)my_items;
            for (int it = 0; it < 2; ++it) {
                const unsigned long long *t = tsbuf + it * 8;
                o[4 + it * 5 + 0] = t[3] - t[0];  // the item's k loop
                o[4 + it * 5 + 1] = t[4] - t[3];  // epilogue
                o[4 + it * 5 + 2] = t[5] - t[4];  // next_item
                o[4 + it * 5 + 3] = t[1] - t[0];  // (QMRI_C4_TIMELINE >= 2: the first three steps, and their counted wait + barrier)
                o[4 + it * 5 + 4] = t[2] - t[1];
            }
        }
    }
#endif
    if (A.sat && amax > 65504.f) *A.sat = 1;
}

template <bool FLAT, int CT>
static constexpr size_t c4_lds_bytes() {
    using G = C4Geo<FLAT, CT>;
    size_t n = (size_t)2 * G::HBUF + (size_t)kRing * 2048 * CT + (FLAT ? G::MTILE * 4 : 0) + (size_t)3 * 32 * CT * 4 * kPrmBlocksC4 + (size_t)G::HSLOTS * kThreads * 4;
#ifdef QMRI_C4_TIMELINE
    n += 128;
#endif
    return n;
}
static_assert(c4_lds_bytes<false, 2>() <= 160 * 1024 && c4_lds_bytes<true, 2>() <= 160 * 1024 && c4_lds_bytes<false, 4>() <= 160 * 1024 &&
                  c4_lds_bytes<true, 4>() <= 160 * 1024,
              "LDS");
// tile geometry of a layer on this kernel (host side of C4Geo)
static int c4_tile_rows(int Cout) { return conv_c4_block_channels(Cout) == 128 ? C4Geo<false, 4>::ROWS : C4Geo<false, 2>::ROWS; }
static int c4_flat_tile(int Cout) { return conv_c4_block_channels(Cout) == 128 ? C4Geo<true, 4>::MTILE : C4Geo<true, 2>::MTILE; }

// which layers the kernel takes: >= 128 output channels in blocks of 128, 32-channel input chunks, a level it tiles
int conv_c4_block_channels(int Cout) { return Cout % 128 == 0 ? 128 : 64; }

bool conv_c4_supported(const ConvS3Args &k) {
    if (k.deconv || k.one || k.head_w) return false;
    if (k.Cin % 32 || k.Cout % 64) return false;
    // 32-bit source offsets relative to the first image a halo touches (the descriptor base moves per work item): the bytes of
    // the images ONE halo can span (+ the chunk's 128) must stay below the descriptor's num_records.  A property of the layer's
    // shape, not of the batch: a slice's bits must not depend on the pass it travels in.
    const unsigned long long img = (unsigned long long)k.H * k.W * (unsigned long long)k.ldx * 4ull;
    if (!conv_tiles_flat(k.W)) {  // image tiles; W % 32 != 0 (and too wide for the flattened tiling): the last column tile is ragged
        if (img >= (unsigned long long)kPadOff) return false;
        return !k.pool_y || (!(k.H & 1) && !(k.W & 1));
    }
    if (k.pool_y) return false;  // flattened zero-framed stack (halo: 512 + 2 (W + 2) + 2 <= 624 pixels)
    const unsigned long long span = (624ull + (unsigned long long)(k.H + 1) * (k.W + 2) - 1) / ((unsigned long long)(k.H + 1) * (k.W + 2)) + 1;
    return span * img < (unsigned long long)kPadOff;
}

template <bool FLAT, int CT>
static hipError_t c4_launch_t(ConvS3Args &k, int num_cu, hipStream_t stream) {
    auto fn = conv_c4_kernel<FLAT, CT>;
    constexpr size_t lds = c4_lds_bytes<FLAT, CT>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int grid = k.nwork < num_cu ? k.nwork : num_cu;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(kThreads), lds, stream, k);
    return hipGetLastError();
}

hipError_t conv_c4_launch(const ConvS3Args &k0, int num_cu, hipStream_t stream) {
    ConvS3Args k = k0;
    if (!conv_c4_supported(k) || !k.w_c4) return hipErrorInvalidValue;
    const bool flat = conv_tiles_flat(k.W);
    k.chunks = k.Cin / 32;
    k.steps = k.chunks * 18;
    const int bn = conv_c4_block_channels(k.Cout);
    k.nb = k.Cout / bn;
    if (flat) {
        k.P = k.W + 2;
        const long long span = (long long)k.B * (k.H + 1) * k.P - k.P;
        const int mtile = c4_flat_tile(k.Cout);
        k.ntiles = (int)((span + mtile - 1) / mtile);
        k.tiles_x = k.tiles_y = 0;
    } else {
        k.P = kPitch2D;
        k.tiles_x = (k.W + 31) / 32;
        const int rows = c4_tile_rows(k.Cout);
        k.tiles_y = (k.H + rows - 1) / rows;
        k.ntiles = k.B * k.tiles_x * k.tiles_y;
    }
    k.nj = 0;  // (conv_s3_kernel's field: the halo geometry is C4Geo's here)
    k.nwork = k.nb * k.ntiles;
    // item order: tile-major in groups of two channel blocks where a group's weights (<= 2.4 MB) stay in an XCD's 4 MB L2 beside the activations -- a
    // property of the layer (QMRI_C4_ORDER = 0 / 1 forces channel- / tile-major: the A/B switch).  The sums do not depend on it.
    static const int order = [] {
        const char *e = std::getenv("QMRI_C4_ORDER");
        return e ? std::atoi(e) : -1;
    }();
    const size_t pair_bytes = (size_t)k.Cin * 9 * 2 * bn * 4;  // weights of two channel blocks
    k.tile_group = (k.nb % 2 == 0 && (order < 0 ? pair_bytes <= (size_t)2400 << 10 : order != 0)) ? 1 : 0;
    static const int split = [] {
        const char *e = std::getenv("QMRI_C4_SPLIT");
        return e ? std::atoi(e) : 1;
    }();
    k.c4_split = k0.c4_split < 0 ? 0 : split;  // (< 0: the caller wants whole items only -- tests compare the two bit for bit)
    static const int dbg = [] {
        const char *e = std::getenv("QMRI_C4_DBG");
        return e ? std::atoi(e) : 0;
    }();
    k.dbg = dbg;
    (void)hipGetLastError();
    if (bn == 128) return flat ? c4_launch_t<true, 4>(k, num_cu, stream) : c4_launch_t<false, 4>(k, num_cu, stream);
    return flat ? c4_launch_t<true, 2>(k, num_cu, stream) : c4_launch_t<false, 2>(k, num_cu, stream);
}

}  // namespace qmri

#ifdef QMRI_C4_TIMELINE
extern "C" int qmri_debug_c4_timeline(unsigned long long *out, int max_records) {
    unsigned int n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(qmri::g_c4_tl_n), sizeof(n)) != hipSuccess) return -1;
    if (n > 256u) n = 256u;
    if ((int)n > max_records) n = (unsigned)max_records;
    if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(qmri::g_c4_tl), (size_t)n * 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(qmri::g_c4_tl_n), &zero, sizeof(zero));
    return (int)n;
}
#endif
