// Fused inverted-residual (MBConv) block for gfx950: expand 1x1 -> depthwise kxk -> project 1x1
// (+ residual) in ONE launch, with the expanded tensor kept in LDS.
//
// In the reference these are three Conv nodes (+ activations, + Add) of the ONNX graph that
// birdnet_onnx::Classifier runs through ONNX Runtime (reference src/inference/classifier.rs:478-488;
// SURVEY.md 8a-8).  Layer by layer the expanded tensor is written and re-read twice (4.7 MB per
// segment for the first block alone), which is what bounds the first stages; here it never leaves
// the CU.
//
// One workgroup (4 waves) owns an output tile TH x TW of S consecutive segments.
//   in-tile   IH x IW = ((TH-1)s + k) x ((TW-1)s + k) input positions; the part inside the image is
//             the "valid rect", M = S * vh * vw source rows.
//   per chunk of CE expanded channels:
//     P1  E[M x CE]   = act(X[M x Cin] . We[Cin x CE] + be)   MFMA, computed as E^T = We^T X^T so that a
//                       lane holds 4 consecutive channels of one row (one 16-byte LDS write); X resident
//                       in registers, We fragment-major in LDS, be the accumulators' start value; rows go
//                       to their slots of the LDS grid Es (the grid's out-of-image border stays zero =
//                       the depthwise conv's zero padding)
//     P2  D[P x CE]   = act(dw_kxk(Es) + bd)                  VALU + LDS, 4 channels per lane
//     P3  acc[P x Co] += D[P x CE] . Wp[CE x Co]              f32 MFMA, accumulators live across chunks
//   epilogue: + bp, activation, + residual, NHWC store.
// This header holds the kernel template, its launcher and the table-entry macros; it is compiled once per activation
// (kernels_mbconv_gelu.hip / _swish.hip / _relu6.hip: three translation units that hipcc builds in parallel -- one unit with
// all three copies of the table took 4.5 minutes) and the planner (kernels_mbconv.hip) sees only MbCfg.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"
#include "mbconv_cfg.hpp"

namespace bh {

// Round 6, VERDICT r5 next #2: the fixed cost of a tile, each lead BUILT and measured on its own (profiles/r6_e_setup_leads.txt):
//   BH_MB_BUFLOAD   the wave's rows of X through a buffer descriptor: lanes that must read zero (rows past the tile, columns past
//                   Cin) take an out-of-range offset and the hardware's range check returns 0 -- no select per loaded value, 32-bit
//                   offsets instead of 64-bit addresses
//   BH_MB_HOSTRCP   tile decode (tile -> segment group, tile row, tile column) by host-computed reciprocals (MbDesc::rcp_*): two
//                   s_mul_hi_u32 instead of two expanded 32-bit divisions
//   BH_MB_CONSTDIV  row decode of a tile whose rectangle lies inside the image (valid width = the tile's own IW, a compile-time
//                   number): division by a constant
#ifndef BH_MB_BUFLOAD
#define BH_MB_BUFLOAD 1
#endif
#ifndef BH_MB_HOSTRCP
#define BH_MB_HOSTRCP 1
#endif
#ifndef BH_MB_CONSTDIV
#define BH_MB_CONSTDIV 1
#endif
//   BH_MB_BUFSTORE  the epilogue's stores (and the residual loads the accumulators start from) through buffer descriptors: a lane
//                   without a pixel or past Cout takes an out-of-range offset -- no exec-mask branch per store, 32-bit offsets
#ifndef BH_MB_BUFSTORE
#define BH_MB_BUFSTORE 1
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

// n / d for 0 <= n < 2^22, d > 0 through the float reciprocal (one multiply, a convert and a
// fix-up) instead of the ~35-instruction 32-bit integer division sequence
// (products through the 24-bit multiplier: v_mul_lo_u32 runs at a quarter of the rate of v_mad_u32_u24)
__device__ __forceinline__ int mb_div(int n, int d, float rcp_d) {
    int q = (int)((float)n * rcp_d);
    q += (n - __mul24(q, d) >= d) ? 1 : 0;
    q -= (n - __mul24(q, d) < 0) ? 1 : 0;
    return q;
}

// diagnostic phase clock (only when d.stamps != nullptr): cycles since the last stamp are summed per
// phase in registers and added to the global counters once, at the end, by lane 0 of every wave
// a pointer of the descriptor as what it is, a global one: the buffer descriptors below are made from it, and where the compiler
// cannot infer the address space (the EXPERIMENTS build's persistent / ring variants) a generic pointer costs an aperture test --
// which this compiler emits as an illegal V_CMP on src_shared_base
typedef __attribute__((address_space(1))) float mb_gfloat;
static __device__ __forceinline__ mb_gfloat *mb_global(const float *p) { return (mb_gfloat *)const_cast<float *>(p); }

struct MbClock {
    unsigned long long last, acc[8];
};
#ifdef BIRDA_HIP_EXPERIMENTS
#define mb_stamp(stamps, clk, ph)                                          \
    do {                                                                   \
        if (stamps) {                                                      \
            const unsigned long long now_ = __builtin_readcyclecounter(); \
            clk.acc[ph] += now_ - clk.last;                                \
            clk.last = now_;                                               \
        }                                                                  \
    } while (0)
#else
// (product build: the phase clock is compiled out -- its eight 64-bit accumulators were live across the whole kernel and every
//  phase boundary carried an s_memtime and a branch; make EXPERIMENTS=1 brings it back for tools/gpu_mb_stamps.py)
#define mb_stamp(stamps, clk, ph) do { } while (0)
#endif

// Template parameters
//   KS, ST      depthwise kernel size / stride          CE     expanded channels per chunk
//   KG          16-deep k groups of the expand GEMM (ceil(Cin / 16))
//   RT_W        most source-row tiles (16 rows) one wave owns in P1
//   NCS         waves splitting the chunk's columns in P1 (1 or 2)
//   WM x WN     wave grid of P3, MT_W x NT_W accumulator tiles per wave
//   TWL         log2(tile width)   XBL  log2(pixels per lane along x in P2)   SS  segments per workgroup
//   OCC         waves per SIMD the register allocator must leave room for
//
// Operand residency: the wave's A fragments of the expand GEMM (its rows of X, all of Cin) are
// loaded ONCE and stay in registers for every chunk.  The chunk's weights reach LDS by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs), issued behind one barrier and drained before the next:
//     after B1(ch): We+be of chunk ch+1, Wp of chunk ch      (waited before B2(ch))
//     after B2(ch): Wd+bd of chunk ch+1                      (waited before B1(ch+1))
// so the MFMA loops read every operand from registers or LDS and never wait on memory.
template <int NFLOATS, int NW = 4>
__device__ __forceinline__ void mb_dma(const float *gsrc, float *lds_dst, int wave, int lane) {
    constexpr int NP = (NFLOATS + 255) / 256;  // 1-KiB pieces, dealt round-robin to the NW waves
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += NW) {
        const int p = p0 + wave;
        const int off = p * 256 + lane * 4;
        if (p < NP && off < NFLOATS) {
            // Inline asm on purpose: after the builtin form hipcc drains vmcnt(0) in front of the next
            // LDS read of ANY array (it cannot prove the DMA's destination is not read), which made
            // every transfer synchronous.  The waits are placed by hand: mb_dma_wait() before the
            // barrier that precedes the first reader.  M0 = LDS byte address of the piece.
            const unsigned la = (unsigned)(size_t)(lds_dst + p * 256);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(gsrc + off), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
        }
    }
}
// The same transfer addressed by an LDS BYTE ADDRESS instead of a pointer.  In some instantiations hipcc does not fold
// `(unsigned)(size_t)lds_pointer` (a generic pointer: LDS -> flat -> integer) back to the LDS offset and its backend then rejects
// the aperture test it builds ("Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base"); the column-task
// instantiations hit that, so they compute the address from float offsets into the one dynamic shared array.
// One piece: the wave's index is made scalar first, so the piece index, the global base (the source is wave-uniform) and the LDS
// address live in SGPRs and the only per-lane value is the 32-bit byte offset `lane * 16` (the saddr form of the load).  With the
// 64-bit per-lane addresses of mb_dma the 8-wave kernels (256 registers per wave) spilled exactly these, and a scratch reload's
// `s_waitcnt vmcnt(0)` in the middle of a burst of pieces waits for every piece issued before it.
template <int NFLOATS>
__device__ __forceinline__ void mb_dma_piece(const float *gsrc, unsigned lds_byte_addr, int p /* scalar */, int lane) {
    constexpr int NP = (NFLOATS + 255) / 256, TAIL = NFLOATS - (NP - 1) * 256;   // floats in the last piece
    if (p < NP && (TAIL == 256 || p < NP - 1 || lane * 4 < TAIL)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"((unsigned)lane * 16u), "s"(gsrc + (size_t)p * 256), "s"(lds_byte_addr + (unsigned)p * 1024u) : "memory", "m0");
    }
}
template <int NFLOATS, int NW = 4>
__device__ __forceinline__ void mb_dma_at(const float *gsrc, unsigned lds_byte_addr, int wave, int lane) {
    constexpr int NP = (NFLOATS + 255) / 256;
    const int ws = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += NW) mb_dma_piece<NFLOATS>(gsrc, lds_byte_addr, p0 + ws, lane);
}
// part `part` of `nparts` of the calling wave's share of the same transfer: the column-task kernels issue their chunk's weight
// pieces a few at a time between the rows of the depthwise phase -- issued as one burst, the 12-13 pieces of a wave block it for
// ~800 cycles while the vector-memory path takes them in (tools/microbench/lds_fill.hip)
template <int NFLOATS, int NW = 4>
__device__ __forceinline__ void mb_dma_at_part(const float *gsrc, unsigned lds_byte_addr, int wave, int lane, int part, int nparts) {
    constexpr int NP = (NFLOATS + 255) / 256;
    const int ws = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += NW) {
        if ((p0 / NW) % nparts != part) continue;
        mb_dma_piece<NFLOATS>(gsrc, lds_byte_addr, p0 + ws, lane);
    }
}
__device__ __forceinline__ void mb_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The depthwise activation of the fused blocks.  GELU in the f16 modes leaves TWICE the GELU (gelu2x_fast4, kernels.hpp): the
// project planes' exponent carries the factor (api.hip plan_fusion), as the depthwise taps do for the expand GELU.
template <int ACT, int PREC> constexpr bool mb_gelu2x() { return BH_GELU_2X != 0 && PREC != 0 && ACT == ACT_GELU_ERF && BH_GELU_DEGREE == 5; }
template <int ACT, int PREC>
__device__ __forceinline__ void mb_act4(bh_f32x2 &v0, bh_f32x2 &v1) {
#if BH_GELU_DEGREE == 5
    if constexpr (mb_gelu2x<ACT, PREC>()) gelu2x_fast4(v0, v1, kGeluUnscaled);
    else
#endif
        bh_act4<ACT>(v0, v1);
}
template <int ACT, int PREC>
__device__ __forceinline__ bh_f32x2 mb_act2(bh_f32x2 v) {
#if BH_GELU_DEGREE == 5
    if constexpr (mb_gelu2x<ACT, PREC>()) return gelu2x_fast2(v, kGeluUnscaled);
    else
#endif
        return bh_act2<ACT>(v);
}

//   PREC        0: f32 MFMA (16x16x4, KG counts 16-deep groups); 3: f16 hi/lo split, three 16x16x32 MFMAs
//               per product (KG counts 32-deep steps); 1: plain f16 operands (one MFMA, ~1e-3 relative)
//   ACT         the expand and depthwise activation (GELU, swish, ReLU6 or ReLU: bh_act<>, kernels.hpp)
//   COLTH       > 0: whole-image tiles of COLTH rows (stride 1) whose depthwise phase runs as COLUMN tasks -- one lane = one
//               output column (all COLTH rows) x 4 channels.  The late blocks' images are 3 or 6 rows high under a 5x5 kernel:
//               40 / 20 % of the taps of a row-wise task multiply the all-zero padding rows, and its 192 tasks leave a wave idle.
//               A column task loads only the COLTH real rows of its 5 (3) grid columns once, skips the padding rows at compile
//               time, and SS * TW * CE / 4 = 256 of them fill the workgroup.
//   PERSIST     2: STRIP-WALKING workgroups (round 4).  A workgroup owns one tile COLUMN of one segment and walks down the image, tile
//               row by tile row.  The last KS - ST rows of a tile's expanded grid are the first rows of the next tile's: they are
//               kept -- per chunk, in an LDS halo store -- instead of being expanded (and GELU-ed) again, so the expand phase covers
//               only the TH * ST NEW rows of a step (3x3 stride 1 at TH = 8: 10 -> 8 rows, the k - 1 halo rows of the old tiles were
//               a fifth of all expand work), the tile decomposition and the grid's zero fill happen once per strip, and the next
//               tile's chunk-0 weights arrive under the last chunk of this one.  Weights stream per chunk as in the plain kernel.
//               1: persistent workgroups (grid = workgroups that fit on the chip at once) walking the tiles, with the weights of
//               EVERY chunk resident in LDS -- loaded once per workgroup, not once per tile and chunk.  For the early blocks
//               (large images, few channels): a tile's compute is ~5k cycles, and 12 barriers each waiting for a freshly
//               issued L2 -> LDS transfer plus the launch and set-up of 48-96 workgroups per segment were 70 % of their time
//               (tools/abl2.sh: chunk loop without any compute 417 of 1061 us, set-up + epilogue 303).
//   SE          1: pass A of a squeeze-excite block (MbDesc::se): the expand and depthwise phases only -- the depthwise output goes to HBM
//               (d.Dout, f32 NHWC) instead of the LDS planes of the project GEMM, and the per-channel sums of the tile's pixels to
//               d.pool_part, summed in a FIXED order (per-task partial sums through LDS, one thread per channel adds them up: no
//               atomics, so identical segments give identical bits wherever they sit in a batch); no project phase, no epilogue
template <int KS, int ST, int CE, int KG, int RT_W, int NCS, int WM, int WN, int MT_W, int NT_W, int TWL, int XBL,
          int SS, int OCC, int STEM, int PREC, int PERSIST = 0, int ACT = ACT_GELU_ERF, int COLTH = 0, int SE = 0>
__global__ __launch_bounds__(64 * WM * WN, OCC) void mbconv_kernel(const MbDesc d, const int n_seg) {
    static_assert(!STEM || SS == 1, "the stem variant handles one segment per workgroup");
    static_assert(!SE || (PERSIST == 0 && !mb_gelu2x<ACT, PREC>()), "squeeze-excite pass A: plain workgroups, activations at their own scale");
    static_assert(PREC == 0 || CE % 32 == 0 || CE == 16, "f16 project GEMM: 32-deep steps, or one 16-deep step for 16-channel chunks");
    // NW waves: 4 (one per SIMD; the workgroups of a CU interleave) or 8 (two per SIMD inside ONE workgroup: the late blocks, whose
    // whole-image tiles leave room for a single workgroup per CU -- with one wave per SIMD nothing fills the issue bubbles of its
    // dependent vector chains, and its MFMA and vector phases cannot overlap with anybody else's)
    constexpr int NW = WM * WN, NTH = 64 * NW;
    constexpr bool RESIDENT = PERSIST == 1, STRIP = PERSIST == 2;
    constexpr int KH = KS - ST;          // STRIP: grid rows a tile inherits from the tile above
    static_assert(!STRIP || (SS == 1 && COLTH == 0 && KH > 0), "strip-walking: one segment per workgroup, row tasks");
    // the ablation bits of MbDesc::dbg (tuning aids, BIRDA_HIP_MB_DBG) exist in the EXPERIMENTS build only: in the product they are a
    // compile-time zero and every test of them folds away
#ifdef BIRDA_HIP_EXPERIMENTS
    const int dbgv = d.dbg;
#else
    constexpr int dbgv = 0;
#endif
    static_assert(NW == 2 || NW == 4 || NW == 8, "2 (experimental: half-width tiles, twice the workgroups per CU), 4 or 8 waves");
    constexpr int NT_E = CE / 16, NT_U = NT_E / NCS, CES = CE + 4, C4N = CE / 4, TW = 1 << TWL;
    constexpr int POUT_PAD = WM * MT_W * 16, NTOP = WN * NT_W;
    constexpr int XB = 1 << XBL, XBN = TW / XB, NCOL = (XB - 1) * ST + KS;
    constexpr int RSTEP = NW / NCS;  // row-tile stride between a wave's P1 tiles
    constexpr int RG = (RT_W * NT_U <= 8) ? RT_W : (8 / NT_U >= 1 ? 8 / NT_U : 1);  // row tiles in flight
    constexpr int FRAG = PREC ? 512 : 256;             // floats per (k step, column tile): f16 = hi + lo planes
    constexpr int PSTEPS = PREC ? (CE + 31) / 32 : NT_E;  // k steps of the project GEMM per chunk
    constexpr bool P16 = PREC != 0 && CE == 16;        // 16-channel chunks: ONE v_mfma_f32_16x16x16_f16 step, no k padding
    constexpr int WE_FLOATS = KG * NT_E * FRAG + CE;   // We fragments + be
    constexpr int WP_FLOATS = P16 ? NTOP * 256 : PSTEPS * NTOP * FRAG;
    constexpr int WD_FLOATS = KS * KS * CE + CE;       // Wd [tap][CE] + bd
    constexpr int DSH = P16 ? 24 : PSTEPS * 32 + 8;    // f16 D planes: row stride in halves (48 / 80 B: conflict-free reads)
    constexpr int DS_FLOATS = PREC ? POUT_PAD * DSH : POUT_PAD * CES;
    static_assert(NT_U * NCS == NT_E, "column split");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6;
    const int IH = d.IH, IW = d.IW, TH = d.TH, THTW = d.TH << TWL;
    const int egrid = SS * IH * IW;
    float *Es = smem;
    float *Ds = Es + (size_t)(egrid + 1) * CES;  // + 1: trash row that padding source rows write to
    // Weight buffers.  ring == 0: one buffer each, refilled one PHASE ahead of its reader (see above).  ring == 1 (blocks whose
    // chunks are small: the early, large-image blocks): We x 2, Wp x 3, Wd x 2, refilled one whole CHUNK ahead -- a phase of
    // those blocks is a few hundred cycles, shorter than an L2 -> LDS transfer, and the waits in front of both barriers were
    // half of the kernel's wave-cycles (tools/gpu_mb_stamps.py); the ring costs 5-7 KB of LDS.
    // (the ring is a measured alternative, BIRDA_HIP_MB_RING in the EXPERIMENTS build: in the product MbDesc::ring is never set, and
    //  as a compile-time false its buffer selects -- `ch & 1`, `ch % 3` and their multiplies, once per chunk -- and branches fold away)
#ifdef BIRDA_HIP_EXPERIMENTS
    const bool ring = !PERSIST && COLTH == 0 && d.ring != 0;
#else
    constexpr bool ring = false;
#endif
    const int nbuf_e = RESIDENT ? d.nchunks : (ring ? 2 : 1), nbuf_p = RESIDENT ? d.nchunks : (ring ? 3 : 1);
    float *WeS = Ds + DS_FLOATS;
    _Float16 *DsH = reinterpret_cast<_Float16 *>(Ds), *DsL = DsH + POUT_PAD * DSH;   // PREC != 0
    float *WpS = WeS + nbuf_e * WE_FLOATS;
    float *Wds = WpS + nbuf_p * WP_FLOATS;
    int *omap = reinterpret_cast<int *>(Wds + nbuf_e * WD_FLOATS);
    // STRIP: the halo store, [chunk][KH grid rows][IW][CES] -- the bottom rows of every chunk's grid, as the next tile needs them
    const int halo_fl = KH > 0 ? KH * IW * CES : 0;
    float *Hs = reinterpret_cast<float *>(omap + POUT_PAD);
    // (column-task instantiations: LDS byte addresses of the weight buffers, see mb_dma_at; the dynamic shared array follows the
    //  kernel's static LDS, of which there is none here)
    static_assert(COLTH == 0 || PERSIST == 0, "column tasks: no persistent variant");
    const unsigned lds0 = (__builtin_amdgcn_groupstaticsize() + 15u) & ~15u;
    const unsigned we_ba = lds0 + 4u * (unsigned)((egrid + 1) * CES + DS_FLOATS), wp_ba = we_ba + 4u * (unsigned)(nbuf_e * WE_FLOATS),
                   wd_ba = wp_ba + 4u * (unsigned)(nbuf_p * WP_FLOATS);

    // PERSIST: tile = (segment group, tile row, tile column), linear; this workgroup takes every gridDim.x-th one
    // (STRIP: the work list is (segment, tile column); the tile rows are walked inside)
    const int tiles_xy = STRIP ? d.tiles_x : d.tiles_x * d.tiles_y;
    const int n_tiles = tiles_xy * ((n_seg + SS - 1) / SS);
    // Tile of this workgroup.  The launch is one-dimensional; workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has
    // its own L2.  Dealt in launch order, x-neighbouring tiles -- whose input rectangles share their halo columns and the 64-byte
    // sectors at the rectangle's edges -- land on DIFFERENT XCDs and every shared sector is fetched from HBM once per XCD (stem
    // block: 832 MB read per 1 000 segments against 392 MB of input).  XCD x instead takes the x-th EIGHTH of the tile list (tiles
    // in x, then y, then segment order): neighbours in space are neighbours in time on one L2.
    // CHANNEL SPLIT (round 6; MbDesc::ksplit > 1, launches of a few segments under BH_FLAG_LOW_LATENCY): the
    // launch is ksplit workgroups deep (blockIdx.y) and workgroup ks walks the chunks [c0, c1) only -- a quarter of the block's
    // weights instead of all of them, which at a handful of workgroups on the whole chip is what a late block's time consists of --
    // leaving its project accumulators, raw, in d.partial [ks][n][Ho Wo][Cout]; mb_reduce_partials adds the ksplit parts in order.
    // (ksplit is a power of two: shifts, not the 40-instruction expansion of a scalar division in every workgroup's set-up)
#ifndef BH_MB_KSPLIT
#define BH_MB_KSPLIT 1      // (0: the channel split compiled out -- A/B aid for what its set-up arithmetic costs the large launches)
#endif
    const int ksp = (BH_MB_KSPLIT && !SE && PERSIST == 0 && d.ksplit > 1) ? d.ksplit : 1, ks = ksp > 1 ? (int)blockIdx.y : 0;
    const int ksl = ksp >= 8 ? 3 : ksp >= 4 ? 2 : ksp >= 2 ? 1 : 0;
    const int c0 = (ks * d.nchunks) >> ksl, c1 = ((ks + 1) * d.nchunks) >> ksl;
    const int per_xcd = (n_tiles + 7) >> 3;
    const int tile_xcd = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (!RESIDENT && (dbgv & 256 ? (int)blockIdx.x >= n_tiles : tile_xcd >= n_tiles)) return;   // (dbg 256: launch order, A/B aid)
    const int tile_first = RESIDENT ? (int)blockIdx.x : (dbgv & 256 ? (int)blockIdx.x : tile_xcd);
#ifdef BIRDA_HIP_EXPERIMENTS
    MbClock t_last{};
    if (d.stamps) t_last.last = __builtin_readcyclecounter();
#endif
    if constexpr (RESIDENT) {
        for (int c = 0; c < d.nchunks; c++) {   // every chunk's weights, once
            mb_dma<WE_FLOATS, NW>(d.We + (size_t)c * WE_FLOATS, WeS + c * WE_FLOATS, wave0, lane0);
            mb_dma<WD_FLOATS, NW>(d.Wd + (size_t)c * WD_FLOATS, Wds + c * WD_FLOATS, wave0, lane0);
            mb_dma<WP_FLOATS, NW>(d.Wp + (size_t)c * WP_FLOATS, WpS + c * WP_FLOATS, wave0, lane0);
        }
    } else {
        if constexpr (COLTH > 0) {
            mb_dma_at<WE_FLOATS, NW>(d.We + (size_t)c0 * WE_FLOATS, we_ba, wave0, lane0);
            mb_dma_at<WD_FLOATS, NW>(d.Wd + (size_t)c0 * WD_FLOATS, wd_ba, wave0, lane0);
        } else {
            mb_dma<WE_FLOATS, NW>(d.We + (size_t)c0 * WE_FLOATS, WeS, wave0, lane0);
            mb_dma<WD_FLOATS, NW>(d.Wd + (size_t)c0 * WD_FLOATS, Wds, wave0, lane0);
        }
        if (ring) {
            mb_dma<WP_FLOATS, NW>(d.Wp, WpS, wave0, lane0);
            if (d.nchunks > 1) {
                mb_dma<WE_FLOATS, NW>(d.We + WE_FLOATS, WeS + WE_FLOATS, wave0, lane0);
                mb_dma<WD_FLOATS, NW>(d.Wd + WD_FLOATS, Wds + WD_FLOATS, wave0, lane0);
                mb_dma<WP_FLOATS, NW>(d.Wp + WP_FLOATS, WpS + WP_FLOATS, wave0, lane0);
            }
        }
    }

    // (STRIP: `tile` stays the strip's index and `trow` walks its tile rows)
    const int n_trows = STRIP ? d.tiles_y : 1;
    for (int tile = tile_first; tile < (RESIDENT ? n_tiles : tile_first + 1); tile += RESIDENT ? (int)gridDim.x : 1)
    for (int trow = 0; trow < n_trows; trow++) {
    // (everything below is per tile.  The thread index goes through an opaque copy so that hipcc does not hoist the
    //  tile-invariant index arithmetic out of the tile loop and keep it in registers across it: a first persistent
    //  version doubled its VGPRs and spilled SGPRs that way)
    int tid = tid0;
    if constexpr (PERSIST != 0) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
#if BH_MB_HOSTRCP
    // q = floor(n / dv) as mulhi(n, floor(2^32 / dv) + 1): the estimate exceeds the quotient by n e / (dv 2^32) < 1 (e = the
    // multiplier's rounding, at most dv), i.e. by at most one for any 32-bit n -- one fix-up step either way makes it exact (and
    // serves dv = 1, whose multiplier saturates one short)
    auto udiv = [](unsigned nn, unsigned dv, unsigned magic) {
        unsigned q = __umulhi(nn, magic);
        const int rem = (int)(nn - q * dv);
        q += (rem >= (int)dv ? 1 : 0) - (rem < 0 ? 1 : 0);
        return (int)q;
    };
    const int tz = udiv((unsigned)tile, (unsigned)tiles_xy, STRIP ? d.rcp_tiles_x : d.rcp_tiles_xy), txy = tile - tz * tiles_xy;
    const int tyi = STRIP ? trow : udiv((unsigned)txy, (unsigned)d.tiles_x, d.rcp_tiles_x), txi = STRIP ? txy : txy - tyi * d.tiles_x;
#else
    const int tz = tile / tiles_xy, txy = tile - tz * tiles_xy;
    const int tyi = STRIP ? trow : txy / d.tiles_x, txi = STRIP ? txy : txy - tyi * d.tiles_x;
#endif
    const int seg0 = tz * SS;
    const int nsv = min(SS, n_seg - seg0);
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * ST - d.pad_t, ix0 = ox0 * ST - d.pad_l;
    const int ya = max(0, -iy0), yb = min(IH, d.H - iy0);
    const int xa = max(0, -ix0), xb = min(IW, d.W - ix0);
    // STRIP, below the first tile row: grid rows [0, KH) come from the halo store; the expand phase starts at row ra = KH
    const int ra = (STRIP && trow > 0) ? max(ya, KH) : ya;
    const int vh = max(yb - ra, 0), vw = max(xb - xa, 0);
    const int Mseg = vh * vw, M = Mseg * nsv, nrt = (M + 15) >> 4;
    const int Cin = d.Cin, Cout = d.Cout, nchunks = (dbgv & 128) ? 0 : d.nchunks;
    // STEM: X is the planar spectrogram [n][C][SH][SW]; "Cin" = kh*kw*C im2col columns
    const float *Xb = STEM ? d.X + (size_t)seg0 * d.stem_c * d.stem_h * d.stem_w : d.X + (size_t)seg0 * d.H * d.W * Cin;
    const int rw = wave / NCS, cs = wave - rw * NCS;  // P1: row-tile lane of the wave, column split
    const int wm = wave / WN, wn = wave - wm * WN;    // P3

    const float rcp_vw = 1.0f / (float)max(vw, 1);
    const float e_unscale = d.e_unscale, p_scale = d.p_scale, p_unscale = d.p_unscale;
    const GeluScaled gelu_sc = d.gelu;
    // ---- the wave's rows of X: A fragments of the expand GEMM, resident for the whole kernel ----
    // f32: afr[i][g] = 4 k values of one 16-deep group; f16: ah/al[i][g] = 8 k values of one 32-deep step
    // KG == 0: a block WITHOUT an expand convolution (EfficientNet's expand-ratio-1 blocks after the first: depthwise -> project
    // + residual).  Its "expanded" tensor is the block input itself: P1 copies the chunk's channels of X into the LDS grid (no
    // GEMM, no activation), everything else is the same kernel.
    constexpr int KGA = KG > 0 ? KG : 1;
    float4 afr[PREC ? 1 : RT_W][PREC ? 1 : KGA];
    f16x8 ah[PREC ? RT_W : 1][PREC ? KGA : 1], al[PREC ? RT_W : 1][PREC ? KGA : 1];
    int xrow[KG == 0 ? RT_W : 1];   // KG == 0: float offset of the lane's source row in X (-1: padding row)
    // the expand GEMM is computed transposed (E^T = We^T X^T): a lane ends up with 4 consecutive
    // channels of ONE source row, li of its row tile, and writes them with one ds_write_b128 at
    // eoff[i] = that row's slot in the LDS grid (padding rows: the trash slot)
    int eoff[RT_W];
    // Two passes over the wave's row tiles: every load of a batch of row tiles is issued UNCONDITIONALLY from a clamped address
    // (rows past the tile, columns past Cin and out-of-image taps read a valid address and are replaced by zero afterwards), and
    // only then are the values split / stored.  With the loads inside `if (valid)` hipcc waited for each row tile's loads before
    // computing the next one's addresses: RT_W dependent HBM round trips at the start of every workgroup.
    constexpr int NV = PREC ? 8 : 4;            // consecutive k per lane and step
    constexpr int NQ = STEM ? 2 : NV / 4;       // 16-byte (stem, f16 modes: 12-byte) loads per (row tile, step)
    // (8-wave kernels: one workgroup per CU, so nothing hides a second round trip of the set-up, and at that point the 256 registers
    //  hold nothing but these loads and the fragments they become: all row tiles in one batch)
    constexpr int XLIM = NW == 8 ? 24 : 16;
    constexpr int XBATCH = (STEM && !PREC) ? 1 : (RT_W * KGA * NQ <= XLIM ? RT_W : (XLIM / (KGA * NQ) >= 1 ? XLIM / (KGA * NQ) : 1));
    struct __attribute__((packed, aligned(4))) F3 { float a, b, c; };
#pragma unroll
    for (int i0 = 0; i0 < RT_W; i0 += XBATCH) {
        int xo[XBATCH];
        bool rvv[XBATCH];
#pragma unroll
        for (int ii = 0; ii < XBATCH; ii++) {
            const int i = i0 + ii;
            if (i >= RT_W) continue;
            const int rt = rw + RSTEP * i;
            rvv[ii] = false;
            xo[ii] = 0;   // row offset computed in registers: the loads go out before the table barrier
            eoff[i] = egrid * CES + 4 * kq;
            const int m = rt * 16 + li;
            if (rt < nrt && m < M) {
                const int sl = SS > 1 ? (m >= Mseg ? 1 : 0) : 0, mm = m - sl * Mseg;
#if BH_MB_CONSTDIV
                // (vw == IWC, wave-uniform: the tile's rectangle is inside the image in x -- every tile but the last column's, and
                //  the first's under left padding; IWC is a compile-time number, so the division is a multiply and a shift)
                constexpr int IWC = ((1 << TWL) - 1) * ST + KS;
                // (exact while row index x the rounding error of the 16-bit reciprocal stays below one: every shipped entry; a tile as wide
                //  as the EXPERIMENTS build's 65-column ones divides as before)
                constexpr bool kConstDiv = (long)(SS * 64 * 16) * IWC < 65536L;
                const int r = kConstDiv && vw == IWC ? (int)(((unsigned)mm * (unsigned)((65536 + IWC - 1) / IWC)) >> 16) : mb_div(mm, vw, rcp_vw), c = mm - __mul24(r, vw);
#else
                const int r = mb_div(mm, vw, rcp_vw), c = mm - __mul24(r, vw);
#endif
                eoff[i] = __mul24(sl * IH * IW + __mul24(ra + r, IW) + xa + c, CES) + 4 * kq;
                xo[ii] = STEM ? (((iy0 + ra + r) << 16) | (ix0 + xa + c))   // stem-output pixel (y, x), gathered below
                              : __mul24(__mul24(sl * d.H + iy0 + ra + r, d.W) + ix0 + xa + c, Cin);
                rvv[ii] = !(dbgv & 64);
            }
        }
        if constexpr (KG == 0) {
#pragma unroll
            for (int ii = 0; ii < XBATCH; ii++)
                if (i0 + ii < RT_W) xrow[i0 + ii < RT_W ? i0 + ii : 0] = rvv[ii] ? xo[ii] : -1;
        } else
        if constexpr (STEM != 0 && PREC == 0) {
            // f32 mode: im2col column k = (dy * 3 + dx) * C + ch, exactly the row order of the [kh][kw][cin][cout] weights
            const int i = i0;
            const int sy = (xo[0] >> 16) * d.stem_s - d.stem_pt, sx = (xo[0] & 0xffff) * d.stem_s - d.stem_pl;
#pragma unroll
            for (int g = 0; g < KG; g++) {
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int k = 16 * g + 4 * kq + c;
                    const int tap = k / (STEM ? STEM : 1), ch = k - tap * STEM;   // STEM = spectrogram channels
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const int y = sy + dy, x = sx + dx;
                    const bool ok = rvv[0] && k < Cin && y >= 0 && y < d.stem_h && x >= 0 && x < d.stem_w;
                    v[c] = ok ? Xb[((size_t)ch * d.stem_h + y) * d.stem_w + x] : 0.0f;
                }
                afr[i][g] = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else if constexpr (STEM != 0) {
            // im2col gather of the 3x3 stem conv (stride s, planar input), f16 modes: ONE 32-deep step whose columns are ordered
            // by memory runs -- lane group kq holds runs 2 kq and 2 kq + 1, a run being the three horizontally adjacent taps
            // (dx = 0, 1, 2) of one (channel, dy); elements 6, 7 of the group are zero (api.hip plan_fusion packs the weight rows
            // to match).  Two 12-byte loads per lane and row tile instead of eight scalar ones with a division chain per
            // element: the gather was the stem block's largest single cost (tools/abl.sh).
            // (more than two spectrogram channels -- a front-end of three mel branches: 9 runs -- take a second step: run 8 g + 2 kq + q)
            static_assert(STEM == 0 || PREC == 0 || 3 * STEM <= 8 * KG, "3 STEM runs fit the 8 run slots of each step");
            F3 t[XBATCH][KG][2];
            bool okr[XBATCH][KG][2];
#pragma unroll
            for (int ii = 0; ii < XBATCH; ii++) {
                if (i0 + ii >= RT_W) continue;
                // (24-bit multiplies and 32-bit element offsets: v_mul_lo_u32 / v_mad_u64_u32 are quarter-rate, and a segment's
                //  spectrogram or image is far below 2^24 elements -- mb_try_th checks it)
                const int sy = __mul24(xo[ii] >> 16, d.stem_s) - d.stem_pt, sx = __mul24(xo[ii] & 0xffff, d.stem_s) - d.stem_pl;
                const int bx = min(max(sx, 0), d.stem_w - 3);   // the 3-float window, kept inside the row
#pragma unroll
                for (int g = 0; g < KG; g++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int r = 8 * g + 2 * kq + q, ch = STEM <= 2 ? (r >= 3 ? 1 : 0) : r / 3, dy = r - 3 * ch, y = sy + dy;
                    okr[ii][g][q] = rvv[ii] && r < 3 * STEM && y >= 0 && y < d.stem_h;
                    const int yc = okr[ii][g][q] ? y : 0, chc = okr[ii][g][q] ? ch : 0;
                    t[ii][g][q] = *reinterpret_cast<const F3 *>(Xb + (__mul24(__mul24(chc, d.stem_h) + yc, d.stem_w) + bx));
                }
            }
#pragma unroll
            for (int ii = 0; ii < XBATCH; ii++) {
                const int i = i0 + ii;
                if (i >= RT_W) continue;
                const int sx = __mul24(xo[ii] & 0xffff, d.stem_s) - d.stem_pl;
                const int sh = sx - min(max(sx, 0), d.stem_w - 3);
#pragma unroll
                for (int g = 0; g < KG; g++) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    // element dx is column sx + dx = bx + sh + dx; sh is -1 / 0 / +1 at the left edge / inside / at the right edge
                    const F3 tt = t[ii][g][q];
                    const float e0 = sh == 0 ? tt.a : sh > 0 ? tt.b : 0.0f;
                    const float e1 = sh == 0 ? tt.b : sh > 0 ? tt.c : tt.a;
                    const float e2 = sh == 0 ? tt.c : sh > 0 ? 0.0f : tt.b;
                    v[3 * q] = (okr[ii][g][q] && sx >= 0) ? e0 : 0.0f;
                    v[3 * q + 1] = (okr[ii][g][q] && sx + 1 >= 0 && sx + 1 < d.stem_w) ? e1 : 0.0f;
                    v[3 * q + 2] = (okr[ii][g][q] && sx + 2 < d.stem_w) ? e2 : 0.0f;
                }
                bh_split8(v, ah[i][g], al[i][g]);
                }
            }
        } else {
            float4 raw[XBATCH][KG][NQ];
            // (EXPERIMENTS build, ONE instantiation -- entry 15 on the f32 MFMA, 3x3 stride 2 with three k groups: with the phase clock and the
            //  ablation bits compiled in, this compiler (ROCm 7.2 clang) emits "V_CMP_NE_U32 0, src_shared_base" for it, an illegal
            //  instruction, whenever its X rows go through the buffer descriptor; found by bisection over the table, no source construct to
            //  point at.  That instantiation of that build loads X through pointers as in round 5: the same values.)
#if defined(BIRDA_HIP_EXPERIMENTS)
            constexpr bool kBufLoad = BH_MB_BUFLOAD != 0 && !(KS == 3 && ST == 2 && CE == 16 && KG == 3 && RT_W == 7 && PREC == 0);
#else
            constexpr bool kBufLoad = BH_MB_BUFLOAD != 0;
#endif
#if BH_MB_BUFLOAD
            // the workgroup's segments of X as ONE buffer: a lane that must read zero takes offset 2^32 - 1 and the range check answers 0
            const auto xrs = __builtin_amdgcn_make_buffer_rsrc(mb_global(Xb), 0, __builtin_amdgcn_readfirstlane(__mul24(nsv * d.H, d.W) * Cin * 4), 0x00020000);
#endif
#pragma unroll
            for (int ii = 0; ii < XBATCH; ii++) {
                if (i0 + ii >= RT_W) continue;
#pragma unroll
                for (int g = 0; g < KG; g++)
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const int kk = (PREC ? 32 * g + 8 * kq : 16 * g + 4 * kq) + 4 * q;
                        const bool ok = rvv[ii] && kk < Cin;
#if BH_MB_BUFLOAD
                        if constexpr (kBufLoad)
                            raw[ii][g][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? (unsigned)(xo[ii] + kk) * 4u : 0xffffffffu, 0, 0));
                        else
#endif
                        raw[ii][g][q] = *reinterpret_cast<const float4 *>(Xb + (ok ? xo[ii] + kk : 0));
                    }
            }
#pragma unroll
            for (int ii = 0; ii < XBATCH; ii++) {
                const int i = i0 + ii;
                if (i >= RT_W) continue;
#pragma unroll
                for (int g = 0; g < KG; g++) {
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const int kk = (PREC ? 32 * g + 8 * kq : 16 * g + 4 * kq) + 4 * q;
                        const bool ok = rvv[ii] && kk < Cin;
                        const float4 t = raw[ii][g][q];
                        if constexpr (kBufLoad) {
                            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;     // (zeros came back from the range check)
                        } else {
                            v[4 * q] = ok ? t.x : 0.0f; v[4 * q + 1] = ok ? t.y : 0.0f;
                            v[4 * q + 2] = ok ? t.z : 0.0f; v[4 * q + 3] = ok ? t.w : 0.0f;
                        }
                    }
                    if constexpr (PREC != 0) bh_split8(v, ah[i][g], al[i][g]);
                    else afr[i][g] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    for (int p = tid; p < POUT_PAD; p += NTH) {
        const int sl = (p >= THTW ? 1 : 0) + (p >= 2 * THTW ? 1 : 0), pp = p - sl * THTW;  // SS <= 2
        const int ty = pp >> TWL, tx = pp & (TW - 1);
        int o = -1;
        if (sl < nsv && oy0 + ty < d.Ho && ox0 + tx < d.Wo) o = (sl * d.Ho + oy0 + ty) * d.Wo + ox0 + tx;
        omap[p] = o;
    }
    if constexpr (STRIP) {
        // the whole grid once, at the top of the strip (image columns outside the strip's rectangle and the rows above the image
        // then stay zero: nothing writes them); further down only the rows below the image -- they hold the previous tile's values
        float4 *z = reinterpret_cast<float4 *>(Es);
        const int z0 = trow == 0 ? 0 : yb * IW * CES / 4, n4 = egrid * CES / 4;   // (IW * CES is a multiple of 4)
        if (trow == 0 || yb < IH)
            for (int i = z0 + tid; i < n4; i += NTH) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else
    if (M != egrid) {  // some of the grid lies outside the image (or a segment is missing): zero padding
        float4 *z = reinterpret_cast<float4 *>(Es);
        const int n4 = egrid * CES / 4;
        for (int i = tid; i < n4; i += NTH) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    mb_dma_wait();
    __syncthreads();

    mb_stamp(d.stamps, t_last, 0);

    // The project accumulators start at bias + residual: those loads overlap the first chunk
    // instead of stalling the epilogue, which is then nothing but stores.
    // (channel split: the raw accumulators go to this part's slab of d.partial; bias and residual start part 0 only)
    float *Yb = (ksp > 1 ? d.partial + (size_t)ks * n_seg * d.Ho * d.Wo * Cout : d.Y) + (size_t)seg0 * d.Ho * d.Wo * Cout;
    const float *Rb = (d.R && ks == 0) ? d.R + (size_t)seg0 * d.Ho * d.Wo * Cout : nullptr;
    const float bias_on = ks == 0 ? 1.0f : 0.0f, y_unscale = ksp > 1 ? 1.0f : p_unscale;
    f32x4 acco[MT_W][NT_W];
#if BH_MB_BUFSTORE
    // the workgroup's segments of Y (and of the residual, shaped alike) as buffers
    const unsigned y_bytes = (unsigned)__builtin_amdgcn_readfirstlane(__mul24(nsv * d.Ho, d.Wo) * Cout * 4);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(mb_global(Yb), 0, SE ? 0 : y_bytes, 0x00020000);
    const auto yrs_r = __builtin_amdgcn_make_buffer_rsrc(mb_global(Rb ? Rb : Yb), 0, (SE || !Rb) ? 0 : y_bytes, 0x00020000);
#endif
    // SE (pass A of a squeeze-excite block): where the depthwise output and the tile's channel sums go
    const bool se_store = SE && d.Dout != nullptr;      // (nullptr: sums only -- the no-expand blocks, whose D is computed again by the gated one-launch block)
    float *Dg = SE ? (se_store ? d.Dout : reinterpret_cast<float *>(size_t(1) << 30)) + (size_t)seg0 * d.Ho * d.Wo * d.Cexp : nullptr;   // (never dereferenced without se_store)
    // SE store phase: thread (channel quad tid % C4N, pixel tid / C4N + k NTH / C4N) -- the same pixels in every chunk, so their rows
    // of D are looked up once per tile (nullptr: outside the image or the batch)
    constexpr int SE_NP = SE ? (POUT_PAD + NTH / C4N - 1) / (NTH / C4N) : 1;
    float *se_dst[SE_NP];
    const int npix_pad = nsv * THTW;
    const int se_cstr = SE && d.dblk ? CE * 16 : CE;      // floats from one chunk's channels to the next's in D
    if constexpr (SE) {
#pragma unroll
        for (int k = 0; k < SE_NP; k++) {
            const int p0 = tid / C4N + k * (NTH / C4N);
            const int o = p0 < npix_pad ? omap[p0] : -1;
            // (dblk: [row tile][Cexp / 16][16 rows][16 channels] -- a segment's pixels are whole tiles, kernels.hpp MbDesc::dblk; the
            //  thread's own channel quad is folded in here, the chunk adds a uniform ch * se_cstr)
            const int c4s = tid % C4N;
            se_dst[k] = o < 0 ? nullptr : d.dblk ? Dg + (size_t)(o >> 4) * 16 * d.Cexp + (o & 15) * 16 + ((c4s >> 2) << 8) + 4 * (c4s & 3)
                                                 : Dg + (size_t)o * d.Cexp + 4 * c4s;
        }
    }
    if constexpr (!SE)
#pragma unroll
    for (int i = 0; i < MT_W; i++) {
        const int4 o4 = *reinterpret_cast<const int4 *>(&omap[(wm * MT_W + i) * 16 + 4 * kq]);
        const int orow[4] = {o4.x, o4.y, o4.z, o4.w};
        // (unconditional loads from clamped addresses, selected afterwards: inside `cond ? load : 0` every load waited for the
        //  one before it)
#pragma unroll
        for (int j = 0; j < NT_W; j++) {
            const int col = (wn * NT_W + j) * 16 + li, colc = min(col, Cout - 1);
            if constexpr (MT_W * NT_W <= 8) {   // (the small-tile instantiations of the early blocks; with 30 accumulator tiles
                                                //  that many loads in flight cost registers the late blocks do not have)
                const float braw = d.bp[colc];
                float rres[4] = {0.f, 0.f, 0.f, 0.f};
                const float bias = col < Cout ? braw * bias_on : 0.0f;
#if BH_MB_BUFSTORE
                if (Rb) {   // (wave-uniform) lanes without a pixel or past Cout: offset 2^32 - 1, the range check answers 0
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        rres[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs_r, (col < Cout && orow[r] >= 0) ? (unsigned)(__mul24(orow[r], Cout) + col) * 4u : 0xffffffffu, 0, 0));
                }
                // (the scale through an opaque vector register: with the kernel argument's SGPR pair as operand hipcc packs these four
                //  FMAs as v_pk_fma_f32 ... op_sel:[0,1,0] -- a half-selecting packed-f32 form; tests/test_abi_and_host.py
                //  ::test_device_code_has_no_half_swapped_packed_f32_ops keeps every such form out of the library)
                float ps_v = p_scale;
                asm volatile("" : "+v"(ps_v));
#pragma unroll
                for (int r = 0; r < 4; r++) acco[i][j][r] = __builtin_fmaf(rres[r], ps_v, bias);
#else
                if (Rb) {   // (wave-uniform)
#pragma unroll
                    for (int r = 0; r < 4; r++) rres[r] = Rb[__mul24(max(orow[r], 0), Cout) + colc];
                }
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acco[i][j][r] = __builtin_fmaf((col < Cout && orow[r] >= 0) ? rres[r] : 0.0f, p_scale, bias);
#endif
            } else {
                const float bias = col < Cout ? d.bp[col] * bias_on : 0.0f;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acco[i][j][r] = __builtin_fmaf((Rb && col < Cout && orow[r] >= 0) ? Rb[__mul24(orow[r], Cout) + col] : 0.0f, p_scale, bias);
            }
        }
    }

    // P2 tasks: XB output pixels x 4 channels each.  The 16-channel-chunk instantiations have at most 256
    // = one per thread, so the first task's decomposition is done once here, not once per chunk.
    const int p2_ntask = nsv * TH * XBN * C4N;
    int p2_c4, p2_eoff, p2_prow;
    auto p2_task = [&](int t) {
        const int q = t / C4N;
        p2_c4 = t - q * C4N;
        const int r = q >> (TWL - XBL), sl = (SS > 1 && r >= TH) ? 1 : 0, ty = r - sl * TH;
        const int tx0 = (q & (XBN - 1)) * XB;
        p2_eoff = (sl * IH * IW + (ty * ST) * IW + tx0 * ST) * CES + 4 * p2_c4;
        p2_prow = sl * THTW + (ty << TWL) + tx0;
    };
    p2_task(tid);

    // SE: the channel sums of chunk `se_pending` wait in LDS as partial sums per wave and row of 16 lanes (WpS: [slot][wave][row][CE]);
    // thread (slot, channel) adds them in (wave, row) order and stores the tile's sum
    int se_pending = -1;
    auto se_flush = [&]() {
        if (se_pending >= 0 && tid < SS * CE) {
            const int sl = tid / CE, c = tid - sl * CE;
            if (sl < nsv && se_pending * CE + c < d.Cexp) {
                float sum = 0.0f;
#pragma unroll
                for (int w = 0; w < NW * 4; w++) sum += WpS[(sl * NW * 4 + w) * CE + c];
                d.pool_part[((size_t)(seg0 + sl) * tiles_xy + txy) * d.Cexp + se_pending * CE + c] = sum;
            }
        }
        se_pending = -1;
    };
    for (int ch = (dbgv & 128) ? nchunks : c0; ch < ((dbgv & 128) ? nchunks : c1); ch++) {
        // (STRIP: behind the last chunk comes chunk 0 of the next tile row -- its weights arrive under this tile's last phases)
        const int chn = STRIP ? (ch + 1 < nchunks ? ch + 1 : 0) : min(ch + 1, nchunks - 1);
        // this chunk's weights
        const float *WeC = WeS + (RESIDENT ? ch : ring ? (ch & 1) : 0) * WE_FLOATS, *WdC = Wds + (RESIDENT ? ch : ring ? (ch & 1) : 0) * WD_FLOATS;
        const float *WpC = WpS + (RESIDENT ? ch : ring ? (ch % 3) : 0) * WP_FLOATS;
        if constexpr (STRIP) {
            // the chunk's rows of the tile above: halo store -> grid rows [0, KH)  (one float4 per thread)
            if (trow > 0) {
                const float4 *hsrc = reinterpret_cast<const float4 *>(Hs + (size_t)ch * halo_fl);
                float4 *hdst = reinterpret_cast<float4 *>(Es);
                for (int i = tid; i < halo_fl / 4; i += NTH) hdst[i] = hsrc[i];
            }
        }
        const float *bes = WeC + KG * NT_E * FRAG, *bds = WdC + KS * KS * CE;
        mb_stamp(d.stamps, t_last, 1);

        // ---- P1: expand ------------------------------------------------------------------
        if constexpr (KG == 0) {
            // no expand convolution: the chunk's CE channels of the block input go into the grid as they are.  Lane (li, kq) of row
            // tile i moves channels 16 j + 4 kq .. + 3 of its row: one 16-byte load and one ds_write_b128, the slots the GEMM's
            // epilogue would have written.
            static_assert(KG != 0 || NCS == 1, "no-expand blocks: no column split");
            static_assert(KG != 0 || COLTH == 0, "no-expand blocks: row tasks (the gate of MbDesc::gate is applied in the row-task depthwise phase only)");
            static_assert(STEM == 0 || COLTH == 0, "stem blocks: row tasks (they take MbDesc::gate too)");
            float4 xv[RT_W][NT_E];
#pragma unroll
            for (int i = 0; i < RT_W; i++)
#pragma unroll
                for (int j = 0; j < NT_E; j++) {
                    const int c0 = ch * CE + 16 * j + 4 * kq;
                    const bool ok = xrow[i] >= 0 && c0 < Cin;     // (Cin % 4 == 0: a quad is wholly inside or outside)
                    xv[i][j] = *reinterpret_cast<const float4 *>(Xb + (ok ? xrow[i] + c0 : 0));
                    if (!ok) xv[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int i = 0; i < RT_W; i++)
                if (rw + RSTEP * i < nrt) {
#pragma unroll
                    for (int j = 0; j < NT_E; j++) *reinterpret_cast<float4 *>(Es + eoff[i] + j * 16) = xv[i][j];
                }
        } else
#pragma unroll
        for (int i0 = 0; i0 < RT_W; i0 += RG) {
            if (rw + RSTEP * i0 < nrt) {  // wave-uniform
                f32x4 acc[RG][NT_U];   // [channel 4 kq + r][source row li], seeded with the bias
#pragma unroll
                for (int j = 0; j < NT_U; j++) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(&bes[(cs * NT_U + j) * 16 + 4 * kq]);
#pragma unroll
                    for (int ii = 0; ii < RG; ii++) acc[ii][j] = b4;
                }
                if (!(dbgv & 8)) {
                    if constexpr (PREC != 0) {
                        // fragment planes: [step][column tile]{hi: 64 lanes x 8 halves, lo: same}
                        const f16x8 *wf = reinterpret_cast<const f16x8 *>(WeC);
                        // the next step's weight fragments are read from LDS BEFORE this step's MFMAs are issued (two register
                        // sets, alternating): with the loads right in front of their MFMAs every k step of a late block
                        // (6 steps x 9 MFMAs) began with an exposed LDS round trip, ~120 of its ~260 cycles
                        f16x8 bhb[2][NT_U], blb[2][NT_U];
                        auto wload = [&](int g, f16x8 (&h)[NT_U], f16x8 (&l)[NT_U]) __attribute__((always_inline)) {
#pragma unroll
                            for (int j = 0; j < NT_U; j++) {
                                h[j] = wf[((g * NT_E + cs * NT_U + j) * 2 + 0) * 64 + lane];
                                if (PREC == 3) l[j] = wf[((g * NT_E + cs * NT_U + j) * 2 + 1) * 64 + lane];
                            }
                        };
                        wload(0, bhb[0], blb[0]);
#pragma unroll
                        for (int g = 0; g < KG; g++) {
                            // (KG >= 6 only, the 192-channel blocks: measured -5 % there, nothing at 3-4 steps, and the early
                            //  blocks' one-step kernels lost 1 % to the changed register allocation)
                            constexpr bool AHEAD = KG >= 6;
                            if (AHEAD && g + 1 < KG) wload(g + 1, bhb[(g + 1) & 1], blb[(g + 1) & 1]);
                            if (!AHEAD && g > 0) wload(g, bhb[g & 1], blb[g & 1]);
                            f16x8 (&bh)[NT_U] = bhb[g & 1], (&bl)[NT_U] = blb[g & 1];
#pragma unroll
                            for (int ii = 0; ii < RG; ii++) {
                                if (i0 + ii >= RT_W) continue;
                                constexpr int dummy = 0; (void)dummy;
                                const int ir = i0 + ii < RT_W ? i0 + ii : 0;
#pragma unroll
                                for (int j = 0; j < NT_U; j++) {
                                    acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[ir][g], acc[ii][j], 0, 0, 0);
                                    if (PREC == 3) {
                                        acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[ir][g], acc[ii][j], 0, 0, 0);
                                        acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[ir][g], acc[ii][j], 0, 0, 0);
                                    }
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                    // B fragments double-buffered by hand, and a scheduling fence per k group: left
                    // alone, hipcc hoists every group's LDS loads to the top of the unrolled loop and
                    // spills the resident A fragments
                    float4 bv[NT_U], bn[NT_U];
#pragma unroll
                    for (int j = 0; j < NT_U; j++)
                        bv[j] = *reinterpret_cast<const float4 *>(&WeC[((cs * NT_U + j) * 64 + lane) * 4]);
#pragma unroll
                    for (int g = 0; g < KG; g++) {
                        if (g + 1 < KG) {
#pragma unroll
                            for (int j = 0; j < NT_U; j++)
                                bn[j] = *reinterpret_cast<const float4 *>(&WeC[(((g + 1) * NT_E + cs * NT_U + j) * 64 + lane) * 4]);
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++)
#pragma unroll
                            for (int ii = 0; ii < RG; ii++) {
                                if (i0 + ii >= RT_W) continue;
                                const float4 av = afr[i0 + ii < RT_W ? i0 + ii : 0][g];
                                const float a = c == 0 ? av.x : c == 1 ? av.y : c == 2 ? av.z : av.w;
#pragma unroll
                                for (int j = 0; j < NT_U; j++) {
                                    const float b = c == 0 ? bv[j].x : c == 1 ? bv[j].y : c == 2 ? bv[j].z : bv[j].w;
                                    acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[ii][j], 0, 0, 0);
                                }
                            }
                        if (g + 1 < KG) {
#pragma unroll
                            for (int j = 0; j < NT_U; j++) bv[j] = bn[j];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    }  // PREC == 0
                }
#pragma unroll
                for (int ii = 0; ii < RG; ii++) {
                    if (i0 + ii >= RT_W) continue;
                    const int rt = rw + RSTEP * (i0 + ii);
                    if (rt < nrt) {
                        float *erow = Es + eoff[i0 + ii < RT_W ? i0 + ii : 0] + cs * NT_U * 16;
#pragma unroll
                        for (int j = 0; j < NT_U; j++) {
                            f32x2 v01 = {acc[ii][j][0], acc[ii][j][1]}, v23 = {acc[ii][j][2], acc[ii][j][3]};
                            if constexpr (PREC != 0 && ACT == ACT_GELU_ERF && BH_GELU_DEGREE == 5) {
                                // weights and bias carry 2^se: the GELU runs on the scaled value and leaves 2^(se + 1) GELU(x) in the grid;
                                // the depthwise taps carry the 2^-(se + 1) (kernels.hpp gelu2x_fast4, api.hip plan_fusion)
                                if constexpr (BH_GELU_2X != 0) gelu2x_fast4(v01, v23, gelu_sc);
                                else gelu_erf_fast4_scaled(v01, v23, gelu_sc);
                            } else {
                                if constexpr (PREC != 0) { v01 *= e_unscale; v23 *= e_unscale; }   // weights and bias carry 2^se
                                bh_act4<ACT>(v01, v23);
                            }
                            *reinterpret_cast<f32x4 *>(erow + j * 16) = (f32x4){v01[0], v01[1], v23[0], v23[1]};
                        }
                    }
                }
            }
        }
        mb_stamp(d.stamps, t_last, 2);
        if (!ring && !RESIDENT) mb_dma_wait();
        __syncthreads();  // B1: Es complete; (ring == 0) WeS / WpS free; Wds (DMA issued after the last B2) landed
        if constexpr (SE) se_flush();   // the previous chunk's wave sums are all in LDS now (their next writer sits behind B2)
        if constexpr (STRIP) {
            // ... and this chunk's bottom rows for the tile below: grid rows [TH * ST, TH * ST + KH) -> halo store.  (Read-only in
            // this phase; the store's next reader is this chunk's P1 phase of the next tile row, many barriers from here.)
            if (trow + 1 < n_trows) {
                const float4 *gsrc = reinterpret_cast<const float4 *>(Es + (size_t)TH * ST * IW * CES);
                float4 *hdst = reinterpret_cast<float4 *>(Hs + (size_t)ch * halo_fl);
                for (int i = tid; i < halo_fl / 4; i += NTH) hdst[i] = gsrc[i];
            }
        }
        if (!ring && !RESIDENT && !(dbgv & 16)) {
            if constexpr (COLTH > 0) {
                // (issued in parts inside the depthwise phase below)
            } else {
                mb_dma<WE_FLOATS, NW>(d.We + (size_t)chn * WE_FLOATS, WeS, wave, lane);
                if constexpr (!SE) mb_dma<WP_FLOATS, NW>(d.Wp + (size_t)ch * WP_FLOATS, WpS, wave, lane);
            }
        }
        mb_stamp(d.stamps, t_last, 3);

        // ---- P2: depthwise, XB output pixels x 4 channels per lane ---------------------------
        // (SE: every variant leaves the chunk's depthwise output in the LDS D buffer as f32 rows [pixel][CE + 4], the f32 mode's layout)
        if constexpr (COLTH > 0 && NW == 8) {
            // column tasks of the 8-wave workgroups: one lane = one output column (all COLTH rows) x TWO channels, so that
            // SS * TW * CE / 2 = 512 tasks fill the workgroup and a task's rows (COLTH x KS float2) fit beside the resident A
            // fragments in the 256 registers two waves per SIMD leave; rows are finished two at a time (two interleaved GELU
            // chains per lane, as bh_act4 does for the 4-channel tasks)
            constexpr int PADT = (KS - 1) / 2, C2N = CE / 2;
            static_assert(ST == 1, "column tasks: stride 1");
            const bool dma_on = !(dbgv & 16);
            if (!(tid < nsv * TW * C2N && !(dbgv & 2))) {   // (wave-uniform)
                if (dma_on) {
                    mb_dma_at<WE_FLOATS, NW>(d.We + (size_t)chn * WE_FLOATS, we_ba, wave, lane);
                    if constexpr (!SE) mb_dma_at<WP_FLOATS, NW>(d.Wp + (size_t)ch * WP_FLOATS, wp_ba, wave, lane);
                }
            } else {
                const int c2 = tid % C2N, q = tid / C2N, x = q & (TW - 1), sl = q >> TWL;
                const float *eb = Es + ((sl * IH + PADT) * IW + x) * CES + 2 * c2;   // grid row PADT = image row 0
                // one grid COLUMN of the task's window at a time (the next one in flight): all COLTH x KS values at once are 60
                // registers under a 5x5 kernel on a 6-row image, which the 256 of a wave here do not have beside the A fragments
                f32x2 e[2][COLTH];
#pragma unroll
                for (int r = 0; r < COLTH; r++) e[0][r] = *reinterpret_cast<const f32x2 *>(eb + (r * IW) * CES);
                const f32x2 bd2 = *reinterpret_cast<const f32x2 *>(&bds[2 * c2]);
                f32x2 acc[COLTH];
#pragma unroll
                for (int r = 0; r < COLTH; r++) acc[r] = bd2;
#pragma unroll
                for (int dx = 0; dx < KS; dx++) {
                    if (dx + 1 < KS) {
#pragma unroll
                        for (int r = 0; r < COLTH; r++) e[(dx + 1) & 1][r] = *reinterpret_cast<const f32x2 *>(eb + (r * IW + dx + 1) * CES);
                    }
#pragma unroll
                    for (int dy = 0; dy < KS; dy++) {
                        const f32x2 w = *reinterpret_cast<const f32x2 *>(&WdC[(dy * KS + dx) * CE + 2 * c2]);
#pragma unroll
                        for (int r = 0; r < COLTH; r++) {
                            constexpr int dummy = 0; (void)dummy;
                            const int src = r + dy - PADT;          // image row under this tap; outside [0, COLTH): zero padding
                            if (src < 0 || src >= COLTH) continue;   // (compile-time after unrolling)
                            acc[r] = __builtin_elementwise_fma(e[dx & 1][src], w, acc[r]);
                        }
                    }
                    // this column's share of the next chunk's expand weights and this chunk's project weights.  (All of it behind the first
                    // one, two or three columns instead, so that no piece is issued right in front of the barrier that waits for it:
                    // measured, no difference -- 6.16 / 6.19 / 6.17 / 6.19 us per segment over all blocks, three alternations.)
                    if (dma_on) {
                        mb_dma_at_part<WE_FLOATS, NW>(d.We + (size_t)chn * WE_FLOATS, we_ba, wave, lane, dx, KS);
                        if constexpr (!SE) mb_dma_at_part<WP_FLOATS, NW>(d.Wp + (size_t)ch * WP_FLOATS, wp_ba, wave, lane, dx, KS);
                    }
                    // the sums are pinned here: left alone, hipcc sinks this column's FMAs below the next columns' loads (three
                    // columns of the window and their taps in flight: +60 registers, which the 136- and 232-channel blocks of the
                    // Perch-sized stack spilled -- 642 -> 366 us and 679 -> 408 us per 1 000 segments for the pin alone)
#pragma unroll
                    for (int r = 0; r < COLTH; r++) asm volatile("" : "+v"(acc[r]) : : "memory");
                    __builtin_amdgcn_sched_barrier(0);   // one window column of loads in flight
                }
#pragma unroll
                for (int r = 0; r < COLTH; r += 2) {
                    f32x2 g0 = acc[r], g1 = acc[r + 1 < COLTH ? r + 1 : r];
                    if (r + 1 < COLTH) mb_act4<ACT, PREC>(g0, g1);
                    else g0 = mb_act2<ACT, PREC>(g0);
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (r + k >= COLTH) continue;
                        const f32x2 g = k ? g1 : g0;
                        const int prow = sl * THTW + ((r + k) << TWL) + x;
                        if constexpr (PREC == 3 && !SE) {
                            bh_f16x2 h, l;
                            bh_split2(g[0], g[1], h, l);
                            *reinterpret_cast<bh_f16x2 *>(&DsH[prow * DSH + 2 * c2]) = h;
                            *reinterpret_cast<bh_f16x2 *>(&DsL[prow * DSH + 2 * c2]) = l;
                        } else if constexpr (PREC == 1 && !SE) {
                            *reinterpret_cast<bh_f16x2 *>(&DsH[prow * DSH + 2 * c2]) = (bh_f16x2){(_Float16)g[0], (_Float16)g[1]};
                        } else {
                            *reinterpret_cast<f32x2 *>(&Ds[prow * CES + 2 * c2]) = g;
                        }
                    }
                }
            }
        } else
        if constexpr (COLTH > 0) {
            // column tasks (see COLTH above): task = (segment slot, column x, channel group c4); mb_try_th guarantees TH == H ==
            // Ho == COLTH, one tile row, pad_t == (KS - 1) / 2 and at most 256 tasks
            constexpr int PADT = (KS - 1) / 2;
            static_assert(ST == 1, "column tasks: stride 1");
            const bool dma_on = !(dbgv & 16);
            if (!(tid < nsv * TW * C4N && !(dbgv & 2))) {   // (wave-uniform: a wave without tasks issues its pieces at once)
                if (dma_on) {
                    mb_dma_at<WE_FLOATS, NW>(d.We + (size_t)chn * WE_FLOATS, we_ba, wave, lane);
                    if constexpr (!SE) mb_dma_at<WP_FLOATS, NW>(d.Wp + (size_t)ch * WP_FLOATS, wp_ba, wave, lane);
                }
            } else {
                const int c4 = tid % C4N, q = tid / C4N, x = q & (TW - 1), sl = q >> TWL;
                const float *eb = Es + ((sl * IH + PADT) * IW + x) * CES + 4 * c4;   // grid row PADT = image row 0
                const float4 bd4 = *reinterpret_cast<const float4 *>(&bds[4 * c4]);
                f32x2 acc[COLTH][2];
#pragma unroll
                for (int r = 0; r < COLTH; r++) { acc[r][0] = (f32x2){bd4.x, bd4.y}; acc[r][1] = (f32x2){bd4.z, bd4.w}; }
                float4 e[COLTH][KS];
#pragma unroll
                for (int r = 0; r < COLTH; r++)
#pragma unroll
                    for (int dx = 0; dx < KS; dx++) e[r][dx] = *reinterpret_cast<const float4 *>(eb + (r * IW + dx) * CES);
#pragma unroll
                for (int dy = 0; dy < KS; dy++) {
#pragma unroll
                    for (int dx = 0; dx < KS; dx++) {
                        const float4 w = *reinterpret_cast<const float4 *>(&WdC[(dy * KS + dx) * CE + 4 * c4]);
                        const f32x2 w0 = (f32x2){w.x, w.y}, w1 = (f32x2){w.z, w.w};
#pragma unroll
                        for (int r = 0; r < COLTH; r++) {
                            constexpr int dummy = 0; (void)dummy;
                            const int src = r + dy - PADT;          // image row under this tap; outside [0, COLTH): zero padding
                            if (src < 0 || src >= COLTH) continue;   // (compile-time after unrolling)
                            const float4 ev = e[src][dx];
                            acc[r][0] = __builtin_elementwise_fma((f32x2){ev.x, ev.y}, w0, acc[r][0]);
                            acc[r][1] = __builtin_elementwise_fma((f32x2){ev.z, ev.w}, w1, acc[r][1]);
                        }
                    }
                    if (dma_on) {   // this row's share of the next chunk's expand weights and this chunk's project weights
                        mb_dma_at_part<WE_FLOATS, NW>(d.We + (size_t)chn * WE_FLOATS, we_ba, wave, lane, dy, KS);
                        if constexpr (!SE) mb_dma_at_part<WP_FLOATS, NW>(d.Wp + (size_t)ch * WP_FLOATS, wp_ba, wave, lane, dy, KS);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one kernel row of weight loads in flight
                }
#pragma unroll
                for (int r = 0; r < COLTH; r++) {
                    f32x2 g0 = acc[r][0], g1 = acc[r][1];
                    mb_act4<ACT, PREC>(g0, g1);
                    const int prow = sl * THTW + (r << TWL) + x;
                    if constexpr (PREC == 3 && !SE) {
                        bh_f16x2 h0, l0, h1, l1;
                        bh_split2(g0[0], g0[1], h0, l0);
                        bh_split2(g1[0], g1[1], h1, l1);
                        *reinterpret_cast<f16x4 *>(&DsH[prow * DSH + 4 * c4]) = (f16x4){h0[0], h0[1], h1[0], h1[1]};
                        *reinterpret_cast<f16x4 *>(&DsL[prow * DSH + 4 * c4]) = (f16x4){l0[0], l0[1], l1[0], l1[1]};
                    } else if constexpr (PREC == 1 && !SE) {
                        *reinterpret_cast<f16x4 *>(&DsH[prow * DSH + 4 * c4]) =
                            (f16x4){(_Float16)g0[0], (_Float16)g0[1], (_Float16)g1[0], (_Float16)g1[1]};
                    } else {
                        *reinterpret_cast<float4 *>(&Ds[prow * CES + 4 * c4]) = make_float4(g0[0], g0[1], g1[0], g1[1]);
                    }
                }
            }
        } else
        if (!(dbgv & 2)) {
            for (int t = tid; t < p2_ntask; t += NTH) {
                if (p2_ntask > NTH) p2_task(t);   // wave-uniform
                const int c4 = p2_c4;
                const float *eb = Es + p2_eoff;
                const float4 bd4 = *reinterpret_cast<const float4 *>(&bds[4 * c4]);
                f32x2 gq0 = (f32x2){1.f, 1.f}, gq1 = gq0;     // (KG = 0, se = 0, gate: the squeeze-excite gate of a no-expand block, MbDesc::gate)
                // (round 6: the STEM block takes the gate here too -- its D, 40 channels at the stem's output size, is ten times its input, the
                //  planar spectrogram: cheaper computed twice than kept, like the no-expand blocks)
                if constexpr ((KG == 0 || STEM != 0) && !SE) {
                    if (d.gate) {
                        const int sl = (SS > 1 && p2_prow >= THTW) ? 1 : 0, cgq = ch * CE + 4 * c4;
                        const float4 gv = *reinterpret_cast<const float4 *>(d.gate + (size_t)min(seg0 + sl, n_seg - 1) * d.Cexp + min(cgq, d.Cexp - 4));
                        if (cgq < d.Cexp) { gq0 = (f32x2){gv.x, gv.y}; gq1 = (f32x2){gv.z, gv.w}; }
                    }
                }
                // two-wide vectors so the taps become v_pk_fma_f32 (2 FMAs per instruction)
                f32x2 acc[XB][2];
#pragma unroll
                for (int x = 0; x < XB; x++) { acc[x][0] = (f32x2){bd4.x, bd4.y}; acc[x][1] = (f32x2){bd4.z, bd4.w}; }
#pragma unroll
                for (int dy = 0; dy < KS; dy++) {
                    float4 e[NCOL];
#pragma unroll
                    for (int j = 0; j < NCOL; j++) e[j] = *reinterpret_cast<const float4 *>(eb + (dy * IW + j) * CES);
#pragma unroll
                    for (int dx = 0; dx < KS; dx++) {
                        const float4 w = *reinterpret_cast<const float4 *>(&WdC[(dy * KS + dx) * CE + 4 * c4]);
                        const f32x2 w0 = (f32x2){w.x, w.y}, w1 = (f32x2){w.z, w.w};
#pragma unroll
                        for (int x = 0; x < XB; x++) {
                            const float4 ev = e[x * ST + dx];
                            acc[x][0] = __builtin_elementwise_fma((f32x2){ev.x, ev.y}, w0, acc[x][0]);
                            acc[x][1] = __builtin_elementwise_fma((f32x2){ev.z, ev.w}, w1, acc[x][1]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep one kernel row of loads in flight, not all KS
                }
#pragma unroll
                for (int x = 0; x < XB; x++) {
                    f32x2 g0 = acc[x][0], g1 = acc[x][1];
                    mb_act4<ACT, PREC>(g0, g1);
                    if constexpr ((KG == 0 || STEM != 0) && !SE) { g0 *= gq0; g1 *= gq1; }
                    const float4 v = make_float4(g0[0], g0[1], g1[0], g1[1]);
                    const int prow = p2_prow + x;
                    if constexpr (PREC != 0 && !SE) {   // the project GEMM's A operand: f16 hi (+ lo) planes
                        if (PREC == 3) {
                            bh_f16x2 h0, l0, h1, l1;
                            bh_split2(v.x, v.y, h0, l0);
                            bh_split2(v.z, v.w, h1, l1);
                            *reinterpret_cast<f16x4 *>(&DsH[prow * DSH + 4 * c4]) = (f16x4){h0[0], h0[1], h1[0], h1[1]};
                            *reinterpret_cast<f16x4 *>(&DsL[prow * DSH + 4 * c4]) = (f16x4){l0[0], l0[1], l1[0], l1[1]};
                        } else {
                            f16x4 h;
                            h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
                            *reinterpret_cast<f16x4 *>(&DsH[prow * DSH + 4 * c4]) = h;
                        }
                    } else {
                        *reinterpret_cast<float4 *>(&Ds[prow * CES + 4 * c4]) = v;
                    }
                }
            }
        }
        mb_stamp(d.stamps, t_last, 4);
        if (!RESIDENT) mb_dma_wait();    // ring: the next chunk's weights, on their way since the previous B2
        __syncthreads();  // B2: Ds complete; (ring == 0) WeS (next chunk) and WpS (this chunk) landed; Wds free
        if (RESIDENT) {
        } else if (!ring) {
            if (!(dbgv & 16)) {
                if constexpr (COLTH > 0) mb_dma_at<WD_FLOATS, NW>(d.Wd + (size_t)chn * WD_FLOATS, wd_ba, wave, lane);
                else mb_dma<WD_FLOATS, NW>(d.Wd + (size_t)chn * WD_FLOATS, Wds, wave, lane);
            }
        } else if (ch + 2 < nchunks && !(dbgv & 16)) {
            // chunk ch + 2 into the buffers chunk ch has just finished with (We, Wd: read before this barrier) and into the Wp
            // buffer of chunk ch - 1 (its project phase ended before B1 of this chunk)
            mb_dma<WE_FLOATS, NW>(d.We + (size_t)(ch + 2) * WE_FLOATS, WeS + (ch & 1) * WE_FLOATS, wave, lane);
            mb_dma<WD_FLOATS, NW>(d.Wd + (size_t)(ch + 2) * WD_FLOATS, Wds + (ch & 1) * WD_FLOATS, wave, lane);
            mb_dma<WP_FLOATS, NW>(d.Wp + (size_t)(ch + 2) * WP_FLOATS, WpS + ((ch + 2) % 3) * WP_FLOATS, wave, lane);
        }
        mb_stamp(d.stamps, t_last, 5);

        if constexpr (SE) {
            // ---- S: the chunk's depthwise output leaves for HBM, and its channel sums are taken on the way ----------------------
            // Thread = (channel quad c4 = tid % C4N, pixel p = tid / C4N, stepping by NTH / C4N): a pixel's CE channels are one
            // contiguous run in LDS and in D (64 or 128 bytes).  Sums: per thread over its pixels (one accumulator per segment
            // slot), then across the lanes of a ROW of 16 that share the quad (row rotations folded into the adds: a fixed tree, no
            // atomics -- identical segments give identical bits wherever they sit in a batch); lanes 0 .. C4N - 1 of every row leave
            // the row's sums in LDS (the project weights' buffer, unused in this pass), and after the NEXT barrier one thread per
            // channel adds the 4 NW of them up.  (A first version: xor-shuffles over the whole wave, 16 ds_bpermute per thread and
            // chunk, and the D rows looked up per chunk: pass A of the early blocks 7-33 % SLOWER than the whole gate-free block.)
            static_assert(16 % C4N == 0 && NTH % C4N == 0, "squeeze-excite pass A: the channel quads of a chunk divide a row of 16 lanes");
            const int c4 = tid % C4N, cg = ch * CE + 4 * c4;
            const int cgo = ch * se_cstr;                                   // the chunk's offset from the thread's place in its pixel of D
            f32x4 ssum[SS];
#pragma unroll
            for (int q = 0; q < SS; q++) ssum[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < SE_NP; k++) {
                const int p0 = tid / C4N + k * (NTH / C4N);
                if (p0 < npix_pad) {      // (uniform but for the last pass)
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(&Ds[p0 * CES + 4 * c4]);
                    if (se_dst[k]) {
                        if (se_store && cg < d.Cexp) *reinterpret_cast<f32x4 *>(se_dst[k] + cgo) = v;
                        if (SS == 1 || p0 < THTW) ssum[0] += v; else ssum[SS - 1] += v;
                    }
                }
            }
            // lanes of a ROW of 16 that share the quad: rotations within the row, folded into the add (DPP: no LDS traffic)
#pragma unroll
            for (int q = 0; q < SS; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float x = ssum[q][j];
                    if constexpr (C4N <= 4) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));   // row_ror:4
                    if constexpr (C4N <= 8) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));   // row_ror:8
                    ssum[q][j] = x;
                }
            if ((lane & 15) < C4N) {
#pragma unroll
                for (int q = 0; q < SS; q++) *reinterpret_cast<f32x4 *>(&WpS[(((q * NW + wave) * 4 + (lane >> 4)) * C4N + (lane & 15)) * 4]) = ssum[q];
            }
            se_pending = ch;
        } else
        // ---- P3: project -----------------------------------------------------------------
        if (!(dbgv & 4)) {
            if constexpr (P16) {
                // fragment planes: [column tile]{hi: 64 lanes x 4 halves, lo: same}; k = 4 (lane >> 4) + 0..3
                const f16x4 *wf = reinterpret_cast<const f16x4 *>(WpC);
                f16x4 a_h[MT_W], a_l[MT_W], b_h[NT_W], b_l[NT_W];
#pragma unroll
                for (int i = 0; i < MT_W; i++) {
                    const int row = (wm * MT_W + i) * 16 + li;
                    a_h[i] = *reinterpret_cast<const f16x4 *>(&DsH[row * DSH + 4 * kq]);
                    if (PREC == 3) a_l[i] = *reinterpret_cast<const f16x4 *>(&DsL[row * DSH + 4 * kq]);
                }
#pragma unroll
                for (int j = 0; j < NT_W; j++) {
                    b_h[j] = wf[((wn * NT_W + j) * 2 + 0) * 64 + lane];
                    if (PREC == 3) b_l[j] = wf[((wn * NT_W + j) * 2 + 1) * 64 + lane];
                }
#pragma unroll
                for (int i = 0; i < MT_W; i++)
#pragma unroll
                    for (int j = 0; j < NT_W; j++) {
                        acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_h[i], b_h[j], acco[i][j], 0, 0, 0);
                        if (PREC == 3) {
                            acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_h[i], b_l[j], acco[i][j], 0, 0, 0);
                            acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a_l[i], b_h[j], acco[i][j], 0, 0, 0);
                        }
                    }
            } else if constexpr (PREC != 0) {
                const f16x8 *wf = reinterpret_cast<const f16x8 *>(WpC);
                // the weight fragments of a step are taken JB column tiles at a time: all NT_W (up to 10 hi +
                // 10 lo quads) at once cost 80 registers and spilled the 320-channel blocks
                constexpr int JB = NT_W <= (NW == 8 ? 4 : 6) ? NT_W : (NT_W + 1) / 2;   // (8 waves: 256 registers each)
#pragma unroll
                for (int g = 0; g < PSTEPS; g++) {
                    f16x8 a_h[MT_W], a_l[MT_W];
#pragma unroll
                    for (int i = 0; i < MT_W; i++) {
                        const int row = (wm * MT_W + i) * 16 + li;
                        a_h[i] = *reinterpret_cast<const f16x8 *>(&DsH[row * DSH + 32 * g + 8 * kq]);
                        if (PREC == 3) a_l[i] = *reinterpret_cast<const f16x8 *>(&DsL[row * DSH + 32 * g + 8 * kq]);
                    }
#pragma unroll
                    for (int j0 = 0; j0 < NT_W; j0 += JB) {
                        f16x8 b_h[JB], b_l[JB];
#pragma unroll
                        for (int jj = 0; jj < JB; jj++) {
                            if (j0 + jj >= NT_W) continue;
                            b_h[jj] = wf[((g * NTOP + wn * NT_W + j0 + jj) * 2 + 0) * 64 + lane];
                            if (PREC == 3) b_l[jj] = wf[((g * NTOP + wn * NT_W + j0 + jj) * 2 + 1) * 64 + lane];
                        }
#pragma unroll
                        for (int i = 0; i < MT_W; i++)
#pragma unroll
                            for (int jj = 0; jj < JB; jj++) {
                                if (j0 + jj >= NT_W) continue;
                                const int j = j0 + jj < NT_W ? j0 + jj : 0;
                                acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[i], b_h[jj], acco[i][j], 0, 0, 0);
                                if (PREC == 3) {
                                    acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[i], b_l[jj], acco[i][j], 0, 0, 0);
                                    acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[i], b_h[jj], acco[i][j], 0, 0, 0);
                                }
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
            const float *dsb = Ds + ((wm * MT_W) * 16 + li) * CES + 4 * kq;
            float4 a[MT_W], b[NT_W], an[MT_W], bn[NT_W];
#pragma unroll
            for (int j = 0; j < NT_W; j++) b[j] = *reinterpret_cast<const float4 *>(&WpC[((wn * NT_W + j) * 64 + lane) * 4]);
#pragma unroll
            for (int i = 0; i < MT_W; i++) a[i] = *reinterpret_cast<const float4 *>(dsb + i * 16 * CES);
#pragma unroll
            for (int g = 0; g < NT_E; g++) {
                if (g + 1 < NT_E) {
#pragma unroll
                    for (int j = 0; j < NT_W; j++)
                        bn[j] = *reinterpret_cast<const float4 *>(&WpC[(((g + 1) * NTOP + wn * NT_W + j) * 64 + lane) * 4]);
#pragma unroll
                    for (int i = 0; i < MT_W; i++) an[i] = *reinterpret_cast<const float4 *>(dsb + i * 16 * CES + 16 * (g + 1));
                }
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int i = 0; i < MT_W; i++) {
                        const float av = c == 0 ? a[i].x : c == 1 ? a[i].y : c == 2 ? a[i].z : a[i].w;
#pragma unroll
                        for (int j = 0; j < NT_W; j++) {
                            const float bv = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
                            acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acco[i][j], 0, 0, 0);
                        }
                    }
                if (g + 1 < NT_E) {
#pragma unroll
                    for (int j = 0; j < NT_W; j++) b[j] = bn[j];
#pragma unroll
                    for (int i = 0; i < MT_W; i++) a[i] = an[i];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }  // PREC == 0
        }
        mb_stamp(d.stamps, t_last, 6);
        // no barrier here: the next chunk's P1 touches Es / WeS (landed before B2) only; its P2
        // (which rewrites Ds) sits behind B1, which also drains the Wd DMA issued above.
    }

    if constexpr (SE) {
        __syncthreads();
        se_flush();
    }
    // ---- epilogue: store (bias and residual are already in the accumulators) -------------------
    if constexpr (!SE)
#pragma unroll
    for (int i = 0; i < MT_W; i++) {
        const int4 o4 = *reinterpret_cast<const int4 *>(&omap[(wm * MT_W + i) * 16 + 4 * kq]);
        const int orow[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int j = 0; j < NT_W; j++) {
            const int col = (wn * NT_W + j) * 16 + li;
#if BH_MB_BUFSTORE
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float yv = PREC != 0 ? acco[i][j][r] * y_unscale : acco[i][j][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), yrs, (col < Cout && orow[r] >= 0 && !(dbgv & 32)) ? (unsigned)(__mul24(orow[r], Cout) + col) * 4u : 0xffffffffu, 0, 0);
            }
#else
            if (col >= Cout) continue;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (orow[r] >= 0 && !(dbgv & 32)) Yb[__mul24(orow[r], Cout) + col] = PREC != 0 ? acco[i][j][r] * y_unscale : acco[i][j][r];
#endif
        }
    }
    mb_stamp(d.stamps, t_last, 7);
    if constexpr (PERSIST != 0) __syncthreads();   // the epilogue has read omap; the next tile's set-up rewrites it
    }   // tiles
#ifdef BIRDA_HIP_EXPERIMENTS
    if (d.stamps && lane0 == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&d.stamps[i], t_last.acc[i]);
#endif
}

template <int KS, int ST, int CE, int KG, int RT_W, int NCS, int WM, int WN, int MT_W, int NT_W, int TWL, int XBL,
          int SS, int OCC, int STEM, int PREC, int PERSIST, int ACT, int COLTH = 0, int SE = 0>
void mb_launch(const MbDesc &d, int n_seg, hipStream_t s) {
    auto kern = mbconv_kernel<KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, SS, OCC, STEM, PREC, PERSIST, ACT, COLTH, SE>;
    static DeviceOnce attr_set;
    attr_set.run([&] { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    // (PERSIST == 2, strip-walking: one workgroup per tile COLUMN and segment)
    const long n_wg = (long)d.tiles_x * (PERSIST == 2 ? 1 : d.tiles_y) * ((n_seg + d.S - 1) / d.S);
    dim3 grid((unsigned)(((n_wg + 7) / 8) * 8)), block(64 * WM * WN);   // one-dimensional: the kernel deals the tiles XCD by XCD
    if (PERSIST == 1) {   // as many workgroups as are resident at once: registers allow OCC per SIMD, LDS 160 KB per CU
        const long total = (long)d.tiles_x * d.tiles_y * ((n_seg + d.S - 1) / d.S);
        const long per_cu = std::max<long>(1, std::min<long>(OCC, (160 * 1024) / (long)(d.lds_bytes + 256)));
        grid = dim3((unsigned)std::min<long>(total, per_cu * device_cu_count()));
    }
    if (!SE && PERSIST == 0 && d.ksplit > 1) grid.y = (unsigned)d.ksplit;   // (channel split: gridDim.x stays a multiple of 8, so workgroup (x, y) still lands on XCD x % 8)
    hipLaunchKernelGGL(kern, grid, block, d.lds_bytes, s, d, n_seg);
}

// (MB_A: the activation the table is being expanded for, see kCfgs below)
// MB_WITH_SE (defined by the translation unit of the activation squeeze-excite stacks use: swish, EfficientNet's): every product entry
// is also instantiated as pass A of a squeeze-excite block (SE = 1); elsewhere the slot is nullptr and such blocks run layer by layer
// (pass A sums the tile's channels with a wave-level tree: the channel quads of a chunk must divide a wave -- the entries that do
//  not qualify keep a nullptr and such blocks run layer by layer)
template <int KS, int ST, int CE, int KG, int RT_W, int NCS, int WM, int WN, int MT_W, int NT_W, int TWL, int XBL,
          int SS, int OCC, int STEM, int PREC, int PERSIST, int ACT, int COLTH>
constexpr auto mb_se_fn() -> void (*)(const MbDesc &, int, hipStream_t) {
    constexpr int NW = WM * WN, C4N = CE / 4;
    // (the wave sums of a chunk, [SS][NW][CE] floats, wait in the project weights' LDS buffer, which this pass does not fill)
    constexpr int WPF = (PREC != 0 && CE == 16) ? WN * NT_W * 256 : (PREC ? (CE + 31) / 32 : CE / 16) * WN * NT_W * (PREC ? 512 : 256);
    if constexpr (16 % C4N == 0 && (64 * NW) % C4N == 0 && SS * NW * 4 * CE <= WPF)
        return mb_launch<KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, SS, OCC, STEM, PREC, PERSIST, ACT, COLTH, 1>;
    else
        return nullptr;
}
#ifdef MB_WITH_SE
#define MB_SE_FN(...) mb_se_fn<__VA_ARGS__>()
#else
#define MB_SE_FN(...) nullptr
#endif
#define MB_ENTRY_P(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, PREC) \
    {KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, TH, S, STEM, XBL, OCC, PREC, 0, MB_A, 0,     \
     mb_launch<KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, S, OCC, STEM, PREC, 0, MB_A>,  \
     MB_SE_FN(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, S, OCC, STEM, PREC, 0, MB_A, 0)}
// column-task depthwise phase (COLTH = TH = the image height), split-f16 and plain-f16 twins
#define MB_ENTRY_PC(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, PREC) \
    {KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, TH, S, STEM, XBL, OCC, PREC, 0, MB_A, TH,    \
     mb_launch<KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, S, OCC, STEM, PREC, 0, MB_A, TH>, \
     MB_SE_FN(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, S, OCC, STEM, PREC, 0, MB_A, TH)}
#define MB_ENTRY_HC(KS, ST, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM)            \
    MB_ENTRY_PC(KS, ST, 32, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, 3),         \
    MB_ENTRY_PC(KS, ST, 32, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, 1)
// persistent workgroups, every chunk's weights resident in LDS (the early blocks)
#define MB_ENTRY_PP(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, PREC) \
    {KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, TH, S, STEM, XBL, OCC, PREC, 1, MB_A, 0,      \
     mb_launch<KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, S, OCC, STEM, PREC, 1, MB_A>, nullptr}
// strip-walking workgroups (PERSIST = 2): the early blocks on large batches
#define MB_ENTRY_PW(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, PREC) \
    {KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, TH, S, STEM, XBL, OCC, PREC, 2, MB_A, 0,      \
     mb_launch<KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, S, OCC, STEM, PREC, 2, MB_A>, nullptr}
#define MB_ENTRY_S(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM) \
    MB_ENTRY_P(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, 0)
#define MB_ENTRY(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC) \
    MB_ENTRY_S(KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, 0)
// split-f16 (PREC 3) and plain-f16 (PREC 1) twins of one tile configuration; KG counts 32-deep steps
#define MB_ENTRY_H(KS, ST, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM)             \
    MB_ENTRY_P(KS, ST, 32, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, 3),          \
    MB_ENTRY_P(KS, ST, 32, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, XBL, TH, S, OCC, STEM, 1)
// A configuration that is not part of this build (measured alternatives live behind -DBIRDA_HIP_EXPERIMENTS, see mbconv_cfgs.inc):
// keeps its index, matches no block (KS = 0).
#define MB_NONE {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, MB_A, 0, nullptr, nullptr}
#define MB_NONE2 MB_NONE, MB_NONE
#ifdef BIRDA_HIP_EXPERIMENTS
#define MB_XENTRY(...) MB_ENTRY(__VA_ARGS__)
#define MB_XENTRY_S(...) MB_ENTRY_S(__VA_ARGS__)
#define MB_XENTRY_P(...) MB_ENTRY_P(__VA_ARGS__)
#define MB_XENTRY_PP(...) MB_ENTRY_PP(__VA_ARGS__)
#define MB_XENTRY_PW(...) MB_ENTRY_PW(__VA_ARGS__)
#define MB_XENTRY_H(...) MB_ENTRY_H(__VA_ARGS__)
#define MB_XENTRY_HC(...) MB_ENTRY_HC(__VA_ARGS__)
#define MB_XENTRY_PC(...) MB_ENTRY_PC(__VA_ARGS__)
#else
#define MB_XENTRY(...) MB_NONE
#define MB_XENTRY_S(...) MB_NONE
#define MB_XENTRY_P(...) MB_NONE
#define MB_XENTRY_PP(...) MB_NONE
#define MB_XENTRY_PW(...) MB_NONE
#define MB_XENTRY_H(...) MB_NONE2
#define MB_XENTRY_HC(...) MB_NONE2
#define MB_XENTRY_PC(...) MB_NONE
#endif

}  // namespace
}  // namespace bh
