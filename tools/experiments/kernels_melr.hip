// EXPERIMENT, not part of the product build (round 3).  Measured on MI355X, 1 000 segments, f16x3: 1.04 us per segment against
// mel_kernel's 0.68 -- parity-green (all front-end tests pass with it), but slower.  Why: with the operator out of the way the
// bound moves to building the MFMA B fragments: 16 lanes of a fragment are 16 FRAMES, i.e. LDS reads at a stride of the hop.
// An item needs 2 K 32 frames 4 B = 262 KB of such reads per 4 608 MFMA cycles = 57 B/clk/CU; ds_read_b32 delivers 64 B/clk/CU
// conflict-free (hop 278) and 16 B/clk at the 4-way conflicts of hop 280 (280 = 24 mod 32: sixteen consecutive frames fall on four
// banks whatever permutation of k the fragment uses), the folded pair x[tH + k + 1] + x[tH + L - 1 - k] has one odd start whatever
// the grouping (no aligned ds_read_b64 / b128 for both), and with ONE wave per SIMD (512 registers for the operator) nothing
// overlaps those reads with the MFMAs unless the loop is software-pipelined by hand.  The branch-1 workgroups (hop 280) are the
// critical path: 8 192 LDS cycles against 2 304 MFMA cycles per item.  What would have to change: a skewed staging layout
// (effective hop 282) written through registers instead of LDS-DMA, plus a hand-interleaved main loop.  DESIGN.md section 3.
//
// melr_kernel: the folded STFT x mel GEMM of the BirdNET front-end with the OPERATOR STATIONARY IN REGISTERS.
//
// (Same mathematics as kernels_frontend.hip: per segment and branch spec_t = Gf^T y_t, y_t[j] = x[tH + j + 1] + x[tH + L - 1 - j],
//  on the split-f16 MFMA; reference: the STFT / mel nodes of the ONNX graph behind birdnet_onnx::Classifier::predict_batch,
//  src/inference/classifier.rs:478-488, SURVEY.md Appendix B.)
//
// mel_kernel streams the operator: every 48-frame item pulls all 590 KB of Gf (hi + lo f16 planes of both branches) from L2 --
// 6.5 MB per segment, 9.7 TB/s at 0.67 us per segment, i.e. the kernel is bound by the L2 -> CU fabric (16-18 TB/s chip-wide),
// not by HBM (1.2 TB/s) nor by the MFMA (its main loop runs at half the matrix rate).  Here the operator never moves:
//
//   * one workgroup of 4 waves per CU, 512 registers per wave.  Wave w keeps ITS k slice of one branch's operator for ALL 96 mels
//     in registers for the whole launch: branch 0 (K = 1024) = 8 steps x 6 mel tiles x {hi, lo} x 4 = 384 registers per wave
//     (the four waves together hold the branch's 393 KB), branch 1 (K = 512) = 192.
//   * workgroups come in groups of 8 on ONE XCD (workgroup b runs on XCD b % 8): five hold branch 0 ("A"), three branch 1 ("B")
//     -- the branches cost 2 : 1 in MFMAs and 5 : 3 with the per-item overhead.  A pass of a group = a quarter (4 tiles of 32
//     frames) of 5 segments: A_i takes the four branch-0 items of segment 5Q + i, the B's share the twenty branch-1 items in
//     tile-major order, so both branches of a (segment, tile) are worked on within a tile's time of each other and the second
//     read of its sample span hits that XCD's L2.
//   * an item = (segment, branch, 32-frame tile).  Its RAW sample span goes global -> LDS by LDS-DMA (no registers; the next
//     item's span lands in the second buffer under the current item's main loop).  Each wave builds the folded frames of its own
//     k slice as MFMA B fragments (two LDS reads, one v_add_f32, one FMA for the min / max normalisation -- applied to the SUM:
//     (x1 + x2) sc + nb -- and the f16 hi / lo split), runs 3 MFMAs per product against its resident A fragments, parks its
//     12 partial tiles in LDS; wave w then sums and finishes tiles w, w + 4, w + 8 (square, power law, affine, flip, store).
//
// Per segment: 452 MFLOP of f16 MFMAs (0.19 us at the matrix peak), no operator traffic, 1.2x the algorithmic HBM bytes.
// Built for the BirdNET-v2.4 front-end (two branches, 96 mels, K = 1024 / 512, 16 tiles of 32 frames); launch_mel falls back to
// mel_kernel / mel32_kernel for anything else.
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace bh {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int MR_MT = 6;          // mel tiles (96 mels)
constexpr int MR_NT = 2;          // 16-frame tiles per item
constexpr int MR_FR = 16 * MR_NT; // frames per item
constexpr int MR_TQ = 4;          // items (tiles) per quarter pass
constexpr int MR_GSEG = 5;        // segments per pass = A workgroups per group
constexpr int MR_NB = 3;          // B workgroups per group
constexpr int MR_MM = 8;          // min / max partial blocks per segment (minmax_kernel)

struct MrItem { int sg, tl, ok; };

// One item on one workgroup: KS = 32-deep k steps per wave (8: branch 0, 4: branch 1).  oph / opl: the wave's resident operator.
template <int KS>
__device__ __forceinline__ void mr_item(const float *__restrict__ x, const float *__restrict__ mm, float *__restrict__ spec,
                                        const BranchParams &bp, const int branch, const int n_branches, const int S, const long n_total,
                                        const float norm_eps, const MrItem it, const MrItem nx, float *buf_cur, float *buf_nxt,
                                        float4 *red, const f16x8 (&oph)[KS][MR_MT], const f16x8 (&opl)[KS][MR_MT],
                                        const int tid, const int dbg) {
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, kq = lane >> 4;
    const int L = bp.L, H = bp.H;
    const int span_pieces = ((MR_FR - 1) * H + L + 255) >> 8;   // 1-KiB pieces of the item's sample span
    // the normalisation of this item's segment: x <- 2((x - min) / (max - min + eps) - 0.5) = (x - min) sc - 1; on a folded pair
    // (x1 + x2) sc + nb with nb = -2 (min sc + 1)
    const float4 *mv = reinterpret_cast<const float4 *>(mm + (size_t)it.sg * MR_MM * 2);
    const float4 m0 = mv[0], m1 = mv[1], m2 = mv[2], m3 = mv[3];
    // the span of this item was issued one item ago (or in the prologue): it must have landed, and every wave must be done with
    // the previous item's reduction buffer and with the buffer the next span is about to overwrite
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nx.ok) {   // the next item's span: global -> LDS, 1 KiB per wave instruction, no registers
        const long base = (long)nx.sg * S + (long)nx.tl * MR_FR * H;
        for (int p = wave; p < span_pieces; p += 4) {
            long idx = base + p * 256 + lane * 4;
            idx = idx < n_total - 4 ? idx : n_total - 4;   // (past the batch: a valid address; only frames that are never stored read it)
            const unsigned la = (unsigned)(size_t)(buf_nxt + p * 256);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(x + idx), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
        }
    }
    const float mn = fminf(fminf(fminf(m0.x, m0.z), fminf(m1.x, m1.z)), fminf(fminf(m2.x, m2.z), fminf(m3.x, m3.z)));
    const float mx = fmaxf(fmaxf(fmaxf(m0.y, m0.w), fmaxf(m1.y, m1.w)), fmaxf(fmaxf(m2.y, m2.w), fmaxf(m3.y, m3.w)));
    const float sc = 2.0f / ((mx - mn) + norm_eps);
    const float nb = -2.0f * __builtin_fmaf(mn, sc, 1.0f);

    f32x4 acc[MR_NT][MR_MT];
#pragma unroll
    for (int f = 0; f < MR_NT; f++)
#pragma unroll
        for (int m = 0; m < MR_MT; m++) acc[f][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop: the wave's k slice [wave KS 32, (wave + 1) KS 32) ----------------------------------------------------
    // element jj of a lane's B fragment is k = 32 st + 4 jj + kq (the operator planes are packed to match, api.hip build_gf):
    // the four lane groups read NEIGHBOURING samples, which spreads a frame-strided access over the banks (mel_kernel)
    const float *xf = buf_cur + li * H;
    if (!(dbg & 1)) {
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int j0 = (wave * KS + s) * 32 + kq;
            f16x8 bhv[MR_NT], blv[MR_NT];
#pragma unroll
            for (int f = 0; f < MR_NT; f++) {
                float y[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    // (one v_add_f32, on purpose: the packed form with swapped halves is not safe beside in-flight f16 MFMAs,
                    //  kernels.hpp bh_add_unpacked)
                    const float sum = (dbg & 4) ? sc * (float)jj : bh_add_unpacked(xf[f * 16 * H + j0 + 4 * jj + 1], xf[f * 16 * H + L - 1 - j0 - 4 * jj]);
                    y[jj] = __builtin_fmaf(sum, sc, nb);
                }
                bh_split8(y, bhv[f], blv[f]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MR_MT; m++)
#pragma unroll
                for (int f = 0; f < MR_NT; f++) {
                    if (dbg & 8) { acc[f][m][0] += (float)bhv[f][0] + (float)blv[f][1]; continue; }
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oph[s][m], bhv[f], acc[f][m], 0, 0, 0);
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oph[s][m], blv[f], acc[f][m], 0, 0, 0);
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(opl[s][m], bhv[f], acc[f][m], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- cross-wave reduction: every wave parks all 12 partial tiles; wave w sums tiles w, w + 4, w + 8 in wave order -----
#pragma unroll
    for (int f = 0; f < MR_NT; f++)
#pragma unroll
        for (int m = 0; m < MR_MT; m++)
            red[(wave * (MR_NT * MR_MT) + f * MR_MT + m) * 64 + lane] = make_float4(acc[f][m][0], acc[f][m][1], acc[f][m][2], acc[f][m][3]);
    __syncthreads();
    float *out = spec + ((size_t)it.sg * n_branches + branch) * bp.n_mels * bp.n_frames;
#pragma unroll
    for (int q = 0; q < (MR_NT * MR_MT) / 4; q++) {
        const int tl = wave + 4 * q, f = tl / MR_MT, m = tl - f * MR_MT;
        float4 v = red[(0 * (MR_NT * MR_MT) + tl) * 64 + lane];
#pragma unroll
        for (int s = 1; s < 4; s++) {
            const float4 p = red[(s * (MR_NT * MR_MT) + tl) * 64 + lane];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int t = it.tl * MR_FR + f * 16 + li;
        if (t < bp.n_frames) {
            const float tot[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int mel = m * 16 + kq * 4 + r;
                if (mel < bp.n_mels) {
                    // (v^2)^expo = exp2(expo * log2(v^2) + log2_bias): the operator planes carry 2^s (BranchParams::log2_bias)
                    float o = (dbg & 2) ? tot[r] : __builtin_amdgcn_exp2f(__builtin_fmaf(bp.expo, __builtin_amdgcn_logf(tot[r] * tot[r]), bp.log2_bias));
                    o = o * bp.out_scale + bp.out_shift;
                    const int row = bp.flip ? (bp.n_mels - 1 - mel) : mel;
                    out[(size_t)row * bp.n_frames + t] = o;
                }
            }
        }
    }
}

// the items of role (isA, idx) in pass P, in processing order; n = how many
struct MrPlan {
    int n_seg, n_pass;
    __device__ MrItem a_item(int P, int i, int jj) const {
        const int Q = P >> 2, p = P & 3;
        MrItem r; r.sg = MR_GSEG * Q + i; r.tl = MR_TQ * p + jj; r.ok = (P < n_pass && r.sg < n_seg) ? 1 : 0;
        return r;
    }
    __device__ MrItem b_item(int P, int k, int n) const {   // n-th item of B workgroup k: q = k + 3 n over (tile jj, segment i), tile-major
        const int Q = P >> 2, p = P & 3, q = k + MR_NB * n, jj = q / MR_GSEG, i = q - jj * MR_GSEG;
        MrItem r; r.sg = MR_GSEG * Q + i; r.tl = MR_TQ * p + jj; r.ok = (P < n_pass && q < MR_GSEG * MR_TQ && r.sg < n_seg) ? 1 : 0;
        return r;
    }
};

template <int KS>
__device__ __forceinline__ void mr_load_operator(const float *__restrict__ gf, int wave, int lane, f16x8 (&oph)[KS][MR_MT], f16x8 (&opl)[KS][MR_MT]) {
    // planes: [step of 32 k][mel tile]{hi, lo}[64 lanes][8 halves] (api.hip build_gf, prec 3)
    const f16x8 *g = reinterpret_cast<const f16x8 *>(gf) + lane;
#pragma unroll
    for (int s = 0; s < KS; s++)
#pragma unroll
        for (int m = 0; m < MR_MT; m++) {
            oph[s][m] = g[(((size_t)(wave * KS + s) * MR_MT + m) * 2 + 0) * 64];
            opl[s][m] = g[(((size_t)(wave * KS + s) * MR_MT + m) * 2 + 1) * 64];
        }
}

// role walk: the workgroup's items over all its passes, with the one-item-ahead span prefetch
template <int KS, bool IS_A>
__device__ __forceinline__ void mr_run(const float *__restrict__ x, const float *__restrict__ mm, float *__restrict__ spec,
                                       const FrontendParams *__restrict__ pp, const float *__restrict__ gf, const int branch,
                                       const MrPlan plan, const int group, const int n_groups, const int ridx, float *smem, const int dbg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f16x8 oph[KS][MR_MT], opl[KS][MR_MT];
    mr_load_operator<KS>(gf, wave, lane, oph, opl);
    const BranchParams bp = pp->br[branch];
    const int S = pp->sample_count, n_branches = pp->n_branches;
    const long n_total = (long)plan.n_seg * S;
    const float eps = pp->norm_eps;
    const int buf_floats = (((MR_FR - 1) * max(pp->br[0].H, pp->br[1].H) + max(pp->br[0].L, pp->br[1].L) + 255) >> 8) << 8;
    float *buf0 = smem, *buf1 = smem + buf_floats;
    float4 *red = reinterpret_cast<float4 *>(smem + 2 * buf_floats);
    constexpr int PER_PASS = IS_A ? MR_TQ : (MR_GSEG * MR_TQ + MR_NB - 1) / MR_NB;   // item slots per pass (B: 7, some empty)
    auto item_at = [&](int n) -> MrItem {   // n-th item slot of this workgroup over all its passes
        const int pi = n / PER_PASS, w = n - pi * PER_PASS, P = group + pi * n_groups;
        return IS_A ? plan.a_item(P, ridx, w) : plan.b_item(P, ridx, w);
    };
    const int n_slots = ((plan.n_pass - group + n_groups - 1) / n_groups) * PER_PASS;
    // first valid item: its span is fetched here
    int n = 0;
    MrItem cur = item_at(0);
    while (n < n_slots && !cur.ok) cur = item_at(++n);
    if (n >= n_slots) return;
    {
        const int span_pieces = ((MR_FR - 1) * bp.H + bp.L + 255) >> 8;
        const long base = (long)cur.sg * S + (long)cur.tl * MR_FR * bp.H;
        for (int p = wave; p < span_pieces; p += 4) {
            long idx = base + p * 256 + lane * 4;
            idx = idx < n_total - 4 ? idx : n_total - 4;
            const unsigned la = (unsigned)(size_t)(buf0 + p * 256);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(x + idx), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
        }
    }
    int parity = 0;
    while (true) {
        int n2 = n + 1;
        MrItem nx = n2 < n_slots ? item_at(n2) : MrItem{0, 0, 0};
        while (n2 < n_slots && !nx.ok) { n2++; nx = n2 < n_slots ? item_at(n2) : MrItem{0, 0, 0}; }
        mr_item<KS>(x, mm, spec, bp, branch, n_branches, S, n_total, eps, cur, nx, parity ? buf1 : buf0, parity ? buf0 : buf1, red, oph, opl, tid, dbg);
        if (!nx.ok) break;
        cur = nx; n = n2; parity ^= 1;
    }
}

__global__ __launch_bounds__(256, 1) void melr_kernel(const float *__restrict__ x, const float *__restrict__ mm, float *__restrict__ spec,
                                                       const FrontendParams *__restrict__ pp, const float *__restrict__ gf0,
                                                       const float *__restrict__ gf1, const int n_seg, const int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // workgroup b runs on XCD b % 8: the 8 workgroups of a group are b = xcd + 8 (8 g + r), r = 0..7
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, g_in = slot >> 3, role = slot & 7;
    const int groups_per_xcd = (int)gridDim.x >> 6, n_groups = groups_per_xcd * 8, group = xcd * groups_per_xcd + g_in;
    MrPlan plan;
    plan.n_seg = n_seg;
    plan.n_pass = ((n_seg + MR_GSEG - 1) / MR_GSEG) * 4;
    if (role < MR_GSEG) mr_run<8, true>(x, mm, spec, pp, gf0, 0, plan, group, n_groups, role, smem, dbg);
    else mr_run<4, false>(x, mm, spec, pp, gf1, 1, plan, group, n_groups, role - MR_GSEG, smem, dbg);
}

}  // namespace

// the front-end shapes this kernel is built for (and BIRDA_HIP_MELR=0 not set)
bool melr_supports(const FrontendParams &p) {
    static const bool off = getenv("BIRDA_HIP_MELR") && getenv("BIRDA_HIP_MELR")[0] == '0';
    if (off || p.prec != 3 || p.n_branches != 2) return false;
    const BranchParams &a = p.br[0], &b = p.br[1];
    if (a.nm_pad != 16 * MR_MT || b.nm_pad != 16 * MR_MT || a.K != 1024 || b.K != 512) return false;
    if (a.n_frames != b.n_frames || (a.n_frames + MR_FR - 1) / MR_FR != 4 * MR_TQ) return false;
    if ((MR_FR * a.H) % 4 || (MR_FR * b.H) % 4 || p.sample_count % 4) return false;   // 16-byte span pieces
    if (device_cu_count() < 64) return false;
    const int buf_floats = (((MR_FR - 1) * std::max(a.H, b.H) + std::max(a.L, b.L) + 255) >> 8) << 8;
    return (size_t)2 * buf_floats * sizeof(float) + (size_t)4 * MR_NT * MR_MT * 64 * sizeof(float4) <= 160 * 1024;
}

// true when the front-end is the shape this kernel is built for and the launch was made
bool launch_melr(const float *x, const float *minmax, float *spec, const FrontendParams &p, const FrontendParams *d_p, int n_seg, hipStream_t s) {
    if (!melr_supports(p)) return false;
    const BranchParams &a = p.br[0], &b = p.br[1];
    const int n_wg = (device_cu_count() / 64) * 64;   // whole groups of 8 per XCD
    const int buf_floats = (((MR_FR - 1) * std::max(a.H, b.H) + std::max(a.L, b.L) + 255) >> 8) << 8;
    const size_t smem = (size_t)2 * buf_floats * sizeof(float) + (size_t)4 * MR_NT * MR_MT * 64 * sizeof(float4);
    static const int dbg = getenv("BIRDA_HIP_MEL_DBG") ? atoi(getenv("BIRDA_HIP_MEL_DBG")) : 0;
    static DeviceOnce attr;
    attr.run([] { (void)hipFuncSetAttribute((const void *)melr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(melr_kernel, dim3(n_wg), dim3(256), smem, s, x, minmax, spec, d_p, a.gf, b.gf, n_seg, dbg);
    return true;
}

}  // namespace bh
