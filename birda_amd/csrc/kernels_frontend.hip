// Front-end kernels for gfx950: per-segment min/max and the folded STFT x mel GEMM.
//
// What the ONNX graph does per segment (birdnet-onnx/ORT in the
// reference, reached from src/inference/classifier.rs:478-488):
//   x <- 2((x - min)/(max - min + eps) - 0.5); frames (L, hop H, periodic Hann);
//   Re(rFFT) . mel_W ; square ; ^expo ; flip mel ; [mel][time].
// Because only Re() of the STFT is kept and the mel projection is applied BEFORE squaring,
// window + DFT + mel are one linear operator G[n][m] = w[n] sum_k cos(2 pi k n / L) W[k][m].
// G[0] = 0 (Hann) and G[L-n] = G[n], so frame t reduces to K = L/2 folded samples
//   y_t[j] = x[tH + j + 1] + x[tH + L - 1 - j]      (j = 0..K-1; last row of Gf halved)
// and spec_t = Gf^T y_t: a [n_mels x K] x [K x n_frames] GEMM per segment and branch, run on
// the f32 MFMA (v_mfma_f32_16x16x4_f32, exact f32 fmaf chains).
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace bh {

constexpr int MM_SPLIT = 8;  // partial min/max blocks per segment

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// grid (MM_SPLIT, n_seg), block 256: partial min/max of one slice of a segment
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ x, float *__restrict__ mm,
                                                      unsigned *__restrict__ in_bad, int sample_count) {
    const int seg = blockIdx.y, part = blockIdx.x;
    const int n4 = sample_count >> 2;
    const int per = (n4 + MM_SPLIT - 1) / MM_SPLIT;
    const int lo = part * per, hi = min(n4, lo + per);
    const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)seg * sample_count);
    float mn = INFINITY, mx = -INFINITY;
    // fminf / fmaxf skip NaN, so a corrupt decode would go unnoticed here: z accumulates v * 0, which is 0 for every finite
    // sample and NaN for inf / NaN (two packed FMAs per 16 bytes)
    float z0 = 0.0f, z1 = 0.0f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float4 v = p[i];
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        z0 = __builtin_fmaf(v.x, 0.0f, z0); z1 = __builtin_fmaf(v.y, 0.0f, z1);
        z0 = __builtin_fmaf(v.z, 0.0f, z0); z1 = __builtin_fmaf(v.w, 0.0f, z1);
    }
    if (part == MM_SPLIT - 1)  // scalar tail when sample_count % 4 != 0
        for (int i = (n4 << 2) + threadIdx.x; i < sample_count; i += 256) {
            float v = x[(size_t)seg * sample_count + i];
            mn = fminf(mn, v); mx = fmaxf(mx, v);
            z0 = __builtin_fmaf(v, 0.0f, z0);
        }
    mn = wave_min(mn); mx = wave_max(mx);
    const int any_bad = __syncthreads_or((z0 + z1) != 0.0f ? 1 : 0);   // NaN != 0
    __shared__ float s[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[w] = mn; s[4 + w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
        mx = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
        mm[((size_t)seg * MM_SPLIT + part) * 2 + 0] = mn;
        mm[((size_t)seg * MM_SPLIT + part) * 2 + 1] = mx;
        if (in_bad) in_bad[(size_t)seg * MM_SPLIT + part] = any_bad ? 1u : 0u;
    }
}

void launch_minmax(const float *x, float *minmax, unsigned *in_bad, int n_seg, int sample_count, hipStream_t s) {
    hipLaunchKernelGGL(minmax_kernel, dim3(MM_SPLIT, n_seg), dim3(256), 0, s, x, minmax, in_bad, sample_count);
}

// ---------------------------------------------------------------------------------------
// PCM -> mono f32 segments on the device (reference src/audio/decode.rs:353-411 append_samples:
// s / 32768.0 per channel for 16-bit, s / 2147483648.0 for 32-bit -- and for 24-bit, which symphonia widens into its S32
// buffer (value << 8) -- float32 as is; channels summed then divided by their count; :150-202 next_segment: the
// segment is zero-padded past the end of the stream).  grid (blocks over the segment, n_seg).  `pcm` is the byte address of
// frame 0 (a virtual origin: the caller's buffer holds the slice's frames only); FMT = the host decoder's sample formats.
// ---------------------------------------------------------------------------------------
template <int FMT>   // 1: int16, 2: int24 (3 bytes, little endian), 3: int32, 4: float32
__device__ __forceinline__ float pcm_sample(const unsigned char *p) {
    if constexpr (FMT == 1) return (float)*reinterpret_cast<const int16_t *>(p) / 32768.0f;
    else if constexpr (FMT == 2) {
        const int32_t v = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
        return (float)v / 2147483648.0f;
    } else if constexpr (FMT == 3) return (float)*reinterpret_cast<const int32_t *>(p) / 2147483648.0f;
    else return *reinterpret_cast<const float *>(p);
}

template <int FMT>
__global__ __launch_bounds__(256) void segment_pcm_kernel(const unsigned char *__restrict__ pcm, long n_frames, int channels,
                                                           const unsigned long long *__restrict__ starts,
                                                           int seg_len, float *__restrict__ out, long out_stride) {
    constexpr int BPS = FMT == 1 ? 2 : FMT == 2 ? 3 : 4;
    const int seg = blockIdx.y;
    const long s0 = (long)starts[seg];
    float *o = out + (long)seg * out_stride;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < seg_len; j += gridDim.x * 256) {
        const long f = s0 + j;
        float v = 0.0f;
        if (f < n_frames) {
            if (channels == 1) {
                v = pcm_sample<FMT>(pcm + f * BPS);
            } else {
                float sum = 0.0f;
                for (int c = 0; c < channels; c++) sum += pcm_sample<FMT>(pcm + (f * channels + c) * BPS);
                v = sum / (float)channels;
            }
        }
        o[j] = v;
    }
}

void launch_segment_pcm(const void *d_pcm_origin, int sample_format, size_t n_frames, int channels, const unsigned long long *d_starts,
                        int n_seg, int seg_len, float *d_out, size_t out_stride, hipStream_t s) {
    dim3 grid((unsigned)std::min<size_t>(((size_t)seg_len + 255) / 256, 64), n_seg), block(256);
    const unsigned char *p = static_cast<const unsigned char *>(d_pcm_origin);
#define BH_SEG(F) hipLaunchKernelGGL(segment_pcm_kernel<F>, grid, block, 0, s, p, (long)n_frames, channels, d_starts, seg_len, d_out, (long)out_stride)
    switch (sample_format) {
    case 2: BH_SEG(2); break;
    case 3: BH_SEG(3); break;
    case 4: BH_SEG(4); break;
    default: BH_SEG(1); break;
    }
#undef BH_SEG
}

// ---------------------------------------------------------------------------------------
// mel kernel: PERSISTENT workgroups (256 threads = 4 waves, two per CU) walking the work items
// (segment, branch, 48-frame tile); the next item's sample span is prefetched into registers while the
// current item's reduction and epilogue run.
//
// LDS holds only the tile's normalised sample span xs[(48-1)H + L] (60 KB for L = 2048; with 64
// frames it was 78 KB and a second block never fitted beside the first: staging and epilogue
// then ran with the MFMA pipe idle).
// The K = L/2 reduction is SPLIT ACROSS THE 4 WAVES: wave w accumulates k in
// [wK/4, (w+1)K/4) for all 64 frames x all mel tiles (4 x MT accumulator tiles), so the main
// loop has no block barrier.  A = Gf^T comes straight from global/L2 in an MFMA-fragment-major
// layout gfF[g][mt][lane][c] (k = 16g + 4(lane>>4) + c, mel = 16mt + (lane&15)): one coalesced
// 1-KiB dwordx4 load per mel tile feeds four MFMA k-steps, prefetched one 16-k group ahead.
// B = folded frames from LDS.  The four partial sums meet in LDS (reusing xs); wave w then
// owns frame tile w for the epilogue (square, power law, affine, flip, store).
// ---------------------------------------------------------------------------------------
// BH_MEL_PIPE = 1: a software-pipelined main loop for the split-f16 kernel (the frame fragments of step s + 1 built under the
// MFMAs of step s, sched_group_barrier groups of {BH_MEL_GM MFMA, BH_MEL_GD LDS reads, BH_MEL_GV VALU}).  Round 4 measured it
// because the kernel turned out to be ISSUE-bound, not bound by the operator stream from L2 (with every step reading the same
// operator fragments, BIRDA_HIP_MEL_DBG=16 in the EXPERIMENTS build, it is as slow: 713.9 against 712.7 us per launch): 0.704 ->
// 0.683 us per segment (-3 %, profiles/r4_l_mel_pipe.txt).  Not the product: two full operator sets, two sets of frame fragments
// and 72 accumulator registers leave hipcc 3-26 registers short whatever the group sizes; it spills thread-invariant offsets and
// reloads them behind `s_waitcnt vmcnt(0)` in the reduction (or, with other group sizes, inside the MFMA tail: 1.05 us).
// BH_MEL_PIPE = 2 (round 6, THE PRODUCT): the same pipeline with the operator held half a step at a time -- one register set per half
// of the mel tiles instead of two whole sets (48 registers instead of 96): 16 bytes of scratch instead of 148, and the loop that
// round 4 could only measure ships: 0.706 -> 0.678 us per segment on a box in its slow mode, -4.2 % with the fused blocks of the same
// runs as the clock reference (profiles/r6_g_mel_pipe_tuning.txt: group sizes 3 / 6 / 9 and the split of the fragment build over the
// two phases within 1 % of each other; round 4's form on today's compiler 1.17 us -- it spills inside the MFMA tail).  0 restores the
// plain loop (two operator sets, fragments built in front of their MFMAs).
#ifndef BH_MEL_PIPE
#define BH_MEL_PIPE 2
#endif
#ifndef BH_MEL_GM
#define BH_MEL_GM 6
#define BH_MEL_GD 6
#define BH_MEL_GV 8
#endif
constexpr int MEL_FT = 3;           // 16-frame tiles per block: 3 keeps the span at 60 KB -> two blocks per CU
constexpr int MEL_TN = 16 * MEL_FT;

// prefetched 16-B loads per thread: 15 cover the 60-KB spans of the 96-mel (v2.4) front-ends in 60 registers, 16 the
// 64-KB span of the 128-mel one, whose workgroups have the whole register file

// (128-mel front-ends need 98 KB of LDS for the reduction: one workgroup per CU, so they get the whole register file)
// HALVES = 2 (round 4): ONE workgroup of 8 waves per CU works a 96-frame item -- waves 0-3 the first 48 frames,
// waves 4-7 the last 48, each four splitting K as before.  Wave w and wave w + 4 sit on the same SIMD and read the SAME operator
// fragments within a few hundred cycles of each other, so the second read is served by the CU's L1 instead of the L2 -> CU
// fabric (the stream that bounds this kernel: 6.5 MB per segment with 48-frame items); the staged span is shared by both
// halves (96 frames: 115 KB), and the cross-wave reduction buffer (2 x 74 KB) aliases it.
template <int MT, int PREC, int HALVES = 1>
__global__ __launch_bounds__(256 * HALVES, (MT <= 6 || HALVES == 2) ? 2 : 1) void mel_kernel(const float *__restrict__ x, const float *__restrict__ mm,
                                                      float *__restrict__ spec, const FrontendParams *__restrict__ pp,
                                                      const float *__restrict__ gf0, const float *__restrict__ gf1,
                                                      const float *__restrict__ gf2, const float *__restrict__ gf3,
                                                      const int dbg_arg, const int n_tiles, const int n_items, const int paired) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (the ablation bits, BIRDA_HIP_MEL_DBG, exist in the EXPERIMENTS build only: a compile-time zero in the product)
#ifdef BIRDA_HIP_EXPERIMENTS
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0; (void)dbg_arg;
#endif
    const int n_branches = pp->n_branches;
    const int S = pp->sample_count;
    float *xs = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int half = HALVES == 2 ? tid >> 8 : 0, wave = (tid >> 6) & 3;   // wave: the K quarter, as in the 4-wave kernel
    const int li = lane & 15, kq = lane >> 4;
    constexpr int TNW = MEL_TN * HALVES, NTHR = 256 * HALVES;             // frames per item, threads

    // PERSISTENT workgroups: each walks the work items (segment, branch, frame tile) with stride gridDim.x, and
    // the sample span of its NEXT item is fetched into registers (MEL_SU x 16 B per thread, free at that point)
    // while the reduction and the epilogue of the current one run; a workgroup that fetched its span only when
    // it started (two per CU) exposed one HBM round trip per tile, 0.37 of this kernel's 0.95 us per segment.
    // item = (seg * n_branches + branch) * n_tiles + tile
    static_assert(MM_SPLIT == 8, "16 floats of min / max partials per segment");
    constexpr int MEL_SU = HALVES == 2 ? 14 : (MT <= 6 ? 15 : 16);   // (two halves: 28 460 floats of span over 512 threads)
    float4 q[MEL_SU], mmq[4];
    // Work items and the XCD-aware pairing.  Unpaired: item = (seg * n_branches + branch) * n_tiles + tile, workgroup b takes
    // b, b + gridDim.x, ...  Paired (n_branches > 1, grid a multiple of 8 n_branches): the branches of one (segment, tile)
    // read almost the same sample span, so they go to workgroups b and b + 8 (+ 16 ...) -- the same XCD under the round-robin
    // dispatch, hence the same L2 -- at the same step n, and the second read of the span hits L2 instead of HBM (it was
    // 2.3x the algorithmic read traffic).  The roles rotate by one branch per step so that partners with unequal branches
    // (K = 1024 against 512) do equal work over n_branches steps and stay within a step of each other.
    const int pr_xcd = blockIdx.x & 7, pr_slot = blockIdx.x >> 3;
    const int pr_role = paired ? pr_slot % n_branches : 0, pr_q = (pr_slot / n_branches) * 8 + pr_xcd;
    const int pr_stride = (int)gridDim.x / n_branches, n_pairs = n_items / n_branches;
    struct MelItem { int sg, br, tl, ok; };
    auto decode = [&](int n) __attribute__((always_inline)) -> MelItem {
        MelItem r;
        if (paired) {
            const int P = pr_q + n * pr_stride;
            r.br = (pr_role + n) % n_branches;
            r.sg = P / n_tiles; r.tl = P - r.sg * n_tiles;
            r.ok = P < n_pairs;
        } else {
            const int it = (int)blockIdx.x + n * (int)gridDim.x;
            r.tl = it % n_tiles;
            const int sb = it / n_tiles;
            r.sg = sb / n_branches; r.br = sb - r.sg * n_branches;
            r.ok = it < n_items;
        }
        return r;
    };
    auto issue = [&](int sg, int br, int tl) __attribute__((always_inline)) {
        const int Hn = pp->br[br].H, Ln = pp->br[br].L;
        const int sp = ((TNW - 1) * Hn + Ln + 3) & ~3, g0 = tl * TNW * Hn;
        const float *xg = x + (size_t)sg * S;
        // Unconditional loads from clamped addresses: every q[u] is (re)defined on every pass, so the 64 registers are
        // live only from here to the LDS write at the top of the next pass, not across the main loop.  Pieces past the
        // segment end (S and g0 are multiples of 4: a piece is wholly inside or wholly outside) or past the span read a
        // valid address and are replaced by -1 / ignored when the span is written.
        (void)sp;
#pragma unroll
        for (int u = 0; u < MEL_SU; u++) q[u] = *reinterpret_cast<const float4 *>(xg + min(g0 + tid * 4 + u * 4 * NTHR, S - 4));
        const float4 *mv = reinterpret_cast<const float4 *>(mm + (size_t)sg * MM_SPLIT * 2);
        mmq[0] = mv[0]; mmq[1] = mv[1]; mmq[2] = mv[2]; mmq[3] = mv[3];
    };
    int step_n = 0;
    MelItem cur = decode(0);
    if (cur.ok) issue(cur.sg, cur.br, cur.tl);

    while (cur.ok) {
    const int seg = cur.sg, branch = cur.br, tile = cur.tl;
    BranchParams bp = pp->br[branch];
    // the operator pointer comes in as a kernel argument (global address space): read through the
    // struct it is a generic pointer, hipcc emits flat_load, and every LDS wait then also drains the
    // prefetched operator loads
    const float *__restrict__ gfp = branch == 0 ? gf0 : branch == 1 ? gf1 : branch == 2 ? gf2 : gf3;
    const int t0 = tile * TNW;
    const int L = bp.L, H = bp.H, K = bp.K;
    const int span = (TNW - 1) * H + L;
    const int span_pad = (span + 3) & ~3;

    // x <- 2((x - min)/(max - min + eps) - 0.5) as a subtract and an fma per sample: (x - min) * sc - 1.
    // (An IEEE division is ~12 VALU instructions; with 2 x 1.1 passes over every sample it was a
    // third of this kernel's vector work.)  Differs from the divide form by <= 2 ulp of the
    // normalised sample and keeps x == min exactly at -1 (constant segments).
    const float mn = fminf(fminf(fminf(mmq[0].x, mmq[0].z), fminf(mmq[1].x, mmq[1].z)), fminf(fminf(mmq[2].x, mmq[2].z), fminf(mmq[3].x, mmq[3].z)));
    const float mx = fmaxf(fmaxf(fmaxf(mmq[0].y, mmq[0].w), fmaxf(mmq[1].y, mmq[1].w)), fmaxf(fmaxf(mmq[2].y, mmq[2].w), fmaxf(mmq[3].y, mmq[3].w)));
    const float sc = 2.0f / ((mx - mn) + pp->norm_eps);

    // the prefetched span -> LDS, normalised; samples past the segment end become the normalised minimum (-1)
    const float *xseg = x + (size_t)seg * S;
    const int g0s = t0 * H;
    if (g0s + span_pad <= S) {
        // (the whole span lies inside the segment -- every tile but the last one or two of a segment: no per-sample end test,
        //  8 of the ~20 vector instructions per staged float4)
#pragma unroll
        for (int u = 0; u < MEL_SU; u++) {
            const int i = tid * 4 + u * 4 * NTHR;
            if (i < span_pad) {
                float4 v = q[u];
                v.x = fmaf(v.x - mn, sc, -1.0f); v.y = fmaf(v.y - mn, sc, -1.0f);
                v.z = fmaf(v.z - mn, sc, -1.0f); v.w = fmaf(v.w - mn, sc, -1.0f);
                *reinterpret_cast<float4 *>(xs + i) = v;
            }
        }
    } else {
#pragma unroll
    for (int u = 0; u < MEL_SU; u++) {
        const int i = tid * 4 + u * 4 * NTHR;
        if (i < span_pad) {
            float4 v = q[u];
            v.x = (g0s + i + 0 < S) ? fmaf(v.x - mn, sc, -1.0f) : -1.0f;
            v.y = (g0s + i + 1 < S) ? fmaf(v.y - mn, sc, -1.0f) : -1.0f;
            v.z = (g0s + i + 2 < S) ? fmaf(v.z - mn, sc, -1.0f) : -1.0f;
            v.w = (g0s + i + 3 < S) ? fmaf(v.w - mn, sc, -1.0f) : -1.0f;
            *reinterpret_cast<float4 *>(xs + i) = v;
        }
    }
    }
    for (int i = MEL_SU * 4 * NTHR + tid * 4; i < span_pad; i += 4 * NTHR) {   // spans beyond the prefetch capacity (none of the built models)
        float4 v;
        v.x = (g0s + i + 0 < S) ? fmaf(xseg[g0s + i + 0] - mn, sc, -1.0f) : -1.0f;
        v.y = (g0s + i + 1 < S) ? fmaf(xseg[g0s + i + 1] - mn, sc, -1.0f) : -1.0f;
        v.z = (g0s + i + 2 < S) ? fmaf(xseg[g0s + i + 2] - mn, sc, -1.0f) : -1.0f;
        v.w = (g0s + i + 3 < S) ? fmaf(xseg[g0s + i + 3] - mn, sc, -1.0f) : -1.0f;
        *reinterpret_cast<float4 *>(xs + i) = v;
    }
    __syncthreads();

    f32x4 acc[MEL_FT][MT];
#pragma unroll
    for (int f = 0; f < MEL_FT; f++)
#pragma unroll
        for (int m = 0; m < MT; m++) acc[f][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int gpw = K / 64;          // 16-k groups per wave
    const int gbeg = wave * gpw;
    const float4 *gA = reinterpret_cast<const float4 *>(gfp) + lane;
    if constexpr (PREC == 3) {
        // Split-f16 MFMA (v_mfma_f32_16x16x32_f16 x 3 per product: hi*hi + hi*lo + lo*hi, f32 accumulate):
        // f32-grade sums at 4.4x the f32 MFMA rate.  The operator arrives pre-split (hi / lo planes);
        // the folded frame samples are split here, 8 consecutive k per lane and 32-deep step.
        const int spw = K / 128, sbeg = wave * spw;              // 32-deep steps per wave (L % 256 == 0, checked at create)
        const f16x8 *gA = reinterpret_cast<const f16x8 *>(gfp) + lane;
        [[maybe_unused]] f16x8 a0h[MT], a0l[MT], a1h[MT], a1l[MT];
        [[maybe_unused]] auto load = [&](int st, f16x8 (&ah)[MT], f16x8 (&al)[MT]) {
            if (dbg & 16) st = sbeg;   // (ablation: every step reads the wave's FIRST operator fragments -- the kernel without the L2 -> CU operator stream)
#pragma unroll
            for (int m = 0; m < MT; m++) {
                ah[m] = gA[(((size_t)st * MT + m) * 2 + 0) * 64];
                al[m] = gA[(((size_t)st * MT + m) * 2 + 1) * 64];
            }
        };
        const float *xf = xs + (half * MEL_TN + li) * H;
        [[maybe_unused]] auto step = [&](int st, const f16x8 (&ah)[MT], const f16x8 (&al)[MT]) {
            // element jj of the lane's operand fragment is k = 32 st + 4 jj + kq (the operator planes are packed to match,
            // api.hip build_gf): the four lane groups read NEIGHBOURING samples, so the 32 lanes of an LDS access spread over
            // (16 frames x hop) + {0, 1}.  With runs of 8 k per group a hop of 278 put two lanes on every bank and 280 eight;
            // now 278 is conflict-free and 280 four-way (a hop that is 24 mod 32 has only four distinct frame banks; padding
            // the staged span so that frames sit hop + 2 apart removes that too on paper, but the scalar staging stores and
            // the pad bookkeeping cost more than the conflicts: 0.83 against 0.67 us per segment)
            const int j0 = st * 32 + kq;
            f16x8 bh[MEL_FT], bl[MEL_FT];
#pragma unroll
            for (int f = 0; f < MEL_FT; f++)
            {
                float y[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++)   // (one v_add_f32 each, on purpose: see bh_add_unpacked)
                    y[jj] = bh_add_unpacked(xf[f * 16 * H + j0 + 4 * jj + 1], xf[f * 16 * H + L - 1 - j0 - 4 * jj]);
                bh_split8(y, bh[f], bl[f]);
            }
            // The split stays out of the MFMA sequence.  (Not needed for correctness any more: the wrong tiles this
            // fence first reduced came from hipcc's v_pk_add_f32 op_sel form of the sums above -- bh_add_unpacked,
            // DESIGN.md section 3; without the fence the kernel is as fast and as deterministic, 200 soak runs.)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int f = 0; f < MEL_FT; f++) {
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[f], acc[f][m], 0, 0, 0);
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[f], acc[f][m], 0, 0, 0);
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[f], acc[f][m], 0, 0, 0);
                }
        };
#if BH_MEL_PIPE == 1
        // Software-pipelined form (round 4, -DBH_MEL_PIPE=1; measured alternative, not the product -- see the note at BH_MEL_PIPE):
        // the folded-and-split frame fragments of step s + 1 are built WHILE the MFMAs of step s issue.
        auto build = [&](int st, f16x8 (&bh)[MEL_FT], f16x8 (&bl)[MEL_FT]) {
            const int j0 = st * 32 + kq;
#pragma unroll
            for (int f = 0; f < MEL_FT; f++) {
                float y[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    y[jj] = bh_add_unpacked(xf[f * 16 * H + j0 + 4 * jj + 1], xf[f * 16 * H + L - 1 - j0 - 4 * jj]);
                bh_split8(y, bh[f], bl[f]);
            }
        };
        auto mma = [&](const f16x8 (&ah)[MT], const f16x8 (&al)[MT], const f16x8 (&bh)[MEL_FT], const f16x8 (&bl)[MEL_FT]) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int f = 0; f < MEL_FT; f++) {
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[f], acc[f][m], 0, 0, 0);
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[f], acc[f][m], 0, 0, 0);
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[f], acc[f][m], 0, 0, 0);
                }
        };
        // 54 MFMAs, 72 vector instructions and 48 LDS reads per step, dealt in groups of {GM MFMA, GD DS read, GV VALU}
        auto interleave = [&]() {
#pragma unroll
            for (int g = 0; g < 9 * MT / BH_MEL_GM; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, BH_MEL_GM, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, BH_MEL_GD, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, BH_MEL_GV, 0);
            }
        };
        f16x8 b0h[MEL_FT], b0l[MEL_FT], b1h[MEL_FT], b1l[MEL_FT];
        load(sbeg, a0h, a0l);
        if (!(dbg & 1) && (HALVES == 1 || t0 + half * MEL_TN < bp.n_frames)) {
            build(sbeg, b0h, b0l);
            __builtin_amdgcn_sched_barrier(0);
            int si = 0;
            for (; si + 2 < spw; si += 2) {
                load(sbeg + si + 1, a1h, a1l);
                __builtin_amdgcn_sched_barrier(0);
                build(sbeg + si + 1, b1h, b1l);
                mma(a0h, a0l, b0h, b0l);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                load(sbeg + si + 2, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                build(sbeg + si + 2, b0h, b0l);
                mma(a1h, a1l, b1h, b1l);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (si + 1 < spw) {
                load(sbeg + si + 1, a1h, a1l);
                __builtin_amdgcn_sched_barrier(0);
                build(sbeg + si + 1, b1h, b1l);
                mma(a0h, a0l, b0h, b0l);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
                mma(a1h, a1l, b1h, b1l);
            } else {
                mma(a0h, a0l, b0h, b0l);
            }
        }
#elif BH_MEL_PIPE == 2
        // Round 6: the pipelined loop with the operator held HALF a step at a time.  Round 4's form kept two whole
        // operator sets (96 registers) beside two sets of frame fragments (48) and the accumulators (72) and was 3-26 registers
        // short of two workgroups per CU.  Here the step's mel tiles are two halves L and H with ONE register set each (48 in all):
        //     phase 1   MFMAs of half L on the current fragments  |  the next step's fragments of frame tiles 0, 1 built beside them
        //               -> half L of the NEXT step is fetched (its registers were read by the MFMAs just issued)
        //     phase 2   MFMAs of half H                           |  frame tile 2 built beside them
        //               -> half H of the next step is fetched
        // so every operator fragment is in flight for half a step (27 MFMAs, ~450 cycles: an L2 round trip) instead of a whole one.
        #ifndef BH_MEL_FT1
#define BH_MEL_FT1 ((MEL_FT + 1) / 2)
#endif
        // (an odd number of mel tiles -- 48, 80, 112 mels -- splits as MH + (MH - 1): the register sets are MH wide, the short half leaves
        //  its last slot unused)
        constexpr int MH = (MT + 1) / 2, FT1 = BH_MEL_FT1;
        f16x8 aLh[MH], aLl[MH], aHh[MH], aHl[MH];
        auto load_half = [&](int st, int h, f16x8 (&ah)[MH], f16x8 (&al)[MH]) {
            if (dbg & 16) st = sbeg;
#pragma unroll
            for (int m = 0; m < MH; m++) {
                if (h * MH + m >= MT) continue;
                ah[m] = gA[(((size_t)st * MT + h * MH + m) * 2 + 0) * 64];
                al[m] = gA[(((size_t)st * MT + h * MH + m) * 2 + 1) * 64];
            }
        };
        auto build_f = [&](int st, int f0, int f1, f16x8 (&bh)[MEL_FT], f16x8 (&bl)[MEL_FT]) {
            const int j0 = st * 32 + kq;
#pragma unroll
            for (int f = 0; f < MEL_FT; f++) {
                if (f < f0 || f >= f1) continue;
                float y[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    y[jj] = bh_add_unpacked(xf[f * 16 * H + j0 + 4 * jj + 1], xf[f * 16 * H + L - 1 - j0 - 4 * jj]);
                bh_split8(y, bh[f], bl[f]);
            }
        };
        auto mma_half = [&](int h, const f16x8 (&ah)[MH], const f16x8 (&al)[MH], const f16x8 (&bh)[MEL_FT], const f16x8 (&bl)[MEL_FT]) {
#pragma unroll
            for (int m = 0; m < MH; m++)
#pragma unroll
                for (int f = 0; f < MEL_FT; f++) {
                    if (h * MH + m >= MT) continue;
                    const int mm = h * MH + m < MT ? h * MH + m : 0;
                    acc[f][mm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[f], acc[f][mm], 0, 0, 0);
                    acc[f][mm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[f], acc[f][mm], 0, 0, 0);
                    acc[f][mm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[f], acc[f][mm], 0, 0, 0);
                }
        };
        // (per phase: 9 MH MFMAs beside 24 VALU + 16 LDS reads per frame tile built)
        auto interleave = [&](int nf) {
            constexpr int NM = 9 * MH;
            const int groups = NM / BH_MEL_GM;
#pragma unroll
            for (int g = 0; g < NM / BH_MEL_GM; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, BH_MEL_GM, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (16 * FT1 + NM / BH_MEL_GM - 1) / (NM / BH_MEL_GM), 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (24 * FT1 + NM / BH_MEL_GM - 1) / (NM / BH_MEL_GM), 0);
            }
            (void)groups; (void)nf;
        };
        f16x8 b0h[MEL_FT], b0l[MEL_FT], b1h[MEL_FT], b1l[MEL_FT];
        load_half(sbeg, 0, aLh, aLl);
        load_half(sbeg, 1, aHh, aHl);
        if (!(dbg & 1) && (HALVES == 1 || t0 + half * MEL_TN < bp.n_frames)) {
            build_f(sbeg, 0, MEL_FT, b0h, b0l);
            __builtin_amdgcn_sched_barrier(0);
            // one step: MFMAs on (cur), fragments of the next step into (nxt); `more`: there is a next step (compile-time per call site)
            auto one = [&](int st, const f16x8 (&ch)[MEL_FT], const f16x8 (&cl)[MEL_FT], f16x8 (&nh)[MEL_FT], f16x8 (&nl)[MEL_FT], bool more) __attribute__((always_inline)) {
                if (more) build_f(st + 1, 0, FT1, nh, nl);
                mma_half(0, aLh, aLl, ch, cl);
                if (more) interleave(FT1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) load_half(st + 1, 0, aLh, aLl);
                __builtin_amdgcn_sched_barrier(0);
                if (more) build_f(st + 1, FT1, MEL_FT, nh, nl);
                mma_half(1, aHh, aHl, ch, cl);
                if (more) interleave(MEL_FT - FT1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) load_half(st + 1, 1, aHh, aHl);
                __builtin_amdgcn_sched_barrier(0);
            };
            int si = 0;
            for (; si + 2 < spw; si += 2) {
                one(sbeg + si, b0h, b0l, b1h, b1l, true);
                one(sbeg + si + 1, b1h, b1l, b0h, b0l, true);
            }
            if (si + 1 < spw) {
                one(sbeg + si, b0h, b0l, b1h, b1l, true);
                one(sbeg + si + 1, b1h, b1l, b0h, b0l, false);
            } else {
                one(sbeg + si, b0h, b0l, b1h, b1l, false);
            }
        }
#else
        // The last one or two steps are peeled so that no load inside the loop is conditional: with
        // `if (si + 2 < spw) load(...)` in the loop, hipcc turned the register sets into loop-carried
        // selects and waited for every load right after issuing it.
        load(sbeg, a0h, a0l);
        if (!(dbg & 1) && (HALVES == 1 || t0 + half * MEL_TN < bp.n_frames)) {   // (a half wholly past the last frame has nothing to add)
            int si = 0;
            for (; si + 2 < spw; si += 2) {
                load(sbeg + si + 1, a1h, a1l);
                __builtin_amdgcn_sched_barrier(0);
                step(sbeg + si, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                load(sbeg + si + 2, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                step(sbeg + si + 1, a1h, a1l);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (si + 1 < spw) {
                load(sbeg + si + 1, a1h, a1l);
                __builtin_amdgcn_sched_barrier(0);
                step(sbeg + si, a0h, a0l);
                __builtin_amdgcn_sched_barrier(0);
                step(sbeg + si + 1, a1h, a1l);
            } else {
                step(sbeg + si, a0h, a0l);
            }
        }
#endif
    } else {
    // Two operator register sets, used alternately (no copies): set B is loaded while set A feeds
    // the MFMAs and vice versa.  The scheduling fences keep each load a full group (96 MFMAs)
    // ahead of its use; left to itself hipcc re-loads the operator right in front of the MFMAs.
    float4 a0[MT], a1[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) a0[m] = gA[((size_t)gbeg * MT + m) * 64];

    const float *xf = xs + li * H;   // frame tile f adds f*16*H
    auto group = [&](int g, const float4 (&a)[MT]) {
        const int jb = g * 16 + 4 * kq;
        float b[4][MEL_FT];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int f = 0; f < MEL_FT; f++)
                b[c][f] = xf[f * 16 * H + jb + c + 1] + xf[f * 16 * H + L - 1 - jb - c];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float av = c == 0 ? a[m].x : c == 1 ? a[m].y : c == 2 ? a[m].z : a[m].w;
#pragma unroll
                for (int f = 0; f < MEL_FT; f++)
                    acc[f][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[c][f], acc[f][m], 0, 0, 0);
            }
    };
    if (!(dbg & 1)) {   // last one or two groups peeled: no conditional loads inside the loop
        int gi = 0;
        for (; gi + 2 < gpw; gi += 2) {
#pragma unroll
            for (int m = 0; m < MT; m++) a1[m] = gA[((size_t)(gbeg + gi + 1) * MT + m) * 64];
            __builtin_amdgcn_sched_barrier(0);
            group(gbeg + gi, a0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; m++) a0[m] = gA[((size_t)(gbeg + gi + 2) * MT + m) * 64];
            __builtin_amdgcn_sched_barrier(0);
            group(gbeg + gi + 1, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (gi + 1 < gpw) {
#pragma unroll
            for (int m = 0; m < MT; m++) a1[m] = gA[((size_t)(gbeg + gi + 1) * MT + m) * 64];
            __builtin_amdgcn_sched_barrier(0);
            group(gbeg + gi, a0);
            __builtin_amdgcn_sched_barrier(0);
            group(gbeg + gi + 1, a1);
        } else {
            group(gbeg + gi, a0);
        }
    }

    }  // PREC
    // the next item's span: in flight while the reduction and the epilogue below run
    const MelItem nxt = decode(step_n + 1);
    // (unconditional on purpose, see issue(); past the end the current span is fetched again and simply not used)
    issue(nxt.ok ? nxt.sg : seg, nxt.ok ? nxt.br : branch, nxt.ok ? nxt.tl : tile);
    // cross-wave reduction: wave s parks its partials for the frame tiles it does not own
    __syncthreads();  // every wave is done reading xs
    float4 *red = reinterpret_cast<float4 *>(smem);
#pragma unroll
    for (int f = 0; f < MEL_FT; f++) {
        if (f == wave || (dbg & 4)) continue;
        const int slot = f - (f > wave ? 1 : 0);
#pragma unroll
        for (int m = 0; m < MT; m++)
            red[(((half * 4 + wave) * 3 + slot) * MT + m) * 64 + lane] = make_float4(acc[f][m][0], acc[f][m][1], acc[f][m][2], acc[f][m][3]);
    }
    __syncthreads();
    if (wave < MEL_FT) {   // waves beyond the frame tiles own no epilogue tile
    f32x4 tot[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        // own partial, selected without dynamic register indexing
        f32x4 v = wave == 0 ? acc[0][m] : wave == 1 ? acc[1][m] : wave == 2 ? acc[2][m] : acc[MEL_FT - 1][m];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (s == wave || (dbg & 4)) continue;
            const int slot = wave - (wave > s ? 1 : 0);
            const float4 q = red[(((half * 4 + s) * 3 + slot) * MT + m) * 64 + lane];
            v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        }
        tot[m] = v;
    }

    // epilogue: square, power law, folded-BN affine, mel flip, [mel][time] store
    const int t = t0 + half * MEL_TN + wave * 16 + li;
    if (t < bp.n_frames) {
        float *out = spec + ((size_t)seg * n_branches + branch) * bp.n_mels * bp.n_frames;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int mel = m * 16 + kq * 4 + r;
                if (mel < bp.n_mels) {
                    const float v = tot[m][r];
                    // (v^2)^expo = exp2(expo * log2(v^2)): v_log_f32 + v_mul + v_exp_f32 instead of ocml powf
                    // (~70 instructions); relative error <= ~5e-7 for v^2 down to 1e-12, 0 -> 0
                    float o = (dbg & 2) ? v : __builtin_amdgcn_exp2f(__builtin_fmaf(bp.expo, __builtin_amdgcn_logf(v * v), bp.log2_bias));
                    o = o * bp.out_scale + bp.out_shift;
                    const int row = bp.flip ? (bp.n_mels - 1 - mel) : mel;
                    if (!(dbg & 8) || o == 12345.678f) out[__mul24(row, bp.n_frames) + t] = o;   // (24-bit product: a branch's [mel][time] plane is < 2^24 elements; v_mad_i64_i32 is quarter-rate)
                }
            }
    }
    }   // owner waves
    __syncthreads();   // the partials have been read: xs / red may be overwritten by the next item's span
    step_n++; cur = nxt;
    }   // items
}

// ---------------------------------------------------------------------------------------
// mel32 kernel (split-f16, FrontendParams::prec == 32): the same folded GEMM on v_mfma_f32_32x32x16_f16, for
// front-ends whose hop collapses mel_kernel's frame-strided LDS reads onto a few banks (a hop of 320 samples puts
// every frame of a fragment on ONE bank).  32-frame items; per wave and chunk of 64 k, phase A builds the folded
// samples y with the LANES ALONG k (conflict-free for any hop) and parks them as split-f16 rows, phase B reads the
// rows back as MFMA operands (see the comment in the main loop).  K is split across the 4 waves as in mel_kernel;
// every wave parks its partials in LDS and wave m sums and stores MEL tile m (32 mels x 32 frames: two 128-byte
// row pieces per store instruction).  The next item's span is fetched BEFORE the main loop.
// Operator planes: [step of 16 k][mel tile of 32]{hi, lo}[64 lanes][8 halves]; element jj of step s holds
// k = 64 (s / 4) + 8 (s % 4) + 4 (lane >> 5) + jj / 2 + 32 (jj % 2), mel = 32 mt + (lane & 31) (api.hip build_gf).
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MEL32_TN = 32;
constexpr int MEL32_YP = 36;   // dwords per staged Y row (32 + 4: 16-byte row pieces of 16 frames tile the 64 banks)

template <int MT32>
__global__ __launch_bounds__(256, 2) void mel32_kernel(const float *__restrict__ x, const float *__restrict__ mm,
                                                        float *__restrict__ spec, const FrontendParams *__restrict__ pp,
                                                        const float *__restrict__ gf0, const float *__restrict__ gf1,
                                                        const float *__restrict__ gf2, const float *__restrict__ gf3,
                                                        const int dbg, const int ybuf_off, const int n_tiles, const int n_items) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_branches = pp->n_branches;
    const int S = pp->sample_count;
    float *xs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fn = lane & 31, kh = lane >> 5;
    constexpr int SU = 11;   // prefetched 16-B pieces per thread (44 KB of span)
    float4 q[SU], mmq[4];
    auto issue = [&](int it) {
        const int tl = it % n_tiles, sb = it / n_tiles, sg = sb / n_branches, br = sb - sg * n_branches;
        const int g0 = tl * MEL32_TN * pp->br[br].H;
        const float *xg = x + (size_t)sg * S;
#pragma unroll
        for (int u = 0; u < SU; u++) q[u] = *reinterpret_cast<const float4 *>(xg + min(g0 + tid * 4 + u * 1024, S - 4));
        const float4 *mv = reinterpret_cast<const float4 *>(mm + (size_t)sg * MM_SPLIT * 2);
        mmq[0] = mv[0]; mmq[1] = mv[1]; mmq[2] = mv[2]; mmq[3] = mv[3];
    };
    int item = blockIdx.x;
    if (item < n_items) issue(item);
    while (item < n_items) {
        const int tile = item % n_tiles, sbr = item / n_tiles, seg = sbr / n_branches, branch = sbr - seg * n_branches;
        BranchParams bp = pp->br[branch];
        const float *__restrict__ gfp = branch == 0 ? gf0 : branch == 1 ? gf1 : branch == 2 ? gf2 : gf3;
        const int t0 = tile * MEL32_TN;
        const int L = bp.L, H = bp.H, K = bp.K;
        const int span_pad = ((MEL32_TN - 1) * H + L + 3) & ~3;
        const float mn = fminf(fminf(fminf(mmq[0].x, mmq[0].z), fminf(mmq[1].x, mmq[1].z)), fminf(fminf(mmq[2].x, mmq[2].z), fminf(mmq[3].x, mmq[3].z)));
        const float mx = fmaxf(fmaxf(fmaxf(mmq[0].y, mmq[0].w), fmaxf(mmq[1].y, mmq[1].w)), fmaxf(fmaxf(mmq[2].y, mmq[2].w), fmaxf(mmq[3].y, mmq[3].w)));
        const float sc = 2.0f / ((mx - mn) + pp->norm_eps);
        const float *xseg = x + (size_t)seg * S;
        const int g0s = t0 * H;
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const int i = tid * 4 + u * 1024;
            if (i < span_pad) {
                float4 v = q[u];
                v.x = (g0s + i + 0 < S) ? fmaf(v.x - mn, sc, -1.0f) : -1.0f;
                v.y = (g0s + i + 1 < S) ? fmaf(v.y - mn, sc, -1.0f) : -1.0f;
                v.z = (g0s + i + 2 < S) ? fmaf(v.z - mn, sc, -1.0f) : -1.0f;
                v.w = (g0s + i + 3 < S) ? fmaf(v.w - mn, sc, -1.0f) : -1.0f;
                *reinterpret_cast<float4 *>(xs + i) = v;
            }
        }
        for (int i = SU * 1024 + tid * 4; i < span_pad; i += 1024) {   // spans beyond the prefetch capacity
            float4 v;
            v.x = (g0s + i + 0 < S) ? fmaf(xseg[g0s + i + 0] - mn, sc, -1.0f) : -1.0f;
            v.y = (g0s + i + 1 < S) ? fmaf(xseg[g0s + i + 1] - mn, sc, -1.0f) : -1.0f;
            v.z = (g0s + i + 2 < S) ? fmaf(xseg[g0s + i + 2] - mn, sc, -1.0f) : -1.0f;
            v.w = (g0s + i + 3 < S) ? fmaf(xseg[g0s + i + 3] - mn, sc, -1.0f) : -1.0f;
            *reinterpret_cast<float4 *>(xs + i) = v;
        }
        __syncthreads();

        const int next = item + (int)gridDim.x;
        issue(min(next, n_items - 1));   // the next span flies under the main loop; unconditional (see mel_kernel)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[MT32];
#pragma unroll
        for (int m = 0; m < MT32; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][r] = 0.0f;
        // Per wave: K / 4 of the reduction, in chunks of 64 k.  Phase A stages the chunk's Y = x[t H + k + 1] + x[t H + L - 1 - k]
        // as split-f16 rows ([plane][frame][32 dwords, dword d = {k0 + d, k0 + d + 32}], pitch 36) with the lanes along k, so both span
        // reads and the row writes walk consecutive banks whatever the hop is (frames along the lanes put H mod 64 between
        // neighbours: 8 banks for a hop of 280, one for 320); the two lane halves take frames pair_dist apart, chosen so that
        // pair_dist * H = 32 (mod 64) when the hop allows it.  Phase B reads the operand fragments back as 16-byte row pieces
        // (pitch 36: conflict-free) and runs the MFMAs.  Both phases are wave-private: LDS executes a wave's accesses in order.
        const int nch = K / 256, kw = wave * (K / 4);
        int dd = 16;
#pragma unroll
        for (int c = 1; c <= 16; c *= 2)
            if (((c * H) & 63) == 32) dd = c;
        uint32_t *yb = reinterpret_cast<uint32_t *>(smem) + ybuf_off + wave * (2 * 32 * MEL32_YP);
        const int dl = lane & 31;
        const f16x8 *gA = reinterpret_cast<const f16x8 *>(gfp) + lane;
        f16x8 a0h[MT32], a0l[MT32], a1h[MT32], a1l[MT32];
        auto load = [&](int st, f16x8 (&ah)[MT32], f16x8 (&al)[MT32]) {
#pragma unroll
            for (int m = 0; m < MT32; m++) {
                ah[m] = gA[(((size_t)st * MT32 + m) * 2 + 0) * 64];
                al[m] = gA[(((size_t)st * MT32 + m) * 2 + 1) * 64];
            }
        };
        const f16x8 *yr = reinterpret_cast<const f16x8 *>(yb + fn * MEL32_YP + 4 * kh);   // row = frame, 4 dwords of this half
        auto step = [&](int sl, const f16x8 (&ah)[MT32], const f16x8 (&al)[MT32]) {
            const f16x8 bh = yr[2 * sl], bl = yr[2 * sl + (32 * MEL32_YP) / 4];
#pragma unroll
            for (int m = 0; m < MT32; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT32; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT32; m++) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh, acc[m], 0, 0, 0);
        };
        const int st0 = kw / 16;
        if (!(dbg & 1)) {
            load(st0, a0h, a0l);
            for (int c = 0; c < nch; c++) {
                const int kc = kw + 64 * c;
                const float *pf = xs + (kh * dd) * H + kc + dl + 1;
                const float *pb = xs + (kh * dd) * H + (L - 1 - kc - dl);
                uint32_t *yw = yb + (kh * dd) * MEL32_YP + dl;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int fs = ((i & ~(dd - 1)) << 1) | (i & (dd - 1));   // wave-uniform; the upper half adds dd
                    const float y0 = bh_add_unpacked(pf[fs * H], pb[fs * H]);            // single v_add_f32s on purpose
                    const float y1 = bh_add_unpacked(pf[fs * H + 32], pb[fs * H - 32]);
                    bh_f16x2 hi, lo;
                    bh_split2(y0, y1, hi, lo);
                    yw[fs * MEL32_YP] = __builtin_bit_cast(uint32_t, hi);
                    yw[fs * MEL32_YP + 32 * MEL32_YP] = __builtin_bit_cast(uint32_t, lo);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                const int sc0 = st0 + 4 * c;
                const bool last = c + 1 == nch;
                load(sc0 + 1, a1h, a1l);
                step(0, a0h, a0l);
                load(sc0 + 2, a0h, a0l);
                step(1, a1h, a1l);
                load(sc0 + 3, a1h, a1l);
                step(2, a0h, a0l);
                load(last ? sc0 + 3 : sc0 + 4, a0h, a0l);   // unconditional: the last chunk re-reads its own last step
                step(3, a1h, a1l);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // cross-wave reduction: every wave parks its partial of every MEL tile ([wave][tile][4 x float4][64 lanes]); wave m
        // sums tile m in wave order (no register selection by wave index: that would put the accumulators on the stack)
        __syncthreads();
        float4 *red = reinterpret_cast<float4 *>(smem);
#pragma unroll
        for (int m = 0; m < MT32; m++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++)
                red[((wave * MT32 + m) * 4 + r4) * 64 + lane] = make_float4(acc[m][4 * r4], acc[m][4 * r4 + 1], acc[m][4 * r4 + 2], acc[m][4 * r4 + 3]);
        __syncthreads();
        if (wave < MT32) {
            f32x16 tot;
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                float4 v = red[((0 * MT32 + wave) * 4 + r4) * 64 + lane];
#pragma unroll
                for (int sw = 1; sw < 4; sw++) {
                    const float4 w = red[((sw * MT32 + wave) * 4 + r4) * 64 + lane];
                    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
                }
                tot[4 * r4] = v.x; tot[4 * r4 + 1] = v.y; tot[4 * r4 + 2] = v.z; tot[4 * r4 + 3] = v.w;
            }
            // epilogue: square, power law, folded-BN affine, mel flip; lanes 0-31 / 32-63 each write a 128-byte row piece
            const int t = t0 + fn;
            if (t < bp.n_frames) {
                float *out = spec + ((size_t)seg * n_branches + branch) * bp.n_mels * bp.n_frames;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int mel = 32 * wave + (r & 3) + 4 * kh + 8 * (r >> 2);
                    if (mel < bp.n_mels) {
                        const float v = tot[r];
                        float o = __builtin_amdgcn_exp2f(__builtin_fmaf(bp.expo, __builtin_amdgcn_logf(v * v), bp.log2_bias));
                        o = o * bp.out_scale + bp.out_shift;
                        const int row = bp.flip ? (bp.n_mels - 1 - mel) : mel;
                        out[(size_t)row * bp.n_frames + t] = o;
                    }
                }
            }
        }
        __syncthreads();
        item = next;
    }
}

// LDS the front-end kernel of `p` asks for: a frame tile's sample span (+ the staged rows of mel32_kernel), or the reduction buffer.
// A hop so long that the span of one tile exceeds a CU's 160 KB cannot be launched: bh_classifier_create refuses the model.
size_t mel_lds_bytes(const FrontendParams &p) {
    int max_span = 0, span32 = 0;
    const int nmp = p.br[0].nm_pad;
    for (int b = 0; b < p.n_branches; b++) {
        max_span = std::max(max_span, (MEL_TN - 1) * p.br[b].H + p.br[b].L);
        span32 = std::max(span32, (MEL32_TN - 1) * p.br[b].H + p.br[b].L);
    }
    if (p.prec == 32)
        return std::max((size_t)(((span32 + 3) & ~3) + 4 * 2 * 32 * MEL32_YP) * sizeof(float), (size_t)4 * (nmp / 32) * 4 * 64 * sizeof(float4));
    return std::max((size_t)((max_span + 3) & ~3) * sizeof(float), (size_t)4 * 3 * (nmp / 16) * 64 * sizeof(float4));
}

void launch_mel(const float *x, const float *minmax, float *spec, const FrontendParams &p,
                const FrontendParams *d_p, int n_seg, hipStream_t s) {
    int max_span = 0, max_frames = 0, nmp = p.br[0].nm_pad;
    for (int b = 0; b < p.n_branches; b++) {
        int span = (MEL_TN - 1) * p.br[b].H + p.br[b].L;
        max_span = span > max_span ? span : max_span;
        max_frames = p.br[b].n_frames > max_frames ? p.br[b].n_frames : max_frames;
    }
    static const int dbg = [] { const char *e = BH_XENV("BIRDA_HIP_MEL_DBG"); return e ? atoi(e) : 0; }();  // tuning ablations
    if (p.prec == 32) {
        int span32 = 0;
        for (int b = 0; b < p.n_branches; b++) span32 = std::max(span32, (MEL32_TN - 1) * p.br[b].H + p.br[b].L);
        const int mt32 = nmp / 32;
        const int ybuf_off = (span32 + 3) & ~3;   // floats; the staged Y rows follow the span, the reduction buffer overlays both
        const size_t smem32 = std::max((size_t)(ybuf_off + 4 * 2 * 32 * MEL32_YP) * sizeof(float), (size_t)4 * mt32 * 4 * 64 * sizeof(float4));
        const int nt32 = (max_frames + MEL32_TN - 1) / MEL32_TN, ni32 = nt32 * p.n_branches * n_seg;
        const int n_cu32 = device_cu_count();
        dim3 g32((unsigned)std::min(ni32, 2 * n_cu32)), b32(256);
#define BH_MEL32(MTV)                                                                                                          \
    case MTV: {                                                                                                                \
        static DeviceOnce set32;                                                                                               \
        set32.run([] { (void)hipFuncSetAttribute((const void *)mel32_kernel<MTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        hipLaunchKernelGGL((mel32_kernel<MTV>), g32, b32, smem32, s, x, minmax, spec, d_p, p.br[0].gf, p.br[1].gf, p.br[2].gf, p.br[3].gf, dbg, ybuf_off, nt32, ni32); \
    } break;
        switch (mt32) {
            BH_MEL32(1)
            BH_MEL32(2)
            BH_MEL32(3)
            BH_MEL32(4)
        default: break;
        }
#undef BH_MEL32
        return;
    }
    const int mt = nmp / 16;
    // 8-wave workgroups on 96-frame items (HALVES = 2 above) for the split-f16 96-mel front-ends: built, parity-green, measured and
    // REJECTED (round 4): 0.714 against 0.661 us per segment on every one of four alternations on one box
    // (profiles/r4_h_mel_halves.txt) -- the twin waves' second reads of the operator fragments do not come cheaper out of L1 than
    // the one workgroup per CU loses by its eight waves meeting at every barrier of an item.  EXPERIMENTS build only,
    // BIRDA_HIP_MEL_HALVES=2.
    static const bool halves_on = BH_XENV("BIRDA_HIP_MEL_HALVES") && BH_XENV("BIRDA_HIP_MEL_HALVES")[0] == '2';
    int max_span2 = 0;
    for (int b = 0; b < p.n_branches; b++) max_span2 = std::max(max_span2, (2 * MEL_TN - 1) * p.br[b].H + p.br[b].L);
    const bool two = halves_on && p.prec == 3 && mt == 6 && ((size_t)((max_span2 + 3) & ~3) * 4 <= 160 * 1024) && ((max_span2 + 3) & ~3) <= 14 * 2048;
    const int tnw = two ? 2 * MEL_TN : MEL_TN;
    const size_t span_bytes = (size_t)(((two ? max_span2 : max_span) + 3) & ~3) * sizeof(float);
    const size_t red_bytes = (size_t)(two ? 8 : 4) * 3 * mt * 64 * sizeof(float4);
    const size_t smem = span_bytes > red_bytes ? span_bytes : red_bytes;
    const int n_tiles = (max_frames + tnw - 1) / tnw, n_items = n_tiles * p.n_branches * n_seg;
    const int n_cu = device_cu_count();
    int n_wg = std::min(n_items, ((mt <= 6 && !two) ? 2 : 1) * n_cu);   // persistent: as many workgroups as fit at once
    // branch partners on one XCD (see the kernel): needs whole groups of 8 n_branches workgroups
    static const bool pair_off = BH_XENV("BIRDA_HIP_MEL_PAIR") && BH_XENV("BIRDA_HIP_MEL_PAIR")[0] == '0';
    const int pair_group = 8 * p.n_branches;
    const int paired = (!pair_off && p.n_branches > 1 && n_wg >= pair_group) ? 1 : 0;
    if (paired) n_wg -= n_wg % pair_group;
    dim3 grid((unsigned)n_wg), block(256);
#ifdef BIRDA_HIP_EXPERIMENTS
    if (two) {
        static DeviceOnce attr2;
        attr2.run([] { (void)hipFuncSetAttribute((const void *)mel_kernel<6, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        hipLaunchKernelGGL((mel_kernel<6, 3, 2>), grid, dim3(512), smem, s, x, minmax, spec, d_p, p.br[0].gf, p.br[1].gf, p.br[2].gf, p.br[3].gf,
                           dbg, n_tiles, n_items, paired);
        return;
    }
#endif
#define BH_MEL_CASE(MTV)                                                                                   \
    case MTV: {                                                                                            \
        static DeviceOnce attr_set;                                                                        \
        attr_set.run([] {                                                                                  \
            (void)hipFuncSetAttribute((const void *)mel_kernel<MTV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                              \
            (void)hipFuncSetAttribute((const void *)mel_kernel<MTV, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                              \
        });                                                                                                \
        if (p.prec == 3)                                                                                   \
            hipLaunchKernelGGL((mel_kernel<MTV, 3>), grid, block, smem, s, x, minmax, spec, d_p, p.br[0].gf,   \
                               p.br[1].gf, p.br[2].gf, p.br[3].gf, dbg, n_tiles, n_items, paired);                 \
        else                                                                                               \
            hipLaunchKernelGGL((mel_kernel<MTV, 0>), grid, block, smem, s, x, minmax, spec, d_p, p.br[0].gf,   \
                               p.br[1].gf, p.br[2].gf, p.br[3].gf, dbg, n_tiles, n_items, paired);                 \
    } break;
    switch (mt) {      // (32, 96 and 128 mels are the published families'; 48 / 64 / 80 / 112 -- round 6 -- whatever else a model file holds)
        BH_MEL_CASE(2)
        BH_MEL_CASE(3)
        BH_MEL_CASE(4)
        BH_MEL_CASE(5)
        BH_MEL_CASE(6)
        BH_MEL_CASE(7)
        BH_MEL_CASE(8)
    default: break;
    }
#undef BH_MEL_CASE
}

}  // namespace bh
