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
#include "kernels.hpp"

namespace bh {

constexpr int MM_SPLIT = 8;  // partial min/max blocks per segment

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
                                                      int sample_count) {
    const int seg = blockIdx.y, part = blockIdx.x;
    const int n4 = sample_count >> 2;
    const int per = (n4 + MM_SPLIT - 1) / MM_SPLIT;
    const int lo = part * per, hi = min(n4, lo + per);
    const float4 *p = reinterpret_cast<const float4 *>(x + (size_t)seg * sample_count);
    float mn = INFINITY, mx = -INFINITY;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float4 v = p[i];
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    if (part == MM_SPLIT - 1)  // scalar tail when sample_count % 4 != 0
        for (int i = (n4 << 2) + threadIdx.x; i < sample_count; i += 256) {
            float v = x[(size_t)seg * sample_count + i];
            mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
    mn = wave_min(mn); mx = wave_max(mx);
    __shared__ float s[8];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s[w] = mn; s[4 + w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(s[0], s[1]), fminf(s[2], s[3]));
        mx = fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7]));
        mm[((size_t)seg * MM_SPLIT + part) * 2 + 0] = mn;
        mm[((size_t)seg * MM_SPLIT + part) * 2 + 1] = mx;
    }
}

void launch_minmax(const float *x, float *minmax, int n_seg, int sample_count, hipStream_t s) {
    hipLaunchKernelGGL(minmax_kernel, dim3(MM_SPLIT, n_seg), dim3(256), 0, s, x, minmax, sample_count);
}

// ---------------------------------------------------------------------------------------
// mel kernel: grid (frame tiles of TN, n_branches, n_seg), block 256 = 4 waves.
// Wave w owns frames [16w, 16w+16) of the tile and all MT mel tiles (MT x 4 accumulators).
// LDS: the tile's normalised sample span xs[(TN-1)H + L] + a double-buffered KC-row slab of
// Gf.  D = A.B with A = Gf^T (mel on M, lane&15), B = folded frames (frame on N, lane&15).
// ---------------------------------------------------------------------------------------
constexpr int MEL_TN = 64;
constexpr int MEL_KC = 32;

template <int MT>
__global__ __launch_bounds__(256) void mel_kernel(const float *__restrict__ x, const float *__restrict__ mm,
                                                   float *__restrict__ spec, const FrontendParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BranchParams &bp = p.br[blockIdx.y];
    const int seg = blockIdx.z;
    const int t0 = blockIdx.x * MEL_TN;
    const int L = bp.L, H = bp.H, K = bp.K;
    const int S = p.sample_count;
    constexpr int NMP = MT * 16;
    constexpr int GS = NMP + 16;  // slab row stride: k-rows 4q apart land 16 banks apart
    const int span = (MEL_TN - 1) * H + L;
    const int span_pad = (span + 3) & ~3;
    float *xs = smem;
    float *gs = smem + span_pad;  // [2][MEL_KC][GS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;

    // segment min/max from the MM_SPLIT partials
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MM_SPLIT; i++) {
        mn = fminf(mn, mm[((size_t)seg * MM_SPLIT + i) * 2]);
        mx = fmaxf(mx, mm[((size_t)seg * MM_SPLIT + i) * 2 + 1]);
    }
    const float denom = (mx - mn) + p.norm_eps;

    // stage the normalised span (16-B loads; the span start t0*H is a multiple of 4 samples)
    const float *xseg = x + (size_t)seg * S;
    const int g0 = t0 * H;
    for (int i = tid * 4; i < span_pad; i += 256 * 4) {
        float v[4];
        if (g0 + i + 3 < S) {
            float4 q = *reinterpret_cast<const float4 *>(xseg + g0 + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (g0 + i + e < S) ? xseg[g0 + i + e] : mn;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = ((v[e] - mn) / denom - 0.5f) * 2.0f;
        *reinterpret_cast<float4 *>(xs + i) = make_float4(v[0], v[1], v[2], v[3]);
    }

    // Gf slab staging: chunk c is MEL_KC contiguous rows of NMP floats
    constexpr int SLAB4 = MEL_KC * NMP / 4;           // float4 per slab
    constexpr int PER_T = (SLAB4 + 255) / 256;        // float4 per thread
    const float4 *gf4 = reinterpret_cast<const float4 *>(bp.gf);
    float4 pre[PER_T];
    auto slab_load = [&](int c) {
#pragma unroll
        for (int i = 0; i < PER_T; i++) {
            int f = tid + 256 * i;
            if (f < SLAB4) pre[i] = gf4[(size_t)c * SLAB4 + f];
        }
    };
    auto slab_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER_T; i++) {
            int f = tid + 256 * i;
            if (f < SLAB4) {
                int row = (f * 4) / NMP, col = (f * 4) % NMP;
                *reinterpret_cast<float4 *>(gs + (size_t)buf * MEL_KC * GS + row * GS + col) = pre[i];
            }
        }
    };
    slab_load(0);
    slab_store(0);
    __syncthreads();

    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fb = (wave * 16 + li) * H;  // this lane's frame origin inside the span
    const int nchunks = K / MEL_KC;
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunks) slab_load(c + 1);
        const float *g = gs + (size_t)buf * MEL_KC * GS;
#pragma unroll
        for (int kk = 0; kk < MEL_KC / 4; kk++) {
            const int j = c * MEL_KC + kk * 4 + kq;
            const float b = xs[fb + j + 1] + xs[fb + L - 1 - j];
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const float a = g[(kk * 4 + kq) * GS + m * 16 + li];
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) slab_store(buf ^ 1);
        __syncthreads();
    }

    // epilogue: square, power law, folded-BN affine, mel flip, [mel][time] store
    const int t = t0 + wave * 16 + li;
    if (t < bp.n_frames) {
        float *out = spec + ((size_t)seg * p.n_branches + blockIdx.y) * bp.n_mels * bp.n_frames;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int mel = m * 16 + kq * 4 + r;
                if (mel < bp.n_mels) {
                    const float v = acc[m][r];
                    float o = powf(v * v, bp.expo);
                    o = o * bp.out_scale + bp.out_shift;
                    const int row = bp.flip ? (bp.n_mels - 1 - mel) : mel;
                    out[(size_t)row * bp.n_frames + t] = o;
                }
            }
    }
}

void launch_mel(const float *x, const float *minmax, float *spec, const FrontendParams &p, int n_seg,
                hipStream_t s) {
    int max_span = 0, max_frames = 0, nmp = p.br[0].nm_pad;
    for (int b = 0; b < p.n_branches; b++) {
        int span = (MEL_TN - 1) * p.br[b].H + p.br[b].L;
        max_span = span > max_span ? span : max_span;
        max_frames = p.br[b].n_frames > max_frames ? p.br[b].n_frames : max_frames;
    }
    const int span_pad = (max_span + 3) & ~3;
    const size_t smem = ((size_t)span_pad + 2 * MEL_KC * (nmp + 16)) * sizeof(float);
    dim3 grid((max_frames + MEL_TN - 1) / MEL_TN, p.n_branches, n_seg), block(256);
    const int mt = nmp / 16;
#define BH_MEL_CASE(MTV)                                                                                   \
    case MTV: {                                                                                            \
        static bool attr_set = false;                                                                      \
        if (!attr_set) {                                                                                   \
            (void)hipFuncSetAttribute((const void *)mel_kernel<MTV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                              \
            attr_set = true;                                                                               \
        }                                                                                                  \
        hipLaunchKernelGGL(mel_kernel<MTV>, grid, block, smem, s, x, minmax, spec, p);                     \
    } break;
    switch (mt) {
        BH_MEL_CASE(2)
        BH_MEL_CASE(6)
        BH_MEL_CASE(8)
    default: break;
    }
#undef BH_MEL_CASE
}

}  // namespace bh
