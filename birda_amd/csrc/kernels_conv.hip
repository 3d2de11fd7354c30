// Conv-stack kernels for gfx950 (NHWC f32, BN folded into weights + bias).
//
// These replace the ORT kernels the reference reaches through birdnet_onnx::Classifier
// (reference src/inference/classifier.rs:478-488): Conv (group = 1 / group = C),
// activation, residual Add, GlobalAveragePool, Gemm.
//   * pointwise 1x1 conv and the dense head: LDS-tiled GEMM on v_mfma_f32_16x16x4_f32
//   * depthwise kxk: direct NHWC conv, 4 channels per lane (16-B loads/stores)
//   * stem conv (tiny Cin): direct conv, weights in LDS, 8 output channels per lane
#include "kernels.hpp"

namespace bh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) { return act_apply_slow(v, act); }

// ---------------------------------------------------------------------------------------
// Pointwise / dense GEMM.  C[M][N] = act(A[M][K] W[K][ldw] + bias[N]) (+ R[M][N]).
// Block = 256 threads = 4 waves stacked along M; wave tile (BM/4) x (16 NT).
// K is consumed in chunks of 32 (whole 128-B lines of every A row), staged through LDS with
// a register prefetch of the next chunk.  Inside a 16-deep group the k order is permuted
// (MFMA step c of lane-quad q uses k = 16g + 4q + c) so that a lane's four A operands are
// one ds_read_b128; the B operand applies the same permutation through its LDS row index.
// ---------------------------------------------------------------------------------------
constexpr int PW_BK = 32;

// GATE (round 5, squeeze-excite blocks): A is the depthwise output D [n][P][K] and every element is multiplied by its segment's
// gate[n][K] on the way into LDS -- the ONNX Mul between the depthwise and the project convolution, without a pass of its own.
template <int BM, int NT, bool GATE = false>
__global__ __launch_bounds__(256) void pw_gemm_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                       const float *__restrict__ bias, const float *__restrict__ R,
                                                       float *__restrict__ C, int M, int K, int N, int ldw, int act,
                                                       const float *__restrict__ gate = nullptr, int rows_per_seg = 1) {
    constexpr int BN = NT * 16;
    constexpr int WM = BM / 4;
    constexpr int MT = WM / 16;
    constexpr int AS = PW_BK + 4;  // As row stride (floats): 16-B aligned rows, odd 16-B slot stride
    constexpr int BS = BN + 4;     // Bs row stride: rows 4 apart land 16 banks apart
    constexpr int A4 = BM * PW_BK / 4 / 256;               // float4 per thread per chunk (A)
    constexpr int B4 = (PW_BK * BN / 4 + 255) / 256;       // float4 per thread per chunk (W)
    __shared__ __attribute__((aligned(16))) float As[2][BM * AS];
    __shared__ __attribute__((aligned(16))) float Bs[2][PW_BK * BS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    float4 pa[A4], pb[B4];
    const float *grow[GATE ? A4 : 1];    // GATE: the gate row of each of this thread's A rows
    if constexpr (GATE) {
#pragma unroll
        for (int i = 0; i < A4; i++) grow[i] = gate + (size_t)(min(m0 + ((tid + 256 * i) >> 3), M - 1) / rows_per_seg) * K;
    }
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A4; i++) {
            const int f = tid + 256 * i;
            const int row = f >> 3, kc = (f & 7) << 2;
            const int gm = m0 + row, gk = k0 + kc;
            pa[i] = (gm < M && gk < K) ? *reinterpret_cast<const float4 *>(A + (size_t)gm * K + gk)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (GATE) {
                const float4 g = gk < K ? *reinterpret_cast<const float4 *>(grow[i] + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
                pa[i].x *= g.x; pa[i].y *= g.y; pa[i].z *= g.z; pa[i].w *= g.w;
            }
        }
#pragma unroll
        for (int i = 0; i < B4; i++) {
            const int f = tid + 256 * i;
            const int row = f / (BN / 4), nc = (f % (BN / 4)) << 2;
            const int gk = k0 + row, gn = n0 + nc;
            pb[i] = (f < PW_BK * BN / 4 && gk < K && gn < ldw)
                        ? *reinterpret_cast<const float4 *>(W + (size_t)gk * ldw + gn)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A4; i++) {
            const int f = tid + 256 * i;
            const int row = f >> 3, kc = (f & 7) << 2;
            *reinterpret_cast<float4 *>(&As[buf][row * AS + kc]) = pa[i];
        }
#pragma unroll
        for (int i = 0; i < B4; i++) {
            const int f = tid + 256 * i;
            if (f < PW_BK * BN / 4) {
                const int row = f / (BN / 4), nc = (f % (BN / 4)) << 2;
                *reinterpret_cast<float4 *>(&Bs[buf][row * BS + nc]) = pb[i];
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    const int nchunks = (K + PW_BK - 1) / PW_BK;
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk((c + 1) * PW_BK);
        const int kleft = K - c * PW_BK;
        const int groups = kleft >= PW_BK ? 2 : (kleft + 15) >> 4;
        const float *as = &As[buf][(wave * WM + li) * AS + 4 * kq];
        const float *bs = &Bs[buf][(4 * kq) * BS + li];
        for (int g = 0; g < groups; g++) {
            float4 a4[MT];
#pragma unroll
            for (int i = 0; i < MT; i++) a4[i] = *reinterpret_cast<const float4 *>(as + i * 16 * AS + g * 16);
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                float b[NT];
#pragma unroll
                for (int j = 0; j < NT; j++) b[j] = bs[(g * 16 + cc) * BS + j * 16];
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    const float a = cc == 0 ? a4[i].x : cc == 1 ? a4[i].y : cc == 2 ? a4[i].z : a4[i].w;
#pragma unroll
                    for (int j = 0; j < NT; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D row = 4*(lane>>4) + r, col = lane & 15
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int col = n0 + j * 16 + li;
        if (col >= N) continue;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + wave * WM + i * 16 + kq * 4 + r;
                if (row < M) {
                    float v = act_apply(acc[i][j][r] + bv, act);
                    if (R) v += R[(size_t)row * N + col];
                    C[(size_t)row * N + col] = v;
                }
            }
    }
}

template <int BM, int NT>
static void pw_launch(const float *A, const float *W, const float *bias, const float *R, float *C, int M, int K,
                      int N, int ldw, int act, hipStream_t s, const float *gate = nullptr, int rows_per_seg = 1) {
    dim3 grid((N + NT * 16 - 1) / (NT * 16), (M + BM - 1) / BM), block(256);
    if (gate) hipLaunchKernelGGL((pw_gemm_kernel<BM, NT, true>), grid, block, 0, s, A, W, bias, R, C, M, K, N, ldw, act, gate, rows_per_seg);
    else hipLaunchKernelGGL((pw_gemm_kernel<BM, NT, false>), grid, block, 0, s, A, W, bias, R, C, M, K, N, ldw, act, nullptr, 1);
}

// pick the widest column tile that wastes the fewest padded columns
static int pick_nt(int N) {
    int best = 1;
    long best_pad = -1;
    for (int nt = 8; nt >= 1; nt--) {
        long bn = nt * 16;
        long padded = ((N + bn - 1) / bn) * bn;
        if (best_pad < 0 || padded < best_pad) { best_pad = padded; best = nt; }
    }
    return best;
}

void launch_pw_gemm(const float *A, const float *W, const float *bias, const float *R, float *C, int M, int K,
                    int N, int ldw, int act, hipStream_t s) {
    launch_pw_gemm_gated(A, nullptr, 1, W, bias, R, C, M, K, N, ldw, act, s);
}

// the same GEMM with A = D x gate (gate [M / rows_per_seg][K], nullptr: plain): the project convolution of a squeeze-excite block
void launch_pw_gemm_gated(const float *A, const float *gate, int rows_per_seg, const float *W, const float *bias, const float *R,
                          float *C, int M, int K, int N, int ldw, int act, hipStream_t s) {
    const int nt = pick_nt(N);
    const long blocks128 = (long)((M + 127) / 128) * ((N + nt * 16 - 1) / (nt * 16));
    const bool small = blocks128 < 512;  // keep >= 2 blocks per CU in flight when M is short
#define BH_PW_CASE(NTV)                                                                    \
    case NTV:                                                                              \
        if (small) pw_launch<64, NTV>(A, W, bias, R, C, M, K, N, ldw, act, s, gate, rows_per_seg);             \
        else pw_launch<128, NTV>(A, W, bias, R, C, M, K, N, ldw, act, s, gate, rows_per_seg);                  \
        break;
    switch (nt) {
        BH_PW_CASE(1) BH_PW_CASE(2) BH_PW_CASE(3) BH_PW_CASE(4)
        BH_PW_CASE(5) BH_PW_CASE(6) BH_PW_CASE(7) BH_PW_CASE(8)
    }
#undef BH_PW_CASE
}

// ---------------------------------------------------------------------------------------
// Split-f16 GEMM (BH_FLAG_F16X3 / BH_FLAG_F16) for the layers outside the fused blocks (head 1x1 conv,
// dense): C[M][N] = act(A[M][K] W[K][N] + bias) (+ R), K % 32 == 0.  Block = 4 waves as 2 x 2, wave
// tile 64 x 64 (4 x 4 MFMA tiles), block tile 128 x 128.  A rows go global -> registers -> hi / lo
// f16 fragments (8 consecutive k per lane); W is pre-split on the host in fragment-major planes
// [k step][column tile]{hi, lo}[64 lanes][8 halves] and streams from L2.  Two register sets alternate
// so that every load is a full k step (48 MFMAs) ahead of its use.
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TERMS, int ACT>
__global__ __launch_bounds__(256) void pw_gemm16_kernel(const float *__restrict__ A, const f16x8 *__restrict__ Wf,
                                                         const float *__restrict__ bias, const float *__restrict__ R,
                                                         float *__restrict__ C, int M, int K, int N, int n_tiles, float w_unscale) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    // Block -> (column block, row block), XCD-aware: workgroup ids go round-robin over the 8 XCDs, each with its own L2.  XCD k
    // takes the column blocks k, k + 8, ... and walks ALL row blocks of one before the next, so the eight row blocks that read the
    // same 128 columns of W run side by side on one L2 and W comes from HBM once.  (Column block fastest, as a plain 2-D grid does
    // it, re-read the Perch-sized model's 363-MB dense plane once per row block: 2.9 GB per launch.)
    const int n_xb = (n_tiles + 7) / 8, n_yb = (M + 127) / 128;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int yb = slot % n_yb, xb = (slot / n_yb) * 8 + xcd;
    if (xb >= n_xb) return;
    const int m0 = yb * 128 + wm * 64, t0 = xb * 8 + wn * 4;   // first row, first column tile
    const int steps = K / 32;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float *arow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) arow[i] = A + (size_t)min(m0 + i * 16 + li, M - 1) * K + 8 * kq;   // rows past M: clamped, never stored

    float4 ra0[4][2], ra1[4][2];
    f16x8 bh0[4], bl0[4], bh1[4], bl1[4];
    auto load = [&](int st, float4 (&ra)[4][2], f16x8 (&bh)[4], f16x8 (&bl)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ra[i][0] = *reinterpret_cast<const float4 *>(arow[i] + 32 * st);
            ra[i][1] = *reinterpret_cast<const float4 *>(arow[i] + 32 * st + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int t = min(t0 + j, n_tiles - 1);
            bh[j] = Wf[(((size_t)st * n_tiles + t) * 2 + 0) * 64 + lane];
            if (TERMS == 3) bl[j] = Wf[(((size_t)st * n_tiles + t) * 2 + 1) * 64 + lane];
        }
    };
    auto step = [&](const float4 (&ra)[4][2], const f16x8 (&bh)[4], const f16x8 (&bl)[4]) {
        f16x8 ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float v[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w, ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
            bh_split8(v, ah[i], al[i]);
        }
        __builtin_amdgcn_sched_barrier(0);   // no VALU split between the MFMAs (see mel_kernel)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                if (TERMS == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
    };
    load(0, ra0, bh0, bl0);
    for (int st = 0; st < steps; st += 2) {
        if (st + 1 < steps) load(st + 1, ra1, bh1, bl1);   // (never a prefetch nobody consumes)
        __builtin_amdgcn_sched_barrier(0);
        step(ra0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < steps) load(st + 2, ra0, bh0, bl0);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < steps) step(ra1, bh1, bl1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int col = (t0 + j) * 16 + li;
        if (col >= N) continue;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + i * 16 + kq * 4 + r;
                if (row < M) {
                    // the activation is a template argument: a run-time switch inlined 64 times bloats
                    // the kernel past the instruction cache
                    float v = __builtin_fmaf(acc[i][j][r], w_unscale, bv);   // the planes hold W / w_unscale
                    v = bh_act<ACT>(v);
                    if (R) v += R[(size_t)row * N + col];
                    C[(size_t)row * N + col] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------
// The same GEMM for a handful of rows (M <= 16: the dense layer of a forward of one to sixteen segments; round 6).  With one row
// tile there is nothing to share and almost nothing to compute: the layer is its weights streamed once (26.7 MB for the 1 024 ->
// 6 522 head), and pw_gemm16_kernel, one 32-deep step ahead on 51 workgroups, waited a memory round trip per step -- 32 of them,
// 35 us, for a launch that moves its bytes in 5.  Here a wave owns TWO column tiles and keeps EIGHT steps of their fragments in
// flight (128 registers), four waves a workgroup, ceil(n_tiles / 8) workgroups.  Per element the same products in the same order
// as the other two kernels: the same bits.
// ---------------------------------------------------------------------------------------
template <int TERMS, int ACT>
__global__ __launch_bounds__(256) void pw_gemm16_skinny_kernel(const float *__restrict__ A, const f16x8 *__restrict__ Wf,
                                                                const float *__restrict__ bias, const float *__restrict__ R,
                                                                float *__restrict__ C, int M, int K, int N, int n_tiles, float w_unscale) {
    constexpr int PF = 8, NT = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int t0 = ((int)blockIdx.x * 4 + wave) * NT;
    if (t0 >= n_tiles) return;
    const int steps = K / 32;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *arow = A + (size_t)min(li, M - 1) * K + 8 * kq;   // rows past M: clamped, never stored
    float4 ra[PF][2];
    f16x8 bh[PF][NT], bl[PF][NT];
    auto load = [&](int st, float4 (&a)[2], f16x8 (&h)[NT], f16x8 (&l)[NT]) {
        a[0] = *reinterpret_cast<const float4 *>(arow + 32 * st);
        a[1] = *reinterpret_cast<const float4 *>(arow + 32 * st + 4);
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int t = min(t0 + j, n_tiles - 1);
            h[j] = Wf[(((size_t)st * n_tiles + t) * 2 + 0) * 64 + lane];
            if (TERMS == 3) l[j] = Wf[(((size_t)st * n_tiles + t) * 2 + 1) * 64 + lane];
        }
    };
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (u < steps) load(u, ra[u], bh[u], bl[u]);
    for (int st0 = 0; st0 < steps; st0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int st = st0 + u;
            if (st < steps) {       // (wave-uniform)
                const float v[8] = {ra[u][0].x, ra[u][0].y, ra[u][0].z, ra[u][0].w, ra[u][1].x, ra[u][1].y, ra[u][1].z, ra[u][1].w};
                f16x8 ah, al;
                bh_split8(v, ah, al);
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u][j], acc[j], 0, 0, 0);
                    if (TERMS == 3) {
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u][j], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u][j], acc[j], 0, 0, 0);
                    }
                }
                if (st + PF < steps) load(st + PF, ra[u], bh[u], bl[u]);   // (the slot's registers have been read)
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int col = (t0 + j) * 16 + li;
        if (col >= N) continue;
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = kq * 4 + r;
            if (row < M) {
                float v = __builtin_fmaf(acc[j][r], w_unscale, bv);
                v = bh_act<ACT>(v);
                if (R) v += R[(size_t)row * N + col];
                C[(size_t)row * N + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same GEMM with both operands staged through LDS (round 4).  pw_gemm16_kernel streams every wave's operand fragments
// straight from L2: 16 KB per wave and 32-deep step for 48 MFMAs, 58 B/clk per CU with two workgroups resident -- the fabric
// delivers about that much and no more, so the matrix pipe idled at ~20 % (the Perch-sized model's dense pair: 1.07 us per
// segment against an MFMA floor of 0.24).  Here a workgroup's 128 x 128 tile reads each operand ONCE per step: the 128 A rows
// are loaded by all 256 threads (two (row, 8-k group) pairs each), split into f16 hi / lo ONCE (half the splits of the streaming
// kernel, where the two waves of a row pair each split the same rows) and written to LDS in MFMA-fragment order; the eight column
// tiles' W fragments, which already are fragment-major 1-KiB planes, arrive by LDS-DMA, four pieces a wave.  Two stages: step
// s + 1 is loaded while step s computes, one barrier per step.  32 KB of L2 traffic per workgroup-step instead of 64, and the
// waves read 16 KB each from LDS (85 B/clk per workgroup, two per CU).
// ---------------------------------------------------------------------------------------
// GATE (round 5): the project convolution of a squeeze-excite block -- A is the depthwise output D [n][P][K], multiplied by its
// segment's gate[n][K] in registers before the f16 split (the graph's Mul, without a pass of its own), and K may be any multiple
// of 4 (the expanded widths of EfficientNet stacks: 144, 816, 1 392 ...): the last 32-deep step loads zeros beyond K, the planes
// are zero-padded on the host.
// NTB: column tiles (of 16) per workgroup -- 8 for the dense layers; the gated project convolutions pick 6, 8 or 10 to fit N = 96, 136,
// 232, 384 with little padding (wave (wm, wn) owns 64 rows x NTB / 2 column tiles)
template <int TERMS, int ACT, bool GATE = false, int NTB = 8, bool BLK = false>
__global__ __launch_bounds__(256, 2) void pw_gemm16s_kernel(const float *__restrict__ A, const f16x8 *__restrict__ Wf,
                                                            const float *__restrict__ bias, const float *__restrict__ R,
                                                            float *__restrict__ C, int M, int K, int N, int n_tiles, float w_unscale,
                                                            const float *__restrict__ gate = nullptr, int rows_per_seg = 1) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];   // [2 stages]{A: [8 row tiles]{hi, lo}[64 lanes][4], B: [NTB column tiles]{hi, lo}[64][4]}
    constexpr int STAGE = (8 + NTB) * 2 * 256;                    // floats per stage (32 KB at NTB = 8)
    constexpr int NJ = NTB / 2;                                   // column tiles per wave
    static_assert(NTB % 2 == 0 && NTB >= 2 && NTB <= 12, "column tiles per workgroup");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_xb = (n_tiles + NTB - 1) / NTB, n_yb = (M + 127) / 128;   // (block order: see pw_gemm16_kernel)
    int yb, xb;
    if constexpr (GATE) {
        // few column blocks (a project convolution: N = 24 .. 384) and very many row blocks: dealt in launch order, which
        // spreads the row blocks over the eight XCDs (the dense layers' order would leave 8 - n_xb of them idle); W is
        // small and lives in every L2
        yb = (int)blockIdx.x / n_xb; xb = (int)blockIdx.x - yb * n_xb;
        if (yb >= n_yb) return;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        yb = slot % n_yb; xb = (slot / n_yb) * 8 + xcd;
        if (xb >= n_xb) return;
    }
    const int mb0 = yb * 128, tb0 = xb * NTB;
    const int steps = GATE ? (K + 31) / 32 : K / 32;
    const unsigned lds0 = (__builtin_amdgcn_groupstaticsize() + 15u) & ~15u;
    const int ws = __builtin_amdgcn_readfirstlane(wave);

    // staging roles: pair q (0, 1) of this thread is A row (tid >> 2) + 64 q, k group tid & 3
    const int srow = tid >> 2, skq = tid & 3;
    const float *ag[2], *gg[2];
    int adst[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int row = srow + 64 * q, grow = min(mb0 + row, M - 1);
        // rows past M: clamped, never stored.  (BLK: MbDesc::dblk's layout -- the 8 floats sit in block 2 st + (skq >> 1), row grow & 15)
        ag[q] = BLK ? A + (size_t)(grow >> 4) * 16 * K + ((skq >> 1) * 16 + (grow & 15)) * 16 + (skq & 1) * 8 : A + (size_t)grow * K + 8 * skq;
        gg[q] = GATE ? gate + (size_t)(grow / rows_per_seg) * K + 8 * skq : nullptr;
        adst[q] = (((row >> 4) * 2) * 64 + skq * 16 + (row & 15)) * 4;           // float offset of the hi fragment slot; lo: + 256
    }
    float4 ra[2][2], rg[GATE ? 2 : 1][2];
    constexpr int a_step = BLK ? 512 : 32;                // floats from one 32-deep step of a row to the next
    auto load_a = [&](int st) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if constexpr (GATE) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const bool in = 32 * st + 8 * skq + 4 * h < K;     // (K % 4 == 0: a quad is wholly inside or outside)
                    ra[q][h] = *reinterpret_cast<const float4 *>(ag[q] + (in ? a_step * st + 4 * h : 0));
                    rg[q][h] = *reinterpret_cast<const float4 *>(gg[q] + (in ? 32 * st + 4 * h : 0));
                    if (!in) { rg[q][h] = make_float4(0.f, 0.f, 0.f, 0.f); ra[q][h] = rg[q][h]; }
                }
            } else {
                ra[q][0] = *reinterpret_cast<const float4 *>(ag[q] + 32 * st);
                ra[q][1] = *reinterpret_cast<const float4 *>(ag[q] + 32 * st + 4);
            }
        }
    };
    auto store_a = [&](int buf) {
        float *as = gsm + buf * STAGE;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            float v[8] = {ra[q][0].x, ra[q][0].y, ra[q][0].z, ra[q][0].w, ra[q][1].x, ra[q][1].y, ra[q][1].z, ra[q][1].w};
            if constexpr (GATE) {
                const float g[8] = {rg[q][0].x, rg[q][0].y, rg[q][0].z, rg[q][0].w, rg[q][1].x, rg[q][1].y, rg[q][1].z, rg[q][1].w};
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] *= g[j];
            }
            f16x8 h, l;
            bh_split8(v, h, l);
            *reinterpret_cast<f16x8 *>(as + adst[q]) = h;
            if (TERMS == 3) *reinterpret_cast<f16x8 *>(as + adst[q] + 256) = l;
        }
    };
    auto dma_b = [&](int st, int buf) {   // 2 NTB (NTB) pieces of 1 KiB: wave w takes column tiles w, w + 4, ...
#pragma unroll
        for (int jj = 0; jj < (NTB + 3) / 4; jj++) {
            const int j = ws + 4 * jj, t = min(tb0 + j, n_tiles - 1);
            if (j >= NTB) continue;     // (wave-uniform)
#pragma unroll
            for (int pl = 0; pl < (TERMS == 3 ? 2 : 1); pl++) {
                const f16x8 *src = Wf + (((size_t)st * n_tiles + t) * 2 + pl) * 64;
                const unsigned dst = lds0 + 4u * (unsigned)(buf * STAGE + 8 * 2 * 256 + (j * 2 + pl) * 256);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"((unsigned)lane * 16u), "s"(src), "s"(dst) : "memory", "m0");
            }
        }
    };

    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    dma_b(0, 0);
    load_a(0);
    store_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int st = 0; st < steps; st++) {
        const int cur = st & 1;
        const bool more = st + 1 < steps;   // (uniform)
        if (more) { dma_b(st + 1, cur ^ 1); load_a(st + 1); }
        const float *as = gsm + cur * STAGE, *bs = as + 8 * 2 * 256;
        f16x8 ah[4], al[4], bh[NJ], bl[NJ];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            ah[i] = *reinterpret_cast<const f16x8 *>(as + (((wm * 4 + i) * 2) * 64 + lane) * 4);
            if (TERMS == 3) al[i] = *reinterpret_cast<const f16x8 *>(as + (((wm * 4 + i) * 2 + 1) * 64 + lane) * 4);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            bh[j] = *reinterpret_cast<const f16x8 *>(bs + (((wn * NJ + j) * 2) * 64 + lane) * 4);
            if (TERMS == 3) bl[j] = *reinterpret_cast<const f16x8 *>(bs + (((wn * NJ + j) * 2 + 1) * 64 + lane) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                if (TERMS == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
            }
        if (more) store_a(cur ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int m0 = mb0 + wm * 64, t0 = tb0 + wn * NJ;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int col = (t0 + j) * 16 + li;
        if (col >= N) continue;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + i * 16 + kq * 4 + r;
                if (row < M) {
                    float v = __builtin_fmaf(acc[i][j][r], w_unscale, bv);   // the planes hold W / w_unscale
                    v = bh_act<ACT>(v);
                    if (R) v += R[(size_t)row * N + col];
                    C[(size_t)row * N + col] = v;
                }
            }
    }
}

bool pw_gemm16_supports(int K, int act) { return K % 32 == 0 && (act == ACT_NONE || act_is_templated(act)); }

// ---------------------------------------------------------------------------------------
// The gated project convolution of the EARLY squeeze-excite blocks: very many rows (16 000 pixels a segment), few columns
// (N = 24 .. 48), K = 24 .. 288 -- a stream of D rows through a small weight matrix, bound by HBM (K + 2 N floats per pixel),
// not a 128 x 128 GEMM tile.  One 8-wave workgroup keeps ALL of W (f16 hi / lo fragments, [K / 32][NT]{hi, lo}[64][8]: <= 54 KB)
// in LDS; every wave walks 16-row tiles: a lane's 8 consecutive k of its row arrive as two 16-byte loads (the four lanes of a row
// read one whole 128-byte line), are multiplied by the segment's gate, split into f16 hi / lo in registers and go straight into
// the MFMA as the A operand -- no staging of A at all.  The next k step's loads are in flight while this one computes.
// ---------------------------------------------------------------------------------------
// (D is read exactly once by these kernels, yet NON-TEMPORAL loads of it are slower, measured: 905 -> 1 625, 651 -> 1 087, 353 -> 416 us
//  per 1 000 segments -- pass A wrote those lines microseconds earlier and much of them still sits in the write-back L2 / Infinity
//  Cache, which a streaming load goes around.)
#define BH_LOAD_STREAM(p) (*(p))
// SHALLOW (K <= 32: one k step a pass, the 24 -> 24 block): 3 (2) row tiles a wave in 128 registers, so that TWO workgroups share a CU
// -- a pass is load -> wait -> 6 MFMAs -> residual -> wait -> store, and only more waves put anything under those waits: 1 410 -> 1 000 us
// per 1 000 segments; deeper layers have their own next step to wait under and lose 0-5 % that way (tools/microbench/gated_gemm.hip).
template <int TERMS, int NT, bool SHALLOW = false, int DBG = 0>      // (DBG: the microbench's switches -- 1 no MFMAs, 8 no epilogue; the product instantiates 0)
__global__ __launch_bounds__(512, SHALLOW ? 4 : 2) void pw_gemm16_thin_kernel(const float *__restrict__ A, const float *__restrict__ gate, int rows_per_seg,
                                                                 const f16x8 *__restrict__ Wf, const float *__restrict__ bias,
                                                                 const float *__restrict__ R, float *__restrict__ C, int M, int K, int N,
                                                                 float w_unscale) {
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int steps = (K + 31) / 32;
    {   // all of W, once per workgroup
        const float4 *src = reinterpret_cast<const float4 *>(Wf);
        float4 *dst = reinterpret_cast<float4 *>(tsm);
        const int n4 = steps * NT * 2 * 64;            // float4s: [step][tile]{hi, lo}[64 lanes] x 16 B (8 halves)
        for (int i = tid; i < n4; i += 512) dst[i] = src[i];
    }
    __syncthreads();
    const f16x8 *wf = reinterpret_cast<const f16x8 *>(tsm);
    const int n_rt = (M + 15) >> 4;
    const float rcp_p = 1.0f / (float)rows_per_seg;
    // RB row tiles per wave and pass, their loads issued together: one tile at a time exposed a whole HBM round trip per 16 rows
    // (K = 24: one k step, nothing to prefetch under) -- 1.9 ms per 1 000 segments for the 24 -> 24 block, 2.4 TB/s
    constexpr int RB = SHALLOW ? (NT <= 2 ? 3 : 2) : (NT <= 2 ? 4 : 3);
    for (int rt0 = (blockIdx.x * 8 + wave) * RB; rt0 < n_rt; rt0 += gridDim.x * 8 * RB) {
        const float *ap[RB], *gp[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const int row = min((rt0 + r) * 16 + li, M - 1);                // rows past M: clamped, never stored
            int seg = (int)((float)row * rcp_p);                            // row / rows_per_seg through the reciprocal, fixed up
            seg += (row - seg * rows_per_seg >= rows_per_seg) ? 1 : 0;
            seg -= (row - seg * rows_per_seg < 0) ? 1 : 0;
            ap[r] = A + (size_t)row * K + 8 * kq;
            gp[r] = gate + (size_t)seg * K + 8 * kq;
        }
        f32x4 acc[RB][NT];
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 ra[RB][2], rg[RB][2];    // (no second register set: RB tiles x 8 waves in flight per CU cover the round trip)
        auto load = [&](int st, float4 (&a)[RB][2], float4 (&g)[RB][2]) {
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const bool in = 32 * st + 8 * kq + 4 * h < K;       // (K % 4 == 0)
                    a[r][h] = BH_LOAD_STREAM(reinterpret_cast<const float4 *>(ap[r] + (in ? 32 * st + 4 * h : 0)));
                    g[r][h] = *reinterpret_cast<const float4 *>(gp[r] + (in ? 32 * st + 4 * h : 0));
                    if (!in) { a[r][h] = make_float4(0.f, 0.f, 0.f, 0.f); g[r][h] = a[r][h]; }
                }
        };
        for (int st = 0; st < steps; st++) {
            load(st, ra, rg);
            f16x8 ah[RB], al[RB];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                const float v[8] = {ra[r][0].x * rg[r][0].x, ra[r][0].y * rg[r][0].y, ra[r][0].z * rg[r][0].z, ra[r][0].w * rg[r][0].w,
                                    ra[r][1].x * rg[r][1].x, ra[r][1].y * rg[r][1].y, ra[r][1].z * rg[r][1].z, ra[r][1].w * rg[r][1].w};
                bh_split8(v, ah[r], al[r]);
            }
            if constexpr ((DBG & 1) != 0) {
#pragma unroll
                for (int r = 0; r < RB; r++) acc[r][0][0] += (float)ah[r][0] + (float)al[r][7];
                continue;
            }
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const f16x8 bh = wf[((st * NT + j) * 2 + 0) * 64 + lane];
                f16x8 bl;
                if (TERMS == 3) bl = wf[((st * NT + j) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int r = 0; r < RB; r++) {
                    acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh, acc[r][j], 0, 0, 0);
                    if (TERMS == 3) {
                        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl, acc[r][j], 0, 0, 0);
                        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh, acc[r][j], 0, 0, 0);
                    }
                }
            }
        }
        // Epilogue through LDS: the RB x 16 rows x N columns of this pass are ONE contiguous run of the output (row-major, N floats a
        // row), so the wave parks its accumulators there as that run and then moves it with whole 16-byte accesses -- residual in,
        // sums out.  Stored straight from the MFMA layout a 96-byte row (N = 24) went out as a 64- and a 32-byte piece per
        // instruction, four rows at a time: 2.4 TB/s on the 24 -> 24 block.
        if constexpr ((DBG & 8) != 0) {      // (keeps the accumulators alive)
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int j = 0; j < NT; j++) t += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
            if (t == 12345.678f) C[0] = t;
            continue;
        }
        float *ep = tsm + (size_t)steps * NT * 2 * 256 + (size_t)wave * (RB * 16 * NT * 16);
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int col = j * 16 + li;
                if (col >= N) continue;
                const float bv = bias[col];
#pragma unroll
                for (int q = 0; q < 4; q++) ep[(r * 16 + kq * 4 + q) * N + col] = __builtin_fmaf(acc[r][j][q], w_unscale, bv);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (wave-private LDS: in-order within the wave; this keeps the compiler from reordering)
        __builtin_amdgcn_wave_barrier();
        const int row0 = rt0 * 16, nrows = min(RB * 16, M - row0);
        const int nflt = nrows * N;                                   // (N % 4 == 0: whole float4s)
        float *cg = C + (size_t)row0 * N;
        const float *rgp = R ? R + (size_t)row0 * N : nullptr;
        for (int f = lane * 4; f < nflt; f += 256) {
            float4 v = *reinterpret_cast<const float4 *>(ep + f);
            if (rgp) { const float4 rr = *reinterpret_cast<const float4 *>(rgp + f); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
            *reinterpret_cast<float4 *>(cg + f) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------
// The gated project convolution of the MIDDLE and LATE squeeze-excite blocks (N = 64 .. 240, K = 288 .. 1 392; 256 or 64 pixels a
// segment): the streaming kernel above with W too large to stay in LDS -- it goes through LDS one 32-deep step at a time (LDS-DMA,
// a barrier per step), every wave keeps its RB x NT accumulator tiles across the steps, and a W fragment read from LDS feeds RB
// MFMA groups; no staging of A at all (a lane's 8 consecutive k of its row: two 16-byte loads, gate applied, split in registers).
// ONE workgroup per CU (256 registers a lane), the D rows PF k steps ahead and the W pieces PF - 1 ahead.  A first version (two
// workgroups per CU, rows one step ahead, W in chunks of <= 24 KB; N <= 144) reached 3.2-3.6 TB/s and left N = 232 to 128 x 128
// staged tiles that read D twice.  What the phases of such a kernel cost was measured with each switched off in turn
// (tools/microbench/gated_gemm.hip, profiles/r5_f_gated_gemm_phases.txt): with rows two steps and pieces one step ahead nothing
// overlapped -- a step lasted as long as the rows' round trip / 2 (4 us under load: 2 us a step) PLUS its MFMAs, a step of a narrow layer as long as the LDS-DMA's 1.1 us,
// and the epilogue's residual loads were waited for one row tile at a time.  The vector-memory counter retires IN ORDER and the W
// pieces share it with the row loads: waiting for the pieces issued one step ago forces every older row load home, however many
// register sets hold rows.  So the pieces go out earlier too (PF W buffers in LDS): at the top of step X the newest
// ND + (PF - 2)(PIECES + ND) operations may stay in flight -- rows X + 1 .. X + PF - 1 and pieces X + 1 .. X + PF - 2 -- an immediate,
// since every wave issues the same number of pieces (the last may repeat one: same bytes to the same place).  The gate rows of
// the pass's segments sit in LDS (zero beyond K): no second stream of global loads.  Order of a step: [wait] split this step's
// rows -> barrier -> issue {pieces X + PF - 1, rows X + PF} -> MFMAs.
// Epilogue: every wave parks a row tile in LDS (the W buffers, free by then) as the contiguous run of the output it is, its
// residual loads already in flight, and moves it with whole 16-byte accesses.
// ---------------------------------------------------------------------------------------
// (DBG: tools/microbench/gated_gemm.hip switches phases off at compile time -- 1 no MFMAs and no W reads, 2 no row loads, 4 no W pieces,
//  8 no epilogue, 16 MFMAs without the W reads, 32 W reads without the MFMAs; the product instantiates DBG = 0 only)
#define BH_GDBG(bit) ((DBG & (bit)) != 0)
template <int N_> __device__ __forceinline__ void bh_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N_) : "memory"); }
template <int TERMS, int NT, int RB, int PF, int DBG = 0>
__global__ __launch_bounds__(512, 1) void pw_gemm16_wide_kernel(const float *__restrict__ A, const float *__restrict__ gate, int rows_per_seg,
                                                                 const f16x8 *__restrict__ Wf, const float *__restrict__ bias,
                                                                 const float *__restrict__ R, float *__restrict__ C, int M, int K, int N,
                                                                 float w_unscale, int gs_max, int a_blocked) {
    // LDS: PF W step buffers [NT]{hi, lo}[64][8 halves] | the gate rows of the pass's segments [gs_max][32 steps], zero beyond K;
    // the epilogue's per-wave tiles [16][N] lie over both
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int STEP_FL = NT * 2 * 256;                         // floats per W step buffer
    constexpr int PIECES = (NT * 2 + 7) / 8;                      // 1-KiB pieces per wave and step
    constexpr int ND = RB * 2;                                    // row loads per lane and step
    constexpr int INFLIGHT = ND + (PF - 2) * (PIECES + ND);       // what may stay in flight at the top of a step (see above)
    static_assert(PF >= 2 && PF <= 4 && INFLIGHT <= 63, "prefetch depth");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int steps = (K + 31) / 32, KP = steps * 32;
    float *gs = wsm + PF * STEP_FL;
    float *ep = wsm + (size_t)wave * (16 * NT * 16);
    const int n_rt = (M + 15) >> 4;
    const float rcp_p = 1.0f / (float)rows_per_seg;
    const unsigned lds0 = (__builtin_amdgcn_groupstaticsize() + 15u) & ~15u;
    const int ws = __builtin_amdgcn_readfirstlane(wave);
    auto dma_w = [&](int st) {
        const int buf = st % PF;
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            const int p = min(ws + 8 * i, NT * 2 - 1);
            const f16x8 *src = Wf + ((size_t)st * NT * 2 + p) * 64;
            const unsigned dst = lds0 + 4u * (unsigned)(buf * STEP_FL + p * 256);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"((unsigned)lane * 16u), "s"(src), "s"(dst) : "memory", "m0");
        }
    };
    for (int rtb = blockIdx.x * 8 * RB; rtb < n_rt; rtb += gridDim.x * 8 * RB) {      // (uniform over the workgroup: barriers inside)
        const int rt0 = rtb + wave * RB;
        const int seg_lo = (rtb * 16) / rows_per_seg;
        const int seg_hi = (min((rtb + 8 * RB) * 16, M) - 1) / rows_per_seg;
        const float *ap[RB];
        const float *gl[RB];        // this lane's gate quads in LDS
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const int row = min((rt0 + r) * 16 + li, M - 1);                // rows past M: clamped, never stored
            int seg = (int)((float)row * rcp_p);                            // row / rows_per_seg through the reciprocal, fixed up
            seg += (row - seg * rows_per_seg >= rows_per_seg) ? 1 : 0;
            seg -= (row - seg * rows_per_seg < 0) ? 1 : 0;
            ap[r] = a_blocked ? A + (size_t)(row >> 4) * 16 * K + ((kq >> 1) * 16 + (row & 15)) * 16 + (kq & 1) * 8 : A + (size_t)row * K + 8 * kq;
            gl[r] = gs + (size_t)(seg - seg_lo) * KP + 8 * kq;
        }
        const int a_step = a_blocked ? 512 : 32;
        f32x4 acc[RB][NT];
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 ra[PF][RB][2];
        auto load_d = [&](int st, float4 (&a)[RB][2]) {
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const bool in = 32 * st + 8 * kq + 4 * h < K;       // (K % 4 == 0); beyond K the gate in LDS is zero, so any finite row bytes do
                    a[r][h] = *reinterpret_cast<const float4 *>(ap[r] + (in ? a_step * st + 4 * h : 0));
                }
        };
        auto step = [&](int st, float4 (&cur)[RB][2]) {
            float4 rg[RB][2];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                rg[r][0] = *reinterpret_cast<const float4 *>(gl[r] + 32 * st);
                rg[r][1] = *reinterpret_cast<const float4 *>(gl[r] + 32 * st + 4);
            }
            if (st + PF - 1 < steps && !BH_GDBG(2 | 4)) bh_wait_vm<INFLIGHT>();
            else bh_wait_vm<0>();         // (the last PF - 1 steps: fewer operations went out behind this step's)
            f16x8 ah[RB], al[RB];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                const float v[8] = {cur[r][0].x * rg[r][0].x, cur[r][0].y * rg[r][0].y, cur[r][0].z * rg[r][0].z, cur[r][0].w * rg[r][0].w,
                                    cur[r][1].x * rg[r][1].x, cur[r][1].y * rg[r][1].y, cur[r][1].z * rg[r][1].z, cur[r][1].w * rg[r][1].w};
                bh_split8(v, ah[r], al[r]);
            }
            __syncthreads();      // everyone's pieces of this step's W have landed; everyone is done with the buffer of step - 1
            if (st + PF - 1 < steps && !BH_GDBG(4)) dma_w(st + PF - 1);
            if (st + PF < steps && !BH_GDBG(2)) load_d(st + PF, cur);
            const f16x8 *wf = reinterpret_cast<const f16x8 *>(wsm + (size_t)(st % PF) * STEP_FL);
            if constexpr (BH_GDBG(1)) { acc[0][0][0] += (float)ah[0][0] + (float)al[RB - 1][7]; return; }
            if constexpr (BH_GDBG(16)) {      // MFMAs on whatever the first fragments hold: no LDS reads
                const f16x8 bh = wf[lane], bl = wf[64 + lane];
#pragma unroll
                for (int j = 0; j < NT; j++)
#pragma unroll
                    for (int r = 0; r < RB; r++) {
                        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh, acc[r][j], 0, 0, 0);
                        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl, acc[r][j], 0, 0, 0);
                        acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh, acc[r][j], 0, 0, 0);
                    }
                return;
            }
            if constexpr (BH_GDBG(32)) {      // the LDS reads without the MFMAs
#pragma unroll
                for (int j = 0; j < NT; j++) {
                    const f16x8 bh = wf[(j * 2 + 0) * 64 + lane], bl = wf[(j * 2 + 1) * 64 + lane];
                    acc[0][j][0] += (float)bh[0] + (float)bl[7] + (float)ah[0][0] + (float)al[RB - 1][7];
                }
                return;
            }
            // the next column tile's fragments are read while this one's MFMAs run; the three products of an accumulator keep
            // their order (hi hi, hi lo, lo hi), the row tiles alternate so that no MFMA waits for the one in front of it
            f16x8 bh = wf[lane], bl = bh;
            if (TERMS == 3) bl = wf[64 + lane];
#pragma unroll
            for (int j = 0; j < NT; j++) {
                f16x8 nh = bh, nl = bl;
                if (j + 1 < NT) {
                    nh = wf[((j + 1) * 2 + 0) * 64 + lane];
                    if (TERMS == 3) nl = wf[((j + 1) * 2 + 1) * 64 + lane];
                }
#pragma unroll
                for (int r = 0; r < RB; r++) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh, acc[r][j], 0, 0, 0);
                if (TERMS == 3) {
#pragma unroll
                    for (int r = 0; r < RB; r++) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl, acc[r][j], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < RB; r++) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh, acc[r][j], 0, 0, 0);
                }
                bh = nh; bl = nl;
            }
        };
        __syncthreads();          // the previous pass is done with the W buffers, the gate rows and the epilogue tiles
        long long t_0 = 0, t_1 = 0, t_2 = 0;
        if constexpr (BH_GDBG(64)) t_0 = __builtin_readcyclecounter();
        // issue order = the order the steps would have issued in: pieces s + PF - 1 in front of rows s + PF
#pragma unroll
        for (int s0 = 0; s0 < PF - 1; s0++) if (s0 < steps) dma_w(s0);
#pragma unroll
        for (int s0 = 0; s0 < PF; s0++) if (s0 < steps) load_d(s0, ra[s0]);
        {   // the gate rows of this pass's segments, zero beyond K
            const int ngs = (seg_hi - seg_lo + 1) * KP;
            for (int i = tid * 4; i < ngs; i += 512 * 4) {
                const int sg = i / KP, k = i - sg * KP;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K) g = *reinterpret_cast<const float4 *>(gate + (size_t)(seg_lo + sg) * K + k);
                *reinterpret_cast<float4 *>(gs + i) = g;
            }
        }
        __syncthreads();
        for (int st = 0; st < steps; st += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++)
                if (st + u < steps) step(st + u, ra[u]);
        }
        if constexpr (BH_GDBG(64)) t_1 = __builtin_readcyclecounter();
        __syncthreads();          // everyone is done with the last W buffers: the epilogue tiles lie over them
        if constexpr (BH_GDBG(64)) t_2 = __builtin_readcyclecounter();
        if constexpr (BH_GDBG(8)) {      // (keeps the accumulators alive)
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RB; r++)
#pragma unroll
                for (int j = 0; j < NT; j++) t += acc[r][j][0] + acc[r][j][1] + acc[r][j][2] + acc[r][j][3];
            if (t == 12345.678f) C[0] = t;
            continue;
        }
        const int Nn = N;
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const int row0 = (rt0 + r) * 16, nrows = min(16, M - row0);
            const int nflt = nrows * Nn;                                  // (N % 4 == 0: whole float4s)
            float4 rr[NT];                                                // this tile of the residual: in flight while the tile is parked
            const float *rgp = R ? R + (size_t)min(row0, M - 1) * Nn : nullptr;
#pragma unroll
            for (int f = 0; f < NT; f++) {
                rr[f] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rgp) rr[f] = *reinterpret_cast<const float4 *>(rgp + max(min(lane * 4 + f * 256, nflt - 4), 0));     // (clamped: unused beyond the tile)
            }
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const int col = j * 16 + li;
                if (col >= Nn) continue;
                const float bv = bias[col];
#pragma unroll
                for (int q = 0; q < 4; q++) ep[(kq * 4 + q) * Nn + col] = __builtin_fmaf(acc[r][j][q], w_unscale, bv);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (nflt > 0) {
                float *cg = C + (size_t)row0 * Nn;
#pragma unroll
                for (int f = 0; f < NT; f++) {
                    if (lane * 4 + f * 256 < nflt) {
                        float4 v = *reinterpret_cast<const float4 *>(ep + lane * 4 + f * 256);
                        if (rgp) { v.x += rr[f].x; v.y += rr[f].y; v.z += rr[f].z; v.w += rr[f].w; }
                        *reinterpret_cast<float4 *>(cg + lane * 4 + f * 256) = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");       // (the next tile's residual loads stay behind this tile's stores: one set of registers)
        }
        if constexpr (BH_GDBG(64)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long t_3 = __builtin_readcyclecounter();
            if (lane == 0) {
                long long *st = reinterpret_cast<long long *>(C + (size_t)M * N) + ((size_t)blockIdx.x * 8 + wave) * 4;
                st[0] = t_1 - t_0; st[1] = t_2 - t_1; st[2] = t_3 - t_2; st[3] = t_0;
            }
        }
    }
}

// the project convolution of a squeeze-excite block on the f16 MFMA: C = (D x gate) W + bias (+ R), no activation; K % 4 == 0,
// planes [ceil(K / 32)][ceil(N / 16)]{hi, lo}[64][8] zero-padded in K
// D blocked (kernels.hpp MbDesc::dblk) where the wide row-streaming kernel takes the layer (N = 96 .. 240): measured on the
// Perch-sized plan, 816 -> 136: 320 -> 280 us per 1 000 segments; the streaming kernel of the early blocks (whole tiles are
// contiguous in NHWC already) and the 128 x 128 staged tiles (N = 384) are 1.1-2x SLOWER on blocked rows
bool pw_gemm16_gated_wants_blocked(int K, int N, int rows_per_seg) {
    const int n_tiles = (N + 15) / 16;
    return K % 16 == 0 && rows_per_seg % 16 == 0 && N % 4 == 0 && n_tiles >= 6 && n_tiles <= 15;
}

void launch_pw_gemm16_gated(const float *A, const float *gate, int rows_per_seg, const void *Wf, const float *bias, const float *R,
                            float *C, int M, int K, int N, int terms, float w_unscale, int a_blocked, hipStream_t s) {
    const int n_tiles = (N + 15) / 16;
    // few columns, all of W in LDS (<= 64 KB), many rows: the streaming kernel above
    const size_t w_bytes = (size_t)((K + 31) / 32) * n_tiles * 2 * 1024;
    if (n_tiles <= 3 && w_bytes <= 64 * 1024 && M >= 4096 && N % 4 == 0 && !a_blocked) {      // (blocked rows: N >= 64 only, pw_gemm16_gated_wants_blocked)
        const bool shallow = K <= 32;
        const size_t thin_lds = w_bytes + (size_t)8 * (shallow ? (n_tiles <= 2 ? 3 : 2) : (n_tiles <= 2 ? 4 : 3)) * 16 * n_tiles * 16 * sizeof(float);   // + the waves' epilogue tiles
        const int n_rt = (M + 15) / 16;
        const int wgs = std::min((n_rt + 31) / 32, 2 * device_cu_count());   // 8-wave workgroups walking the row tiles, 2-4 per wave and pass (one or two resident per CU)
#define BH_THIN(T, NTV, SH)                                                                                                        \
        do {                                                                                                                       \
            static DeviceOnce attr;                                                                                                \
            attr.run([] { (void)hipFuncSetAttribute((const void *)pw_gemm16_thin_kernel<T, NTV, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); }); \
            hipLaunchKernelGGL((pw_gemm16_thin_kernel<T, NTV, SH>), dim3(wgs), dim3(512), thin_lds, s, A, gate, rows_per_seg, (const f16x8 *)Wf, bias, R, C, M, K, N, w_unscale); \
        } while (0)
#define BH_THIN_S(T, NTV) do { if (shallow) BH_THIN(T, NTV, true); else BH_THIN(T, NTV, false); } while (0)
        if (terms == 3) { if (n_tiles == 1) BH_THIN_S(3, 1); else if (n_tiles == 2) BH_THIN_S(3, 2); else BH_THIN_S(3, 3); }
        else { if (n_tiles == 1) BH_THIN_S(1, 1); else if (n_tiles == 2) BH_THIN_S(1, 2); else BH_THIN_S(1, 3); }
#undef BH_THIN_S
#undef BH_THIN
        return;
    }
    // N = 64 .. 240 with many rows: the row-streaming kernel with one workgroup per CU, rows and W pieces several steps ahead
    // (N > 144: two row tiles a wave, 256 rows a pass, and a pass lasts ~125 us however few there are -- below ~40 000 rows, fewer
    //  passes than three quarters of the CUs, the staged tiles' 2 x M / 128 workgroups win: 1 392 -> 232 at 16 384 rows 88 against
    //  128 us, at 32 768 110 against 144, at 64 000 294 against 187; N <= 144 wins from 16 384 rows down to the 4 096 measured.
    //  The same bits either way.)
    if (n_tiles >= 4 && n_tiles <= 15 && M >= (n_tiles >= 10 ? 40960 : 4096) && N % 4 == 0) {
        const int n_rt = (M + 15) / 16;
#define BH_WIDE(T, NTV, RBV, PFV)                                                                                                  \
        do {                                                                                                                       \
            const int gs_max = (8 * RBV * 16 + rows_per_seg - 2) / rows_per_seg + 1;                                              \
            const size_t lds = std::max((size_t)PFV * NTV * 2048 + (size_t)gs_max * ((K + 31) / 32 * 32) * sizeof(float),         \
                                        (size_t)8 * 16 * NTV * 16 * sizeof(float));                                              \
            if (lds > 160 * 1024) break;     /* (very many segments a pass: the staged kernel below) */                           \
            const int wgs = std::min((n_rt + 8 * RBV - 1) / (8 * RBV), device_cu_count());                                       \
            static DeviceOnce attr;                                                                                                \
            attr.run([] { (void)hipFuncSetAttribute((const void *)pw_gemm16_wide_kernel<T, NTV, RBV, PFV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
            hipLaunchKernelGGL((pw_gemm16_wide_kernel<T, NTV, RBV, PFV>), dim3(wgs), dim3(512), lds, s, A, gate, rows_per_seg, (const f16x8 *)Wf, bias, R, C, M, K, N, w_unscale, gs_max, a_blocked); \
            return;                                                                                                                \
        } while (0)
#define BH_WIDE_T(T)                                                                             \
        switch (n_tiles) {                                                                      \
        case 4: BH_WIDE(T, 4, 3, 3); break; case 5: BH_WIDE(T, 5, 3, 3); break; case 6: BH_WIDE(T, 6, 3, 3); break; case 7: BH_WIDE(T, 7, 3, 3); break; case 8: BH_WIDE(T, 8, 2, 4); break; case 9: BH_WIDE(T, 9, 2, 4); break;  \
        case 10: BH_WIDE(T, 10, 2, 3); break; case 11: BH_WIDE(T, 11, 2, 3); break; case 12: BH_WIDE(T, 12, 2, 3); break; case 13: BH_WIDE(T, 13, 2, 3); break; \
        case 14: BH_WIDE(T, 14, 2, 3); break; default: BH_WIDE(T, 15, 2, 3); break; }
        if (terms == 3) { BH_WIDE_T(3) } else { BH_WIDE_T(1) }
#undef BH_WIDE_T
#undef BH_WIDE
    }
    // column tiles per workgroup: the width that pads N least (96 -> 6, 136 -> 10, 232 -> 8 + 8, 384 -> 3 x 8); ties go to the wider
    int ntb = 8;
    {
        long best = -1;
        for (int cand : {10, 8, 6}) {
            const long padded = (long)((n_tiles + cand - 1) / cand) * cand;
            if (best < 0 || padded < best) { best = padded; ntb = cand; }
        }
    }
    const int n_xb = (n_tiles + ntb - 1) / ntb, n_yb = (M + 127) / 128;
    dim3 grid((unsigned)(n_xb * n_yb)), block(256);
#define BH_GS(T, NTBV, BLKV)                                                                                                      \
    do {                                                                                                                          \
        constexpr size_t lds = 2 * ((8 + NTBV) * 2 * 256) * sizeof(float);                                                       \
        static DeviceOnce attr;                                                                                                   \
        attr.run([] { (void)hipFuncSetAttribute((const void *)pw_gemm16s_kernel<T, ACT_NONE, true, NTBV, BLKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
        hipLaunchKernelGGL((pw_gemm16s_kernel<T, ACT_NONE, true, NTBV, BLKV>), grid, block, lds, s, A, (const f16x8 *)Wf, bias, R, C, M, K, N, n_tiles, w_unscale, gate, rows_per_seg); \
    } while (0)
#define BH_GS_B(T, NTBV) do { if (a_blocked) BH_GS(T, NTBV, true); else BH_GS(T, NTBV, false); } while (0)
    if (terms == 3) { if (ntb == 6) BH_GS_B(3, 6); else if (ntb == 10) BH_GS_B(3, 10); else BH_GS_B(3, 8); }
    else { if (ntb == 6) BH_GS_B(1, 6); else if (ntb == 10) BH_GS_B(1, 10); else BH_GS_B(1, 8); }
#undef BH_GS_B
#undef BH_GS
}

void launch_pw_gemm16(const float *A, const void *Wf, const float *bias, const float *R, float *C, int M, int K, int N,
                      int act, int terms, float w_unscale, hipStream_t s) {
    const int n_tiles = (N + 15) / 16;
    const int n_xb = (n_tiles + 7) / 8, n_yb = (M + 127) / 128;
    dim3 grid((unsigned)(8 * ((n_xb + 7) / 8) * n_yb)), block(256);   // (one-dimensional: the kernel deals the blocks XCD by XCD)
    // (the LDS-staged kernel wherever a workgroup's 128 rows exist; a handful of rows stream as before: nothing to share)
    static const bool stream_only = [] { const char *e = BH_XENV("BIRDA_HIP_GEMM_STREAM"); return e && e[0] == '1'; }();   // A/B aid
    const bool staged = M >= 64 && !stream_only;
    // (round 6: a launch of a few hundred rows has one or two row blocks, and with eight column tiles a workgroup the dense layers
    //  leave half the CUs idle and every workgroup alone on its CU, its step a bare HBM round trip + its MFMAs: four or two column
    //  tiles a workgroup while the grid stays under ~1.5 workgroups a CU.  The same products in the same order: the same bits.)
    const int ntb = !staged ? 8 : n_xb * n_yb >= 384 ? 8 : ((n_tiles + 3) / 4) * n_yb >= 384 ? 4 : 2;
    const int n_xb_s = (n_tiles + ntb - 1) / ntb;
    const dim3 grid_s((unsigned)(8 * ((n_xb_s + 7) / 8) * n_yb));
#define BH_G16S(T, ACTV, NTBV)                                                                                                    \
    do {                                                                                                                          \
        constexpr size_t lds = 2 * ((8 + NTBV) * 2 * 256) * sizeof(float);                                                        \
        static DeviceOnce attr;                                                                                                   \
        attr.run([] { (void)hipFuncSetAttribute((const void *)pw_gemm16s_kernel<T, ACTV, false, NTBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
        hipLaunchKernelGGL((pw_gemm16s_kernel<T, ACTV, false, NTBV>), grid_s, block, lds, s, A, (const f16x8 *)Wf, bias, R, C, M, K, N, n_tiles, w_unscale); \
    } while (0)
#define BH_G16(T, ACTV)                                                                                                           \
    do {                                                                                                                          \
        if (staged) {                                                                                                             \
            if (ntb == 8) BH_G16S(T, ACTV, 8); else if (ntb == 4) BH_G16S(T, ACTV, 4); else BH_G16S(T, ACTV, 2);                  \
        } else if (M <= 32) {    /* one or two row tiles: the skinny kernel per row tile, eight steps in flight (the same bits) */ \
            for (int m0 = 0; m0 < M; m0 += 16)                                                                                    \
                hipLaunchKernelGGL((pw_gemm16_skinny_kernel<T, ACTV>), dim3((unsigned)((n_tiles + 7) / 8)), block, 0, s, A + (size_t)m0 * K, (const f16x8 *)Wf, bias, \
                                   R ? R + (size_t)m0 * N : nullptr, C + (size_t)m0 * N, std::min(16, M - m0), K, N, n_tiles, w_unscale); \
        } else {                                                                                                                  \
            hipLaunchKernelGGL((pw_gemm16_kernel<T, ACTV>), grid, block, 0, s, A, (const f16x8 *)Wf, bias, R, C, M, K, N, n_tiles, w_unscale); \
        }                                                                                                                         \
    } while (0)
#define BH_G16A(T)                                                   \
    switch (act) {                                                   \
    case ACT_GELU_ERF: BH_G16(T, ACT_GELU_ERF); break;               \
    case ACT_SWISH: BH_G16(T, ACT_SWISH); break;                     \
    case ACT_RELU6: BH_G16(T, ACT_RELU6); break;                     \
    default: BH_G16(T, ACT_NONE); break;                             \
    }
    if (terms == 3) { BH_G16A(3) } else { BH_G16A(1) }
#undef BH_G16A
#undef BH_G16
#undef BH_G16S
}

// ---------------------------------------------------------------------------------------
// Head 1x1 conv + GELU + global average pool in one launch (split-f16 / f16 MFMA):
//     out[seg][n] = (1 / P) sum_p GELU(X[seg][p][:] . W[:][n] + b[n])
// The unfused pair writes the [n_seg * P][N] activation (196 MB per 1000 BirdNET segments) and reads it
// straight back to average it; here it never exists.  (reference: the last Conv + Mul/Erf + GlobalAveragePool
// nodes of the ONNX graph behind birdnet_onnx::Classifier::predict_batch, src/inference/classifier.rs:478-488.)
//
// Block = 4 waves, 4 * SW segments x 128 output channels; wave w owns SW whole segments (PT row tiles of
// 16 pixels each) x the block's 8 column tiles: its accumulators hold every pixel of its segments, so
// the pool is an in-register sum plus two cross-lane adds.  A (f32 rows, 8 consecutive k per lane) goes
// global -> registers one k step ahead and is split into f16 hi / lo there; the block's 16-KB slice of
// the pre-split weights (kernels.hpp: [k step][column tile]{hi, lo}[64 lanes][8 halves]) is shared by
// the four waves through LDS, filled by LDS-DMA one step ahead (double-buffered).
// Blocks that share their rows of X are placed on the same XCD (blockIdx % 8) back to back, so X is
// fetched from HBM once and re-read from that XCD's L2 by the other column blocks.
// ---------------------------------------------------------------------------------------
// CT: column tiles a workgroup owns -- 8 (128 output channels), or 2 for launches of a few segments (round 6): the grid is then four
// times as wide and a workgroup's per-step work a quarter, which is what a launch that cannot fill the chip lasts (35 -> ~15 us for
// one segment).  The same products in the same order either way.
template <int PT, int SW, int TERMS, int ACT, int CT = 8>
__global__ __launch_bounds__(256, 1) void head_gap16_kernel(const float *__restrict__ A, const f16x8 *__restrict__ Wf,
                                                             const float *__restrict__ bias, float *__restrict__ out,
                                                             int n_seg, int P, int K, int N, int n_tiles, int n_cb, float w_unscale) {
    constexpr int RT = PT * SW;
    __shared__ __attribute__((aligned(16))) f16x8 Bs[2][CT * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int cb = bi % n_cb, mb = (bi / n_cb) * 8 + xcd;
    const int seg0 = (mb * 4 + wave) * SW;
    if (mb * 4 * SW >= n_seg) return;   // whole block
    const int steps = K / 32;

    // B slice of step st -> Bs[buf]: 16 pieces of 1 KiB, four per wave (M0 = LDS address of the piece)
    auto dma = [&](int st, int buf) {
        const f16x8 *src = Wf + ((size_t)st * n_tiles + cb * CT) * 2 * 64;
#pragma unroll
        for (int q = 0; q < (CT * 2 + 3) / 4; q++) {
            const int piece = q * 4 + wave;
            if (piece >= CT * 2) continue;     // (CT = 2: one piece a wave)
            const unsigned la = (unsigned)(size_t)(&Bs[buf][piece * 64]);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                         :: "v"(src + piece * 64 + lane), "s"(__builtin_amdgcn_readfirstlane(la)) : "memory", "m0");
        }
    };
    // element offset of the lane's row in A; padding rows (pixel >= P, segment >= n_seg) read a valid row
    // instead and are left out of the pool below  (a conditional load here crashes hipcc 7.2's machine
    // copy propagation pass)
    int aoff[RT];
#pragma unroll
    for (int i = 0; i < RT; i++) {
        const int seg = min(seg0 + i / PT, n_seg - 1), px = min((i % PT) * 16 + li, P - 1);
        aoff[i] = (seg * P + px) * K + 8 * kq;
    }
    float4 ra[RT][2];
    auto load_a = [&](int st) {
#pragma unroll
        for (int i = 0; i < RT; i++) {
            ra[i][0] = *reinterpret_cast<const float4 *>(A + (size_t)aoff[i] + 32 * st);
            ra[i][1] = *reinterpret_cast<const float4 *>(A + (size_t)aoff[i] + 32 * st + 4);
        }
    };
    f32x4 acc[RT][CT];
#pragma unroll
    for (int j = 0; j < CT; j++)
#pragma unroll
        for (int i = 0; i < RT; i++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 ah[RT], al[RT];
    auto split_a = [&]() {
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const float v[8] = {ra[i][0].x, ra[i][0].y, ra[i][0].z, ra[i][0].w, ra[i][1].x, ra[i][1].y, ra[i][1].z, ra[i][1].w};
            bh_split8(v, ah[i], al[i]);
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CT; j++) {
            const f16x8 bh = Bs[buf][(j * 2 + 0) * 64 + lane];
            f16x8 bl;
            if (TERMS == 3) bl = Bs[buf][(j * 2 + 1) * 64 + lane];
#pragma unroll
            for (int i = 0; i < RT; i++) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
                if (TERMS == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    dma(0, 0);
    load_a(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int st = 0; st + 1 < steps; st++) {   // the last step is peeled: no conditional loads in the loop
        dma(st + 1, (st + 1) & 1);
        split_a();
        __builtin_amdgcn_sched_barrier(0);    // the split stays out of the MFMA sequence (see mel_kernel)
        load_a(st + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(st & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    split_a();
    __builtin_amdgcn_sched_barrier(0);
    compute((steps - 1) & 1);

    // GELU, then the mean over the segment's pixels: rows 4 kq + r of PT tiles in registers, kq across lanes
    const float inv_p = 1.0f / (float)P;
#pragma unroll
    for (int sg = 0; sg < SW; sg++) {
        const int seg = seg0 + sg;
#pragma unroll
        for (int j = 0; j < CT; j++) {
            float sum = 0.0f;
            const float b = bias[(cb * CT + j) * 16 + li];
#pragma unroll
            for (int t = 0; t < PT; t++) {
                const f32x4 a4 = acc[sg * PT + t][j];   // the planes hold W / w_unscale
                bh_f32x2 v01 = {__builtin_fmaf(a4[0], w_unscale, b), __builtin_fmaf(a4[1], w_unscale, b)},
                         v23 = {__builtin_fmaf(a4[2], w_unscale, b), __builtin_fmaf(a4[3], w_unscale, b)};
                bh_act4<ACT>(v01, v23);
                const int px = t * 16 + 4 * kq;
                if (PT * 16 == P || px + 3 < P) sum += (v01[0] + v01[1]) + (v23[0] + v23[1]);
                else sum += (px < P ? v01[0] : 0.f) + (px + 1 < P ? v01[1] : 0.f) + (px + 2 < P ? v23[0] : 0.f);
            }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (kq == 0 && seg < n_seg) out[(size_t)seg * N + (cb * CT + j) * 16 + li] = sum * inv_p;
        }
    }
}

// P pixels per segment; the instantiations cover P <= 48 (two segments per wave) and P <= 80 (one)
bool head_gap16_supports(int P, int K, int N, int act) {
    return act_is_templated(act) && K % 32 == 0 && N % 128 == 0 && P >= 1 && P <= 80;
}

void launch_head_gap16(const float *A, const void *Wf, const float *bias, float *out, int n_seg, int P, int K, int N,
                       int act, int terms, float w_unscale, hipStream_t s) {
    const bool few = n_seg <= 32;      // (a launch that leaves most of the chip idle: two column tiles a workgroup, four times the workgroups)
    const int n_tiles = N / 16, n_cb = few ? N / 32 : N / 128;
    const int pt = (P + 15) / 16, sw = pt <= 3 ? 2 : 1;
    const int n_mb = (n_seg + 4 * sw - 1) / (4 * sw);
    dim3 grid((unsigned)(((n_mb + 7) / 8) * n_cb * 8)), block(256);
#define BH_HG(PTV, SWV, T, ACTV) do { if (few) hipLaunchKernelGGL((head_gap16_kernel<PTV, SWV, T, ACTV, 2>), grid, block, 0, s, A, (const f16x8 *)Wf, bias, out, \
                                                    n_seg, P, K, N, n_tiles, n_cb, w_unscale); \
    else hipLaunchKernelGGL((head_gap16_kernel<PTV, SWV, T, ACTV, 8>), grid, block, 0, s, A, (const f16x8 *)Wf, bias, out, \
                                                    n_seg, P, K, N, n_tiles, n_cb, w_unscale); } while (0)
#define BH_HGA(PTV, SWV, T)                                          \
    switch (act) {                                                   \
    case ACT_SWISH: BH_HG(PTV, SWV, T, ACT_SWISH); break;            \
    case ACT_RELU6: BH_HG(PTV, SWV, T, ACT_RELU6); break;            \
    default: BH_HG(PTV, SWV, T, ACT_GELU_ERF); break;                \
    }
    if (pt <= 3) { if (terms == 3) { BH_HGA(3, 2, 3) } else { BH_HGA(3, 2, 1) } }
    else { if (terms == 3) { BH_HGA(5, 1, 3) } else { BH_HGA(5, 1, 1) } }
#undef BH_HGA
#undef BH_HG
}

// ---------------------------------------------------------------------------------------
// Depthwise conv, NHWC.  One lane = one output pixel x 4 channels; lanes run over the
// channel groups first, so a wave reads/writes contiguous 16-B pieces of NHWC rows.
// ---------------------------------------------------------------------------------------
template <int KS, int ST>
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                      const float *__restrict__ b, float *__restrict__ out,
                                                      ConvParams p, int n_seg) {
    const int c4n = p.cout >> 2;
    const long total = (long)n_seg * p.out_h * p.out_w * c4n;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int c4 = (int)(g % c4n);
        long px = g / c4n;
        const int ox = (int)(px % p.out_w);
        px /= p.out_w;
        const int oy = (int)(px % p.out_h);
        const int seg = (int)(px / p.out_h);
        const int c = c4 << 2;
        float4 acc = *reinterpret_cast<const float4 *>(b + c);
        const float *ib = in + (size_t)seg * p.in_h * p.in_w * p.cout + c;
        const int iy0 = oy * ST - p.pad_t, ix0 = ox * ST - p.pad_l;
#pragma unroll
        for (int dy = 0; dy < KS; dy++) {
            const int iy = iy0 + dy;
            if (iy < 0 || iy >= p.in_h) continue;
#pragma unroll
            for (int dx = 0; dx < KS; dx++) {
                const int ix = ix0 + dx;
                if (ix < 0 || ix >= p.in_w) continue;
                const float4 a = *reinterpret_cast<const float4 *>(ib + ((size_t)iy * p.in_w + ix) * p.cout);
                const float4 ww = *reinterpret_cast<const float4 *>(w + (size_t)(dy * KS + dx) * p.cout + c);
                acc.x += a.x * ww.x; acc.y += a.y * ww.y; acc.z += a.z * ww.z; acc.w += a.w * ww.w;
            }
        }
        acc.x = act_apply(acc.x, p.act); acc.y = act_apply(acc.y, p.act);
        acc.z = act_apply(acc.z, p.act); acc.w = act_apply(acc.w, p.act);
        *reinterpret_cast<float4 *>(out + (((size_t)seg * p.out_h + oy) * p.out_w + ox) * p.cout + c) = acc;
    }
}

void launch_dwconv(const float *in, const float *w, const float *b, float *out, const ConvParams &p, int n_seg,
                   hipStream_t s) {
    const long total = (long)n_seg * p.out_h * p.out_w * (p.cout / 4);
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 64) blocks = 256L * 64;
    dim3 grid((unsigned)blocks), block(256);
#define BH_DW_CASE(KSV, STV)                                                                       \
    if (p.kh == KSV && p.sh == STV) {                                                               \
        hipLaunchKernelGGL((dwconv_kernel<KSV, STV>), grid, block, 0, s, in, w, b, out, p, n_seg);  \
        return;                                                                                     \
    }
    BH_DW_CASE(3, 1) BH_DW_CASE(3, 2) BH_DW_CASE(5, 1) BH_DW_CASE(5, 2)
#undef BH_DW_CASE
}

// ---------------------------------------------------------------------------------------
// Direct conv for the stem (Cin = n_branches = 2): weights [kh][kw][cin][cout] in LDS,
// one lane = one output pixel x NC = 8 output channels (4 where the width is not a multiple of 8: stems of 20, 28, 36 ... channels).
// ---------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void conv_direct_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                           const float *__restrict__ b, float *__restrict__ out,
                                                           ConvParams p, int n_seg) {
    extern __shared__ __attribute__((aligned(16))) float ws[];
    const int wn = p.kh * p.kw * p.cin * p.cout;
    for (int i = threadIdx.x; i < wn; i += 256) ws[i] = w[i];
    __syncthreads();
    const int cgn = p.cout / NC;
    const long total = (long)n_seg * p.out_h * p.out_w * cgn;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int cg = (int)(g % cgn);
        long px = g / cgn;
        const int ox = (int)(px % p.out_w);
        px /= p.out_w;
        const int oy = (int)(px % p.out_h);
        const int seg = (int)(px / p.out_h);
        const int co = cg * NC;
        float acc[NC];
#pragma unroll
        for (int n = 0; n < NC; n++) acc[n] = b[co + n];
        const float *ib = in + (size_t)seg * p.in_h * p.in_w * p.cin;
        for (int dy = 0; dy < p.kh; dy++) {
            const int iy = oy * p.sh - p.pad_t + dy;
            if (iy < 0 || iy >= p.in_h) continue;
            for (int dx = 0; dx < p.kw; dx++) {
                const int ix = ox * p.sw - p.pad_l + dx;
                if (ix < 0 || ix >= p.in_w) continue;
                for (int c = 0; c < p.cin; c++) {
                    const float a = p.in_layout == 1 ? ib[((size_t)c * p.in_h + iy) * p.in_w + ix]
                                                     : ib[((size_t)iy * p.in_w + ix) * p.cin + c];
                    const float *wr = ws + ((dy * p.kw + dx) * p.cin + c) * p.cout + co;
#pragma unroll
                    for (int n = 0; n < NC; n++) acc[n] += a * wr[n];
                }
            }
        }
        float *o = out + (((size_t)seg * p.out_h + oy) * p.out_w + ox) * p.cout + co;
#pragma unroll
        for (int q = 0; q < NC; q += 4)
            *reinterpret_cast<float4 *>(o + q) = make_float4(act_apply(acc[q], p.act), act_apply(acc[q + 1], p.act), act_apply(acc[q + 2], p.act),
                                                             act_apply(acc[q + 3], p.act));
    }
}

void launch_conv_direct(const float *in, const float *w, const float *b, float *out, const ConvParams &p,
                        int n_seg, hipStream_t s) {
    const int nc = p.cout % 8 ? 4 : 8;
    const long total = (long)n_seg * p.out_h * p.out_w * (p.cout / nc);
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    const size_t smem = (size_t)p.kh * p.kw * p.cin * p.cout * sizeof(float);
    if (nc == 8) hipLaunchKernelGGL(conv_direct_kernel<8>, dim3((unsigned)blocks), dim3(256), smem, s, in, w, b, out, p, n_seg);
    else hipLaunchKernelGGL(conv_direct_kernel<4>, dim3((unsigned)blocks), dim3(256), smem, s, in, w, b, out, p, n_seg);
}

// ---------------------------------------------------------------------------------------
// Global average pool [n][P][C] -> [n][C]; sum in pixel order then * (1/P) like the oracle
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_kernel(const float *__restrict__ in, float *__restrict__ out, int n_seg,
                                                   int P, int C) {
    const int c4n = C >> 2;
    const long total = (long)n_seg * c4n;
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int c = (int)(g % c4n) << 2;
    const int seg = (int)(g / c4n);
    const float *ib = in + (size_t)seg * P * C + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int px = 0; px < P; px++) {
        const float4 a = *reinterpret_cast<const float4 *>(ib + (size_t)px * C);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    const float inv = 1.0f / (float)P;
    *reinterpret_cast<float4 *>(out + (size_t)seg * C + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

void launch_gap(const float *in, float *out, int n_seg, int P, int C, hipStream_t s) {
    const long total = (long)n_seg * (C / 4);
    hipLaunchKernelGGL(gap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, n_seg, P, C);
}

// ---------------------------------------------------------------------------------------
// The gate of a squeeze-excite block in one launch (round 5): GlobalAveragePool -> 1x1 conv (C -> Cr, act1) -> 1x1 conv (Cr -> C,
// act2 = sigmoid) of the ONNX graph, from the per-tile channel sums pass A of the fused block left (mbconv_kernel SE = 1).
// One workgroup per segment; every sum runs in a fixed order (no atomics: identical segments give identical gates).
//   part [n][tiles][C]   W1 [C][ld1]  b1 [Cr]   W2 [Cr][ld2]  b2 [C]   gate [n][C]
// LDS: pooled [C] | partial dot products [256] | hidden [Cr]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_gate_kernel(const float *__restrict__ part, int tiles, float inv_p, const float *__restrict__ W1,
                                                       const float *__restrict__ b1, int ld1, int act1, const float *__restrict__ W2,
                                                       const float *__restrict__ b2, int ld2, int act2, float *__restrict__ gate, int C, int Cr,
                                                       int crp) {
    extern __shared__ __attribute__((aligned(16))) float gs[];
    float *pooled = gs, *hp = gs + C, *hid = hp + 256;
    const int tid = threadIdx.x, seg = blockIdx.x;
    const float *ps = part + (size_t)seg * tiles * C;
    // (round 6: the early blocks arrive as 64-128 tile rows of 24-144 channels -- 256 / C lanes of threads share the tiles of a channel,
    //  lane l the tiles l, l + lanes, ..., and the lanes' sums add in lane order: a tenth of the dependent loads, the same order
    //  whatever the launch)
    const int lanes = C <= 128 && tiles >= 8 ? 256 / C : 1;
    if (lanes > 1) {
        const int c = tid % C, l = tid / C;
        float sum = 0.0f;
        if (l < lanes) {
#pragma unroll 8
            for (int t = l; t < tiles; t += lanes) sum += ps[(size_t)t * C + c];
            hp[l * C + c] = sum;
        }
        __syncthreads();
        if (tid < C) {
            float tot = hp[tid];
            for (int q = 1; q < lanes; q++) tot += hp[q * C + tid];
            pooled[tid] = tot * inv_p;
        }
    } else
        for (int c = tid; c < C; c += 256) {
            float sum = 0.0f;
#pragma unroll 8
            for (int t = 0; t < tiles; t++) sum += ps[(size_t)t * C + c];
            pooled[c] = sum * inv_p;
        }
    __syncthreads();
    // hidden layer: thread (r, part) sums its slice of the channels; crp = the power of two >= Cr, 256 / crp slices
    const int nparts = 256 / crp, r = tid & (crp - 1), pt = tid / crp, slice = (C + nparts - 1) / nparts;
    {
        float sum = 0.0f;
        if (r < Cr) {
            const int c0 = pt * slice, c1 = min(C, c0 + slice);
#pragma unroll 16
            for (int c = c0; c < c1; c++) sum = __builtin_fmaf(pooled[c], W1[(size_t)c * ld1 + r], sum);
        }
        hp[tid] = sum;
    }
    __syncthreads();
    if (tid < Cr) {
        float sum = b1[tid];
        for (int q = 0; q < nparts; q++) sum += hp[q * crp + tid];
        hid[tid] = act_apply(sum, act1);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float sum = b2[c];
#pragma unroll 16
        for (int q = 0; q < Cr; q++) sum = __builtin_fmaf(hid[q], W2[(size_t)q * ld2 + c], sum);
        gate[(size_t)seg * C + c] = act_apply(sum, act2);
    }
}

// The gate of the wide blocks (round 6, VERDICT r5 next #3b: beyond 576 channels it was three launches -- pool, GEMM, GEMM -- whose
// first GEMM has a handful of tiles and a K loop of 816-3 840 channels: 64-134 us per launch set whatever the block).  Two launches of
// many small workgroups instead, sixteen segments each:
//   se_hidden_kernel   grid (groups of 16 segments, KS channel slices): pools its slice of the per-tile channel sums into LDS (tiles in
//                      ascending order, x 1/P), multiplies by its rows of W1, leaves PARTIAL hidden sums [KS][n][Cr];
//   se_gate16_kernel   grid (groups, 256-channel column blocks): hidden = act1(b1 + the KS partials in ascending order) into LDS,
//                      gate = act2(b2 + hidden x W2), a channel a thread.
// What these launches cost is DEPENDENT MEMORY ROUNDS, not bytes or flops: a block's gate weights are touched once a forward, every
// load of them is an HBM miss (~0.55 us measured per round trip), and a workgroup is a chain of such round trips whatever the launch
// size (tools/microbench/se_gate.hip, tools/se_trace.py).  So: the slice width is chosen for <= 32 rows of W1 a thread, every batch
// of loads is issued whole (up to 32 in flight a thread, indices clamped and the weight of a row beyond the end zeroed: no branch
// between the loads and their uses -- with one the compiler sinks each load to its use and the chain is back), the first batch of
// weights is on its way before the pooled sums / the partial sums are.
// KS and every summation order depend on the block's widths alone: a segment's gate does not depend on the launch it ran in.
constexpr int SE_SG = 16;
constexpr int SE_RB = 32;          // weight rows in flight a thread
struct SeHiddenShape { int slice, slices; };
static inline SeHiddenShape se_hidden_shape(int C, int Cr) {
    int slice = 256;
    while (slice > 64 && slice * Cr > SE_RB * 256) slice >>= 1;                             // <= 32 rows of W1 a thread
    while (slice < 256 && ((C + slice - 1) / slice) * Cr > C) slice <<= 1;                    // the partial sums must fit their scratch
    return {slice, (C + slice - 1) / slice};
}
// keeps a batch of loaded values where it was loaded (see above)
#define SE_PIN(arr, n)                                                                                                                     \
    _Pragma("unroll") for (int j_ = 0; j_ < (n); j_++) asm volatile("" : "+v"((arr)[j_]))

__global__ __launch_bounds__(256) void se_hidden_kernel(const float *__restrict__ part, int tiles, float inv_p, const float *__restrict__ W1, int ld1,
                                                         float *__restrict__ hpart, int n_seg, int C, int Cr, int slice) {
    extern __shared__ __attribute__((aligned(16))) float gs[];
    float *pooledT = gs, *hp = gs + (size_t)slice * SE_SG;          // [slice][16] | [256][16]
    const int tid = threadIdx.x, seg0 = blockIdx.x * SE_SG, ns = min(SE_SG, n_seg - seg0), ks = blockIdx.y;
    const int c0 = ks * slice, len = min(slice, C - c0);
    // thread (r, pt): hidden unit r over the rows [s0, s1) of the slice
    const int nparts = 256 / Cr, r = tid % Cr, pt = tid / Cr, sub = (len + nparts - 1) / nparts;
    const int s0 = min(pt * sub, len), s1 = pt < nparts ? min(len, s0 + sub) : s0, last = max(s1 - 1, 0);
    const float *w1 = W1 + (size_t)c0 * ld1 + r;
    float w[SE_RB];
#pragma unroll
    for (int j = 0; j < SE_RB; j++) w[j] = w1[(size_t)min(s0 + j, last) * ld1];
    // a thread pools ONE channel of the sixteen segments: sixteen independent loads a tile (a segment beyond the batch reads the
    // last one's sums again and is never written)
    if (tid < len) {
        float acc[SE_SG];
        const float *ps[SE_SG];
#pragma unroll
        for (int sg = 0; sg < SE_SG; sg++) acc[sg] = 0.0f, ps[sg] = part + (size_t)(seg0 + min(sg, ns - 1)) * tiles * C + c0 + tid;
#pragma unroll 2
        for (int t = 0; t < tiles; t++)
#pragma unroll
            for (int sg = 0; sg < SE_SG; sg++) acc[sg] += ps[sg][(size_t)t * C];
        float ip = inv_p;
        asm volatile("" : "+v"(ip));          // (a VGPR: the scalar pair would make these v_pk_mul_f32 with op_sel -- tests/test_abi_and_host.py)
#pragma unroll
        for (int v = 0; v < SE_SG / 4; v++)
            reinterpret_cast<float4 *>(pooledT + tid * SE_SG)[v] = make_float4(acc[4 * v] * ip, acc[4 * v + 1] * ip, acc[4 * v + 2] * ip, acc[4 * v + 3] * ip);
    }
    __syncthreads();
    float sum[SE_SG];
#pragma unroll
    for (int sg = 0; sg < SE_SG; sg++) sum[sg] = 0.0f;
    for (int b0 = s0; b0 < s1; b0 += SE_RB) {
        if (b0 != s0) {
#pragma unroll
            for (int j = 0; j < SE_RB; j++) w[j] = w1[(size_t)min(b0 + j, last) * ld1];
        }
        SE_PIN(w, SE_RB);
#pragma unroll
        for (int j = 0; j < SE_RB; j++) {
            const float wj = b0 + j < s1 ? w[j] : 0.0f;
            const float4 *pp = reinterpret_cast<const float4 *>(pooledT + min(b0 + j, last) * SE_SG);
#pragma unroll
            for (int v = 0; v < SE_SG / 4; v++) {
                const float4 p = pp[v];
                sum[4 * v + 0] = __builtin_fmaf(p.x, wj, sum[4 * v + 0]);
                sum[4 * v + 1] = __builtin_fmaf(p.y, wj, sum[4 * v + 1]);
                sum[4 * v + 2] = __builtin_fmaf(p.z, wj, sum[4 * v + 2]);
                sum[4 * v + 3] = __builtin_fmaf(p.w, wj, sum[4 * v + 3]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < SE_SG / 4; v++)
        reinterpret_cast<float4 *>(hp + tid * SE_SG)[v] = make_float4(sum[4 * v], sum[4 * v + 1], sum[4 * v + 2], sum[4 * v + 3]);
    __syncthreads();
    for (int idx = tid; idx < ns * Cr; idx += 256) {
        const int sg = idx / Cr, q = idx - sg * Cr;
        float acc = 0.0f;
        for (int p = 0; p < nparts; p++) acc += hp[(p * Cr + q) * SE_SG + sg];
        hpart[((size_t)ks * n_seg + seg0 + sg) * Cr + q] = acc;
    }
}

__global__ __launch_bounds__(256) void se_gate16_kernel(const float *__restrict__ hpart, int nslices, const float *__restrict__ b1, int act1,
                                                         const float *__restrict__ W2, const float *__restrict__ b2, int ld2, int act2,
                                                         float *__restrict__ gate, int n_seg, int C, int Cr) {
    __shared__ __attribute__((aligned(16))) float hidT[256 * SE_SG];          // [Cr][16]
    const int tid = threadIdx.x, seg0 = blockIdx.x * SE_SG, ns = min(SE_SG, n_seg - seg0);
    const bool has = (int)blockIdx.y * 256 + tid < C;
    const int c = min((int)blockIdx.y * 256 + tid, C - 1);          // (clamped: a thread beyond the last channel loads and drops)
    const float *w2 = W2 + c;
    float w[SE_RB];
#pragma unroll
    for (int j = 0; j < SE_RB; j++) w[j] = w2[(size_t)min(j, Cr - 1) * ld2];
    const float bias = b2[c];
    // hidden units: the group's partial sums are one contiguous run of ns x Cr floats per slice; two units x sixteen slices in flight
    const int nval = ns * Cr;
    const float *hp0 = hpart + (size_t)seg0 * Cr;
    const size_t kstride = (size_t)n_seg * Cr;
    for (int i0 = tid; i0 < SE_SG * Cr; i0 += 512) {
        const int ia = min(i0, nval - 1), ib = min(i0 + 256, nval - 1);
        const int qa = ia % Cr, qb = ib % Cr;
        float suma = b1[qa], sumb = b1[qb];
        for (int k0 = 0; k0 < nslices; k0 += 16) {
            float va[16], vb[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const size_t off = (size_t)min(k0 + k, nslices - 1) * kstride;
                va[k] = hp0[off + ia], vb[k] = hp0[off + ib];
            }
            SE_PIN(va, 16);
            SE_PIN(vb, 16);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                suma += k0 + k < nslices ? va[k] : 0.0f;
                sumb += k0 + k < nslices ? vb[k] : 0.0f;
            }
        }
        // (unit i of the run is hidden unit i % Cr of segment i / Cr)
        if (i0 < SE_SG * Cr) hidT[(i0 % Cr) * SE_SG + i0 / Cr] = i0 < nval ? act_apply(suma, act1) : 0.0f;
        if (i0 + 256 < SE_SG * Cr) hidT[((i0 + 256) % Cr) * SE_SG + (i0 + 256) / Cr] = i0 + 256 < nval ? act_apply(sumb, act1) : 0.0f;
    }
    __syncthreads();
    float s[SE_SG];
#pragma unroll
    for (int sg = 0; sg < SE_SG; sg++) s[sg] = bias;
    for (int q0 = 0; q0 < Cr; q0 += SE_RB) {
        if (q0) {
#pragma unroll
            for (int j = 0; j < SE_RB; j++) w[j] = w2[(size_t)min(q0 + j, Cr - 1) * ld2];
        }
        SE_PIN(w, SE_RB);
#pragma unroll
        for (int j = 0; j < SE_RB; j++) {
            const float wj = q0 + j < Cr ? w[j] : 0.0f;
            const float4 *hh = reinterpret_cast<const float4 *>(hidT + min(q0 + j, Cr - 1) * SE_SG);
#pragma unroll
            for (int v = 0; v < SE_SG / 4; v++) {
                const float4 h = hh[v];
                s[4 * v + 0] = __builtin_fmaf(h.x, wj, s[4 * v + 0]);
                s[4 * v + 1] = __builtin_fmaf(h.y, wj, s[4 * v + 1]);
                s[4 * v + 2] = __builtin_fmaf(h.z, wj, s[4 * v + 2]);
                s[4 * v + 3] = __builtin_fmaf(h.w, wj, s[4 * v + 3]);
            }
        }
    }
    if (has)
#pragma unroll
        for (int sg = 0; sg < SE_SG; sg++)
            if (sg < ns) gate[(size_t)(seg0 + sg) * C + c] = act_apply(s[sg], act2);
}

// (the partial hidden sums live in the caller's pooled-tensor scratch, [n][C] floats: KS x Cr <= C is asked of the block)
bool se_gate16_supports(int C, int Cr) { return C >= 1 && Cr >= 1 && Cr <= 256 && se_hidden_shape(C, Cr).slices * Cr <= C; }

void launch_se_gate16(const float *part, int tiles, int P, float *hpart, const float *W1, const float *b1, int ld1, int act1, const float *W2,
                      const float *b2, int ld2, int act2, float *gate, int n_seg, int C, int Cr, hipStream_t s) {
    const SeHiddenShape sh = se_hidden_shape(C, Cr);
    const int groups = (n_seg + SE_SG - 1) / SE_SG;
    const size_t lds = ((size_t)sh.slice * SE_SG + 256 * SE_SG) * sizeof(float);          // 32 KB at most
    hipLaunchKernelGGL(se_hidden_kernel, dim3((unsigned)groups, (unsigned)sh.slices), dim3(256), lds, s, part, tiles, 1.0f / (float)P, W1, ld1, hpart, n_seg, C, Cr,
                       sh.slice);
    hipLaunchKernelGGL(se_gate16_kernel, dim3((unsigned)groups, (unsigned)((C + 255) / 256)), dim3(256), 0, s, hpart, sh.slices, b1, act1, W2, b2, ld2, act2,
                       gate, n_seg, C, Cr);
}

// (se_gate_kernel keeps pooled [C] + partial sums [256] + hidden [Cr] in LDS and is launched without a raised dynamic-LDS limit: 64 KB)
bool se_gate_supports(int C, int Cr) { return C >= 1 && Cr >= 1 && Cr <= 256 && ((size_t)C + 256 + (size_t)Cr) * sizeof(float) <= 64 * 1024; }

// the pool alone: pooled[n][C] = (sum over tiles of part[n][tiles][C]) / P, tiles in index order (fixed)
__global__ __launch_bounds__(256) void se_pool_kernel(const float *__restrict__ part, int tiles, float inv_p, float *__restrict__ pooled, int n_seg, int C) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_seg * C) return;
    const size_t seg = i / C, c = i - seg * C;
    const float *ps = part + seg * tiles * C + c;
    float sum = 0.0f;
    for (int t = 0; t < tiles; t++) sum += ps[(size_t)t * C];
    pooled[i] = sum * inv_p;
}

// The gate for a LARGE launch: the two dense layers as GEMMs over all segments (the per-segment kernel above re-reads both weight
// matrices for every segment -- 1.8 MB at C = 2 304 -- and runs their dot products as serial loops: 0.5 us per segment for the last
// block alone), pooled and hidden in the arena slots of the layers they stand for
void launch_se_gate_gemm(const float *part, int tiles, int P, float *pooled, float *hidden, const float *W1, const float *b1, int ld1, int act1,
                         const float *W2, const float *b2, int ld2, int act2, float *gate, int n_seg, int C, int Cr, hipStream_t s) {
    const size_t n = (size_t)n_seg * C;
    hipLaunchKernelGGL(se_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, tiles, 1.0f / (float)P, pooled, n_seg, C);
    launch_pw_gemm(pooled, W1, b1, nullptr, hidden, n_seg, C, Cr, ld1, act1, s);
    launch_pw_gemm(hidden, W2, b2, nullptr, gate, n_seg, Cr, C, ld2, act2, s);
}

void launch_se_gate(const float *part, int tiles, int P, const float *W1, const float *b1, int ld1, int act1, const float *W2, const float *b2,
                    int ld2, int act2, float *gate, int n_seg, int C, int Cr, hipStream_t s) {
    int crp = 4;
    while (crp < Cr) crp <<= 1;
    const size_t lds = ((size_t)C + 256 + Cr) * sizeof(float);
    hipLaunchKernelGGL(se_gate_kernel, dim3(n_seg), dim3(256), lds, s, part, tiles, 1.0f / (float)P, W1, b1, ld1, act1, W2, b2, ld2, act2, gate, C, Cr, crp);
}

// ---------------------------------------------------------------------------------------
// Squeeze-excite gate (ONNX Mul of a feature map with a [N, C, 1, 1] tensor): HBM-bound, 16-byte accesses
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_kernel(const float4 *__restrict__ in, const float4 *__restrict__ gate,
                                                     float4 *__restrict__ out, long total4, int pc4, int c4n) {
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total4; g += (long)gridDim.x * 256) {
        const long seg = g / pc4;
        const int c4 = (int)(g % c4n);
        const float4 a = in[g], s = gate[seg * c4n + c4];
        out[g] = make_float4(a.x * s.x, a.y * s.y, a.z * s.z, a.w * s.w);
    }
}

void launch_scale(const float *in, const float *gate, float *out, int n_seg, int P, int C, hipStream_t s) {
    const int c4n = C / 4;
    const long total4 = (long)n_seg * P * c4n;
    const unsigned blocks = (unsigned)std::min<long>((total4 + 255) / 256, 256L * 64);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, s, (const float4 *)in, (const float4 *)gate, (float4 *)out, total4, P * c4n, c4n);
}

// ---------------------------------------------------------------------------------------
// activation + top-k (block per segment).  Ranks on the logit (monotone in the confidence,
// immune to saturated-sigmoid ties), ties to the lower class index -- same rule as the
// oracle's bo_topk.  idx = -1 / conf = 0 pad the unused slots.
// The row is read from HBM once into LDS; a selection pass is a per-thread scan of LDS, a wave
// reduction through lane shuffles and one barrier; a chosen class is struck out by overwriting its
// LDS entry with NaN (NaN logits are never chosen).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void topk_better(float &bv, int &bi, float ov, int oi) {
    if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
}

__global__ __launch_bounds__(256) void topk_kernel(const float *__restrict__ logits, int n_classes, int out_act,
                                                    int top_k, float min_conf, const TopkFilter flt,
                                                    int32_t *__restrict__ idx, float *__restrict__ conf,
                                                    const unsigned *__restrict__ in_bad, unsigned *__restrict__ nonfinite) {
    extern __shared__ float row[];   // n_classes logits
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float red[1];
    __shared__ int koi[32];          // kept predictions (top_k <= BH_MAX_TOP_K = 32)
    __shared__ float koc[32];
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *lg = logits + (size_t)seg * n_classes;
    float m = -INFINITY;
    bool bad = false;   // an inf / NaN logit
    // (eight loads in flight a thread: one at a time, each waiting for its own round trip, was 13 of the kernel's 27 us on a single
    //  row -- a forward of one segment is 0.42 ms in all)
    for (int i0 = tid; i0 < n_classes; i0 += 8 * 256) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v8[u] = lg[min(i0 + u * 256, n_classes - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 256;
            if (i < n_classes) { const float v = v8[u]; row[i] = v; m = fmaxf(m, v); bad |= !(fabsf(v) <= 3.4028235e38f); }
        }
    }
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    bool mark = false;   // (thread 0) this row is counted: it leaves with index -2 in its first slot (BH_TOPK_NONFINITE, birda_hip.h)
    if (nonfinite && any_bad && tid == 0) {
        // non-finite logits from FINITE samples: an operand overflowed on the way (f16 range); a segment that came in with
        // NaN / inf samples is the caller's business and is not counted
        bool in_ok = true;
        if (in_bad)
            for (int q = 0; q < 8; q++) in_ok &= in_bad[(size_t)seg * 8 + q] == 0u;
        if (in_ok) { atomicAdd(nonfinite, 1u); mark = true; }
    }
    // softmax statistics
    float mx = -INFINITY, sum = 0.f;
    if (out_act == 2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) sv[wave] = m;
        __syncthreads();
        mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        // sum of exp(logit - max): every thread its strided classes in ascending order, then a fixed tree (lanes, waves): the
        // same bits on every run and for every batch position.  (The oracle sums in class order on one thread; the two
        // orders differ by ~1e-7 relative, far inside the confidence tolerance -- and a single lane walking 14 795 classes was
        // 1.8 us per segment of the Perch-shaped model, an eighth of its whole forward.)
        float sacc = 0.f;
#pragma unroll 4
        for (int i = tid; i < n_classes; i += 256) sacc += expf(row[i] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
        __syncthreads();            // sv[] (the maxima) has been read by every thread
        if (lane == 0) sv[wave] = sacc;
        __syncthreads();
        if (tid == 0) red[0] = (sv[0] + sv[1]) + (sv[2] + sv[3]);
        __syncthreads();
        sum = red[0];
    }
    __syncthreads();
    bool stop = false;
    for (int k = 0; k < top_k; k++) {
        float bv = -INFINITY; int bi = -1;
        if (!stop) {
            for (int i0 = tid; i0 < n_classes; i0 += 8 * 256) {     // (eight LDS reads in flight, then the comparisons in index order)
                float v8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v8[u] = row[min(i0 + u * 256, n_classes - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = i0 + u * 256;
                    const float v = v8[u];
                    if (i >= n_classes || v != v) continue;          // NaN logit, or a class already chosen
                    if (bi < 0 || v > bv) { bv = v; bi = i; }   // ascending i: ties keep the lower index
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) topk_better(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        bv = sv[0]; bi = si[0];
        topk_better(bv, bi, sv[1], si[1]);
        topk_better(bv, bi, sv[2], si[2]);
        topk_better(bv, bi, sv[3], si[3]);
        float p = 0.f;
        if (bi >= 0) p = out_act == 1 ? 1.0f / (1.0f + expf(-bv)) : out_act == 2 ? expf(bv - mx) / sum : bv;
        const bool ok = !stop && bi >= 0 && p >= min_conf;
        if (!ok) stop = true;
        if (tid == 0) {
            koi[k] = ok ? bi : -1;
            koc[k] = ok ? p : 0.f;
            if (ok) row[bi] = __builtin_nanf("");
        }
        __syncthreads();
    }
    if (tid == 0) {
        int n = 0;
        while (n < top_k && koi[n] >= 0) n++;
        if (flt.bsg_intercept) {
            // BsgPostProcessor::calibrate / process on the kept predictions: logistic calibration in logit space, the SDM
            // occurrence prior when one is set, descending re-sort (stable)
            for (int i = 0; i < n; i++) {
                const int ci = koi[i];
                const float p0 = fminf(fmaxf(koc[i], 1e-7f), 1.0f - 1e-7f);
                const float lg = logf(p0 / (1.0f - p0));
                float pc = 1.0f / (1.0f + expf(-(flt.bsg_intercept[ci] + flt.bsg_slope[ci] * lg)));
                if (flt.bsg_prior) pc *= flt.bsg_prior[ci];
                koc[i] = pc;
            }
            for (int i = 1; i < n; i++) {
                const int ti = koi[i]; const float tc = koc[i];
                int j = i;
                while (j > 0 && koc[j - 1] < tc) { koi[j] = koi[j - 1]; koc[j] = koc[j - 1]; j--; }
                koi[j] = ti; koc[j] = tc;
            }
        }
        if (flt.class_score) {
            // geomodel_filter.rs:46-82: in range -> keep (scaled when reranking), out of range -> drop,
            // no geomodel entry -> keep only under the keep policy without rerank; rerank re-sorts descending
            const bool keeps = flt.keep_unmatched && !flt.rerank;
            int w = 0;
            for (int i = 0; i < n; i++) {
                const int ci = koi[i];
                const float c0 = koc[i], sc = flt.class_score[ci];
                if (sc != sc) {
                    if (keeps) { koi[w] = ci; koc[w] = c0; w++; }
                } else if (sc >= flt.threshold) {
                    koi[w] = ci; koc[w] = flt.rerank ? c0 * sc : c0; w++;
                }
            }
            if (flt.rerank)
                for (int i = 1; i < w; i++) {   // stable insertion sort, descending
                    const int ti = koi[i]; const float tc = koc[i];
                    int j = i;
                    while (j > 0 && koc[j - 1] < tc) { koi[j] = koi[j - 1]; koc[j] = koc[j - 1]; j--; }
                    koi[j] = ti; koc[j] = tc;
                }
            for (int i = w; i < n; i++) { koi[i] = -1; koc[i] = 0.f; }
        } else if (flt.species_keep) {   // classifier.rs:617-640
            int w = 0;
            for (int i = 0; i < n; i++)
                if (flt.species_keep[koi[i]]) { const int ci = koi[i]; const float c0 = koc[i]; koi[w] = ci; koc[w] = c0; w++; }
            for (int i = w; i < n; i++) { koi[i] = -1; koc[i] = 0.f; }
        }
        for (int k = 0; k < top_k; k++) {
            idx[(size_t)seg * top_k + k] = mark ? (k == 0 ? -2 : -1) : koi[k];
            conf[(size_t)seg * top_k + k] = mark ? 0.f : koc[k];
        }
    }
}

void launch_topk(const float *logits, int n_seg, int n_classes, int out_act, int top_k, float min_conf,
                 const TopkFilter &filter, int32_t *idx, float *conf, const unsigned *in_bad, unsigned *nonfinite, hipStream_t s) {
    static DeviceOnce once;
    once.run([] {
        // (the kernel also has a few static __shared__ words: ask for less than the full 160 KB)
        if (hipFuncSetAttribute((const void *)topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            (void)hipGetLastError();
    });
    hipLaunchKernelGGL(topk_kernel, dim3(n_seg), dim3(256), (size_t)n_classes * sizeof(float), s, logits, n_classes, out_act, top_k,
                       min_conf, filter, idx, conf, in_bad, nonfinite);
}

}  // namespace bh
