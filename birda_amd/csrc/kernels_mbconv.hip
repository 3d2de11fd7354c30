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
//     P1  E[M x CE]   = act(X[M x Cin] . We[Cin x CE] + be)   f32 MFMA; A straight from global/L2,
//                       B fragment-major from L2; rows scattered into the LDS grid Es (the grid's
//                       out-of-image border stays zero = the depthwise conv's zero padding)
//     P2  D[P x CE]   = act(dw_kxk(Es) + bd)                  VALU + LDS, 4 channels per lane
//     P3  acc[P x Co] += D[P x CE] . Wp[CE x Co]              f32 MFMA, accumulators live across chunks
//   epilogue: + bp, activation, + residual, NHWC store.
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace bh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float mb_act(float v, int act) {
    switch (act) {
    case ACT_NONE: return v;
    case ACT_GELU_ERF: return gelu_erf_fast(v);
    default: return act_apply_slow(v, act);
    }
}

// diagnostic phase clock (only when d.stamps != nullptr): cycles since the last stamp are added to
// slot `ph` by lane 0 of every wave
__device__ __forceinline__ void mb_stamp(unsigned long long *stamps, unsigned long long &t_last, int ph) {
    if (!stamps) return;
    const unsigned long long now = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) atomicAdd(&stamps[ph], now - t_last);
    t_last = now;
}

template <int KS, int ST, int CE, int NCS, int WM, int WN, int MT_W, int NT_W, int TWL>
__global__ __launch_bounds__(256) void mbconv_kernel(const MbDesc d, const int n_seg) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int NT_E = CE / 16, NT_U = NT_E / NCS, CES = CE + 4, C4N = CE / 4, TW = 1 << TWL;
    constexpr int POUT_PAD = WM * MT_W * 16, NTOP = WN * NT_W;
    static_assert(NT_U * NCS == NT_E, "column split");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int IH = d.IH, IW = d.IW, S = d.S, TH = d.TH, THTW = d.TH << TWL;
    const int egrid = S * IH * IW;
    float *Es = smem;
    float *Ds = Es + (size_t)egrid * CES;
    float *Wds = Ds + POUT_PAD * CES;
    float *bds = Wds + KS * KS * CE;
    int *emap = reinterpret_cast<int *>(bds + CE);
    int *xoff = emap + d.mpad_max;
    int *omap = xoff + d.mpad_max;

    unsigned long long t_last = d.stamps ? __builtin_readcyclecounter() : 0ull;
    const int tyi = blockIdx.x / d.tiles_x, txi = blockIdx.x - tyi * d.tiles_x;
    const int seg0 = blockIdx.y * S;
    const int nsv = min(S, n_seg - seg0);
    const int oy0 = tyi * TH, ox0 = txi * TW;
    const int iy0 = oy0 * ST - d.pad_t, ix0 = ox0 * ST - d.pad_l;
    const int ya = max(0, -iy0), yb = min(IH, d.H - iy0);
    const int xa = max(0, -ix0), xb = min(IW, d.W - ix0);
    const int vh = max(yb - ya, 0), vw = max(xb - xa, 0);
    const int Mseg = vh * vw, M = Mseg * nsv, nrt = (M + 15) >> 4;
    const int Cin = d.Cin, Cout = d.Cout, KG = d.KG;
    const float *Xb = d.X + (size_t)seg0 * d.H * d.W * Cin;

    for (int m = tid; m < nrt * 16; m += 256) {
        int e = -1, xo = 0;
        if (m < M) {
            const int sl = m / Mseg, mm = m - sl * Mseg;
            const int r = mm / vw, c = mm - r * vw;
            e = sl * IH * IW + (ya + r) * IW + xa + c;
            xo = ((sl * d.H + iy0 + ya + r) * d.W + ix0 + xa + c) * Cin;
        }
        emap[m] = e;
        xoff[m] = xo;
    }
    for (int p = tid; p < POUT_PAD; p += 256) {
        const int sl = p / THTW, pp = p - sl * THTW;
        const int ty = pp >> TWL, tx = pp & (TW - 1);
        int o = -1;
        if (sl < nsv && oy0 + ty < d.Ho && ox0 + tx < d.Wo) o = (sl * d.Ho + oy0 + ty) * d.Wo + ox0 + tx;
        omap[p] = o;
    }
    {
        float4 *z = reinterpret_cast<float4 *>(Es);
        const int n4 = egrid * CES / 4;
        for (int i = tid; i < n4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    mb_stamp(d.stamps, t_last, 0);

    const int wm = wave / WN, wn = wave - wm * WN;
    f32x4 acco[MT_W][NT_W];
#pragma unroll
    for (int i = 0; i < MT_W; i++)
#pragma unroll
        for (int j = 0; j < NT_W; j++) acco[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ch = 0; ch < d.nchunks; ch++) {
        // depthwise weights + bias of this chunk -> LDS (free since the barrier after the last P2)
        for (int i = tid; i < KS * KS * C4N; i += 256) {
            const int tap = i / C4N, c4 = i - tap * C4N;
            *reinterpret_cast<float4 *>(&Wds[tap * CE + 4 * c4]) =
                *reinterpret_cast<const float4 *>(&d.Wd[(size_t)tap * d.Cexp + ch * CE + 4 * c4]);
        }
        if (tid < C4N)
            *reinterpret_cast<float4 *>(&bds[4 * tid]) = *reinterpret_cast<const float4 *>(&d.bd[ch * CE + 4 * tid]);

        mb_stamp(d.stamps, t_last, 1);
        // ---- P1: expand ------------------------------------------------------------------
        {
            const float4 *WeF = reinterpret_cast<const float4 *>(d.We) + (size_t)ch * KG * NT_E * 64 + lane;
            for (int u = wave; u < nrt * NCS; u += 4) {
                const int rt = u / NCS, cs = u - rt * NCS;
                const float *xp = Xb + xoff[rt * 16 + li] + 4 * kq;
                f32x4 acc[NT_U];
#pragma unroll
                for (int j = 0; j < NT_U; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                float4 a_cur = (4 * kq < Cin) ? *reinterpret_cast<const float4 *>(xp) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 b_cur[NT_U], b_nxt[NT_U];
#pragma unroll
                for (int j = 0; j < NT_U; j++) b_cur[j] = WeF[(cs * NT_U + j) * 64];
                for (int g = 0; g < KG; g++) {
                    const int gn = min(g + 1, KG - 1);
                    const float4 a_nxt = (16 * gn + 4 * kq < Cin) ? *reinterpret_cast<const float4 *>(xp + 16 * gn)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NT_U; j++) b_nxt[j] = WeF[(gn * NT_E + cs * NT_U + j) * 64];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float a = c == 0 ? a_cur.x : c == 1 ? a_cur.y : c == 2 ? a_cur.z : a_cur.w;
#pragma unroll
                        for (int j = 0; j < NT_U; j++) {
                            const float b = c == 0 ? b_cur[j].x : c == 1 ? b_cur[j].y : c == 2 ? b_cur[j].z : b_cur[j].w;
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
                        }
                    }
                    a_cur = a_nxt;
#pragma unroll
                    for (int j = 0; j < NT_U; j++) b_cur[j] = b_nxt[j];
                }
                const int4 e4 = *reinterpret_cast<const int4 *>(&emap[rt * 16 + 4 * kq]);
                const int er[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                for (int j = 0; j < NT_U; j++) {
                    const int col = (cs * NT_U + j) * 16 + li;
                    const float bias = d.be[ch * CE + col];
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (er[r] >= 0) Es[er[r] * CES + col] = mb_act(acc[j][r] + bias, d.act_e);
                }
            }
        }
        mb_stamp(d.stamps, t_last, 2);
        __syncthreads();
        mb_stamp(d.stamps, t_last, 3);

        // ---- P2: depthwise ---------------------------------------------------------------
        for (int sl = 0; sl < nsv; sl++) {
            const float *eseg = Es + (size_t)sl * IH * IW * CES;
            for (int t = tid; t < THTW * C4N; t += 256) {
                const int p = t / C4N, c4 = t - p * C4N;
                const int ty = p >> TWL, tx = p & (TW - 1);
                const float *eb = eseg + ((ty * ST) * IW + tx * ST) * CES + 4 * c4;
                float4 acc = *reinterpret_cast<const float4 *>(&bds[4 * c4]);
#pragma unroll
                for (int dy = 0; dy < KS; dy++) {
                    const float *er = eb + dy * IW * CES;
#pragma unroll
                    for (int dx = 0; dx < KS; dx++) {
                        const float4 e = *reinterpret_cast<const float4 *>(er + dx * CES);
                        const float4 w = *reinterpret_cast<const float4 *>(&Wds[(dy * KS + dx) * CE + 4 * c4]);
                        acc.x += e.x * w.x; acc.y += e.y * w.y; acc.z += e.z * w.z; acc.w += e.w * w.w;
                    }
                }
                acc.x = mb_act(acc.x, d.act_d); acc.y = mb_act(acc.y, d.act_d);
                acc.z = mb_act(acc.z, d.act_d); acc.w = mb_act(acc.w, d.act_d);
                *reinterpret_cast<float4 *>(&Ds[(sl * THTW + p) * CES + 4 * c4]) = acc;
            }
        }
        mb_stamp(d.stamps, t_last, 4);
        __syncthreads();
        mb_stamp(d.stamps, t_last, 5);

        // ---- P3: project -----------------------------------------------------------------
        {
            const float4 *WpF = reinterpret_cast<const float4 *>(d.Wp) + ((size_t)ch * NT_E * NTOP + wn * NT_W) * 64 + lane;
            const float *dsb = Ds + ((wm * MT_W) * 16 + li) * CES + 4 * kq;
#pragma unroll
            for (int g = 0; g < NT_E; g++) {
                float4 a[MT_W], b[NT_W];
#pragma unroll
                for (int j = 0; j < NT_W; j++) b[j] = WpF[(g * NTOP + j) * 64];
#pragma unroll
                for (int i = 0; i < MT_W; i++) a[i] = *reinterpret_cast<const float4 *>(dsb + i * 16 * CES + 16 * g);
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
            }
        }
        mb_stamp(d.stamps, t_last, 6);
        // no barrier here: the next chunk's P1 touches Es / Wds only, and every wave has passed the
        // barrier after P2; its P2 (which rewrites Ds) sits behind the barrier after P1.
    }

    // ---- epilogue: bias, activation, residual, store -----------------------------------------
    float *Yb = d.Y + (size_t)seg0 * d.Ho * d.Wo * Cout;
    const float *Rb = d.R ? d.R + (size_t)seg0 * d.Ho * d.Wo * Cout : nullptr;
#pragma unroll
    for (int i = 0; i < MT_W; i++) {
        const int4 o4 = *reinterpret_cast<const int4 *>(&omap[(wm * MT_W + i) * 16 + 4 * kq]);
        const int orow[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int j = 0; j < NT_W; j++) {
            const int col = (wn * NT_W + j) * 16 + li;
            if (col >= Cout) continue;
            const float bias = d.bp[col];
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (orow[r] >= 0) {
                    const size_t idx = (size_t)orow[r] * Cout + col;
                    float v = mb_act(acco[i][j][r] + bias, d.act_p);
                    if (Rb) v += Rb[idx];
                    Yb[idx] = v;
                }
        }
    }
    mb_stamp(d.stamps, t_last, 7);
}

struct MbCfg {
    int KS, ST, CE, NCS, WM, WN, MT_W, NT_W, TWL, TH, S;
    void (*launch)(const MbDesc &, int, hipStream_t);
};

template <int KS, int ST, int CE, int NCS, int WM, int WN, int MT_W, int NT_W, int TWL>
void mb_launch(const MbDesc &d, int n_seg, hipStream_t s) {
    auto kern = mbconv_kernel<KS, ST, CE, NCS, WM, WN, MT_W, NT_W, TWL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(d.tiles_y * d.tiles_x, (n_seg + d.S - 1) / d.S), block(256);
    hipLaunchKernelGGL(kern, grid, block, d.lds_bytes, s, d, n_seg);
}

#define MB_ENTRY(KS, ST, CE, NCS, WM, WN, MT_W, NT_W, TWL, TH, S) \
    {KS, ST, CE, NCS, WM, WN, MT_W, NT_W, TWL, TH, S, mb_launch<KS, ST, CE, NCS, WM, WN, MT_W, NT_W, TWL>}

// The instantiations cover the BirdNET-v2.4 / Perch-shaped stacks (EfficientNet-B0 stages);
// mb_plan() picks, per block, the valid entry with the least MFMA work.
const MbCfg kCfgs[] = {
    MB_ENTRY(3, 2, 48, 1, 4, 1, 1, 2, 4, 4, 1),    // 0: 48x256 -> 24x128, Cout <= 32
    MB_ENTRY(3, 1, 48, 1, 4, 1, 2, 2, 4, 8, 1),    // 1: 24x128 s1, Cout <= 32
    MB_ENTRY(5, 2, 48, 1, 4, 1, 1, 3, 4, 4, 1),    // 2: 24x128 -> 12x64, Cout <= 48
    MB_ENTRY(5, 1, 48, 1, 4, 1, 3, 3, 4, 12, 1),   // 3: 12x64 s1 full-height tiles, Cout <= 48
    MB_ENTRY(3, 2, 48, 1, 2, 2, 3, 3, 4, 6, 1),    // 4: 12x64 -> 6x32, Cout <= 96
    MB_ENTRY(3, 1, 32, 1, 4, 1, 3, 5, 5, 6, 1),    // 5: 6x32 whole image, Cout <= 80
    MB_ENTRY(5, 1, 32, 1, 4, 1, 3, 7, 5, 6, 1),    // 6: 6x32 whole image, Cout <= 112
    MB_ENTRY(5, 2, 32, 1, 1, 4, 3, 3, 4, 3, 1),    // 7: 6x32 -> 3x16, Cout <= 192
    MB_ENTRY(5, 1, 32, 2, 2, 2, 3, 6, 4, 3, 2),    // 8: 3x16 x 2 segments, Cout <= 192
    MB_ENTRY(3, 1, 32, 2, 2, 2, 3, 10, 4, 3, 2),   // 9: 3x16 x 2 segments, Cout <= 320
};
constexpr int kNCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

// fills the derived fields for entry `ci`; returns the estimated MFMA work per segment (in
// 16x16x4 steps), or -1 when the entry cannot run this block
double mb_try(MbDesc &d, int ci) {
    const MbCfg &c = kCfgs[ci];
    if (c.KS != d.KS || c.ST != d.ST || d.Cexp % c.CE || d.Cin % 4 || d.Cexp % 4) return -1;
    const int nto = (d.Cout + 15) / 16;
    if (nto > c.WN * c.NT_W) return -1;
    const int TW = 1 << c.TWL;
    MbDesc t = d;
    t.cfg = ci; t.TH = c.TH; t.S = c.S;
    t.tiles_y = (d.Ho + c.TH - 1) / c.TH; t.tiles_x = (d.Wo + TW - 1) / TW;
    t.IH = (c.TH - 1) * c.ST + c.KS; t.IW = (TW - 1) * c.ST + c.KS;
    t.KG = (d.Cin + 15) / 16; t.nchunks = d.Cexp / c.CE; t.NTOP = c.WN * c.NT_W; t.CE = c.CE;
    const int mseg = std::min(t.IH, d.H) * std::min(t.IW, d.W);
    t.mpad_max = (c.S * mseg + 15) / 16 * 16;
    const int ces = c.CE + 4, pout_pad = c.WM * c.MT_W * 16;
    t.lds_bytes = ((size_t)c.S * t.IH * t.IW * ces + (size_t)pout_pad * ces + (size_t)c.KS * c.KS * c.CE + c.CE) * 4 +
                  ((size_t)2 * t.mpad_max + pout_pad) * 4;
    if (t.lds_bytes > 160 * 1024) return -1;
    d = t;
    const double tiles = (double)t.tiles_y * t.tiles_x / c.S;
    return tiles * ((double)t.mpad_max / 16 * t.KG * 4 * (d.Cexp / 16) + (double)pout_pad / 16 * t.NTOP * (d.Cexp / 4));
}

}  // namespace

int mb_config_count() { return kNCfgs; }

bool mb_plan(MbDesc &d, int force_cfg) {
    d.cfg = -1;
    if (force_cfg >= 0) {
        if (force_cfg >= kNCfgs) return false;
        MbDesc t = d;
        if (mb_try(t, force_cfg) < 0) return false;
        d = t;
        return true;
    }
    double best = -1;
    MbDesc bestd = d;
    for (int ci = 0; ci < kNCfgs; ci++) {
        MbDesc t = d;
        const double w = mb_try(t, ci);
        if (w < 0) continue;
        if (best < 0 || w < best) { best = w; bestd = t; }
    }
    if (best < 0) return false;
    d = bestd;
    return true;
}

void launch_mbconv(const MbDesc &d, int n_seg, hipStream_t s) { kCfgs[d.cfg].launch(d, n_seg, s); }

}  // namespace bh
