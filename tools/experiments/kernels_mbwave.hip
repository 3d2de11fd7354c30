// Wave-private fused MBConv block for the EARLY blocks (large images, few channels): expand 1x1 -> depthwise -> project 1x1
// (+ residual) with NO workgroup barrier in steady state.
//
// kernels_mbconv.hip has four waves cooperate on one tile and meet at two barriers per 16-channel chunk.  For the first
// blocks of the stack a chunk's three phases are a few hundred cycles each, and the waves spent 95 % of their cycles parked:
// at the barriers (each also waiting for a freshly issued L2 -> LDS weight transfer) and in per-workgroup set-up
// (tools/gpu_mb_stamps.py, tools/abl2.sh: of 1 061 us for the 16 -> 96 -> 24 block, 303 us were set-up / epilogue and 417 us
// the chunk loop WITHOUT any compute).  Weight rings and persistent workgroups with resident weights did not help: both cost a
// workgroup per CU, and occupancy was the only thing hiding those waits.
//
// Here every WAVE owns its own output tile (MT x 16 pixels), its own slice of LDS for the expanded tile (Es) and the
// depthwise output (Ds), and walks the tiles of the launch on its own (wave-level persistence): expand, depthwise and project
// of a chunk follow each other inside the wave, ordered by LDS's in-order execution per wave plus compiler fences, and the
// waves of a SIMD drift apart and fill each other's stalls.  The expand / project weight fragments of the NEXT chunk come
// from L1 / L2 straight into registers (every wave of the chip reads the same few KB) while the current chunk computes; the
// depthwise taps of every chunk are staged in LDS once per workgroup (< 6 KB) behind the kernel's only barrier.
// Same weight layouts as the 16-channel-chunk f16 instantiations of kernels_mbconv.hip (api.hip plan_fusion), same arithmetic
// (three v_mfma_f32_16x16x32_f16 per expand product, three 16x16x16 per project product, GELU through gelu_erf_fast4).
// (reference: three Conv nodes + activations + Add of the ONNX graph behind birdnet_onnx::Classifier::predict_batch,
// src/inference/classifier.rs:478-488)
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"

namespace bh {

typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MBW_CE = 16, MBW_CES = 20, MBW_DSH = 24;   // chunk width; Es row pitch (floats); Ds row pitch (halves)

__device__ __forceinline__ int mbw_div(int n, int d, float rcp_d) {
    int q = (int)((float)n * rcp_d);
    q += (n - __mul24(q, d) >= d) ? 1 : 0;
    q -= (n - __mul24(q, d) < 0) ? 1 : 0;
    return q;
}

// wave-private LDS phases: LDS executes one wave's accesses in order; this keeps the COMPILER from moving them across
__device__ __forceinline__ void mbw_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

//   KS, ST   depthwise kernel / stride        KG   32-deep k steps of the expand GEMM (ceil(Cin / 32))
//   NTO      16-column tiles of the output    RT   16-row tiles of source pixels a wave tile may have
//   MT       16-pixel tiles of output per wave tile; the tile is (16 MT >> TWL) rows x (1 << TWL) columns
//   NW       waves per workgroup              OCC  workgroups per CU the register budget must allow
//   PREC     3: split f16 (x3), 1: plain f16
template <int KS, int ST, int KG, int NTO, int RT, int MT, int TWL, int NW, int OCC, int PREC>
__global__ __launch_bounds__(NW * 64, OCC * NW / 4) void mbw_kernel(const MbDesc d, const int n_seg) {
    constexpr int TW = 1 << TWL, TH = (16 * MT) >> TWL;
    constexpr int IH = (TH - 1) * ST + KS, IW = (TW - 1) * ST + KS, M = IH * IW;
    static_assert(M <= RT * 16, "source pixels of a wave tile must fit its row tiles");
    constexpr int WE_FLOATS = KG * 512 + MBW_CE;       // hi + lo planes of one 16-column tile per k step, then be
    constexpr int WP_FLOATS = NTO * 256;               // per column tile {hi, lo}[64 lanes][4 halves]
    constexpr int WD_FLOATS = KS * KS * MBW_CE + MBW_CE;
    constexpr int ES_FLOATS = (RT * 16 + 1) * MBW_CES; // + the trash row padding rows write to
    constexpr int DS_FLOATS = MT * 16 * MBW_DSH;       // hi plane + lo plane, MBW_DSH halves per pixel each
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nchunks = d.nchunks;
    float *Wds = smem;                                 // [nchunks][WD_FLOATS]
    const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    float *Es = smem + ((nchunks * WD_FLOATS + 3) & ~3) + wave * (ES_FLOATS + DS_FLOATS);
    _Float16 *DsH = reinterpret_cast<_Float16 *>(Es + ES_FLOATS), *DsL = DsH + MT * 16 * MBW_DSH;

    for (int i = tid0; i < nchunks * WD_FLOATS; i += NW * 64) Wds[i] = d.Wd[i];
    __syncthreads();   // the only barrier: the depthwise taps of every chunk are in LDS

    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, Cin = d.Cin, Cout = d.Cout;
    const int tiles_x = (Wo + TW - 1) >> TWL, tiles_y = (Ho + TH - 1) / TH, tiles_xy = tiles_x * tiles_y;
    const int n_tiles = tiles_xy * n_seg, n_waves = gridDim.x * NW;
    const float e_unscale = d.e_unscale, p_scale = d.p_scale, p_unscale = d.p_unscale;
    const float rcp_iw = 1.0f / (float)IW;

    for (int tile = blockIdx.x * NW + wave; tile < n_tiles; tile += n_waves) {
        // (the lane index goes through an opaque copy: hipcc otherwise hoists the tile-invariant index arithmetic out of the tile
        //  loop and keeps it in registers across it)
        int lane = tid0 & 63;
        asm volatile("" : "+v"(lane));
        const int li = lane & 15, kq = lane >> 4;
        const int seg = tile / tiles_xy, txy = tile - seg * tiles_xy;
        const int tyi = txy / tiles_x, txi = txy - tyi * tiles_x;
        const int oy0 = tyi * TH, ox0 = txi << TWL;
        const int iy0 = oy0 * ST - d.pad_t, ix0 = ox0 * ST - d.pad_l;
        const float *Xb = d.X + (size_t)seg * H * W * Cin;

        // ---- the tile's source rows: B fragments of the expand GEMM, resident for every chunk ----
        wf16x8 xh[RT][KG], xl[RT][KG];
        int eoff[RT];       // this lane's 16-byte slot in Es for each of its rows (trash row for rows past M)
        unsigned in_img = 0;
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const int m = i * 16 + li;
            const int r = mbw_div(m, IW, rcp_iw), c = m - r * IW;
            const int y = iy0 + r, x = ix0 + c;
            const bool ok = m < M && y >= 0 && y < H && x >= 0 && x < W;
            in_img |= ok ? (1u << i) : 0u;
            eoff[i] = (m < M ? m : RT * 16) * MBW_CES + 4 * kq;
            const float *xp = Xb + ((size_t)(ok ? y : 0) * W + (ok ? x : 0)) * Cin;
#pragma unroll
            for (int g = 0; g < KG; g++) {
                const int k0 = 32 * g + 8 * kq;
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const float4 t = (ok && k0 + 4 * q < Cin) ? *reinterpret_cast<const float4 *>(xp + k0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
                }
                if (PREC == 3) bh_split8(v, xh[i][g], xl[i][g]);
                else {
#pragma unroll
                    for (int q = 0; q < 8; q++) xh[i][g][q] = (_Float16)v[q];
                }
            }
        }
        // ---- project accumulators start at (bias + residual) 2^sp; pixel of accumulator row 4 kq + r of tile i ----
        wf32x4 acco[MT][NTO];
        int opix[MT][4];
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int p = i * 16 + 4 * kq + r, oy = oy0 + (p >> TWL), ox = ox0 + (p & (TW - 1));
                opix[i][r] = (oy < Ho && ox < Wo) ? ((seg * Ho + oy) * Wo + ox) : -1;
            }
#pragma unroll
        for (int j = 0; j < NTO; j++) {
            const int col = 16 * j + li;
            const float bias = col < Cout ? d.bp[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < MT; i++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acco[i][j][r] = __builtin_fmaf((d.R && col < Cout && opix[i][r] >= 0) ? d.R[(size_t)opix[i][r] * Cout + col] : 0.0f, p_scale, bias);
        }

        // ---- weight fragments of chunk 0; inside the loop chunk ch + 1 is fetched while chunk ch computes ----
        wf16x8 weh[KG], wel[KG], nweh[KG], nwel[KG];
        wf16x4 wph[NTO], wpl[NTO], nwph[NTO], nwpl[NTO];
        wf32x4 be4, nbe4;
        auto fetch = [&](int ch, wf16x8 (&eh)[KG], wf16x8 (&el)[KG], wf16x4 (&ph)[NTO], wf16x4 (&pl)[NTO], wf32x4 &b4) {
            const wf16x8 *we = reinterpret_cast<const wf16x8 *>(d.We + (size_t)ch * WE_FLOATS);
#pragma unroll
            for (int g = 0; g < KG; g++) {
                eh[g] = we[(g * 2 + 0) * 64 + lane];
                if (PREC == 3) el[g] = we[(g * 2 + 1) * 64 + lane];
            }
            b4 = *reinterpret_cast<const wf32x4 *>(d.We + (size_t)ch * WE_FLOATS + KG * 512 + 4 * kq);
            const wf16x4 *wp = reinterpret_cast<const wf16x4 *>(d.Wp + (size_t)ch * WP_FLOATS);
#pragma unroll
            for (int j = 0; j < NTO; j++) {
                ph[j] = wp[(j * 2 + 0) * 64 + lane];
                if (PREC == 3) pl[j] = wp[(j * 2 + 1) * 64 + lane];
            }
        };
        fetch(0, weh, wel, wph, wpl, be4);

        for (int ch = 0; ch < nchunks; ch++) {
            if (ch + 1 < nchunks) fetch(ch + 1, nweh, nwel, nwph, nwpl, nbe4);
            // ---- P1: expand, transposed (E^T = We^T X^T): the lane gets channels 4 kq .. 4 kq + 3 of source row li ----
#pragma unroll
            for (int i = 0; i < RT; i++) {
                if (i * 16 < M) {
                    wf32x4 acc = be4;
#pragma unroll
                    for (int g = 0; g < KG; g++) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(weh[g], xh[i][g], acc, 0, 0, 0);
                        if (PREC == 3) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wel[g], xh[i][g], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(weh[g], xl[i][g], acc, 0, 0, 0);
                        }
                    }
                    wf32x2 v01 = {acc[0], acc[1]}, v23 = {acc[2], acc[3]};
                    v01 *= e_unscale; v23 *= e_unscale;
                    gelu_erf_fast4(v01, v23);
                    const bool ok = (in_img >> i) & 1u;   // positions outside the image are the depthwise conv's zero padding
                    *reinterpret_cast<wf32x4 *>(Es + eoff[i]) = ok ? (wf32x4){v01[0], v01[1], v23[0], v23[1]} : (wf32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            mbw_lds_fence();
            // ---- P2: depthwise + GELU: MT tasks per lane, each 4 channels of one output pixel ----
            const float *WdC = Wds + ch * WD_FLOATS;
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const int task = t * 64 + lane, px = task >> 2, c4 = task & 3;
                const int py = px >> TWL, pxx = px & (TW - 1);
                const float *eb = Es + ((py * ST) * IW + pxx * ST) * MBW_CES + 4 * c4;
                const float4 bd4 = *reinterpret_cast<const float4 *>(WdC + KS * KS * MBW_CE + 4 * c4);
                wf32x2 a0 = {bd4.x, bd4.y}, a1 = {bd4.z, bd4.w};
#pragma unroll
                for (int dy = 0; dy < KS; dy++) {
#pragma unroll
                    for (int dx = 0; dx < KS; dx++) {
                        const float4 e = *reinterpret_cast<const float4 *>(eb + (dy * IW + dx) * MBW_CES);
                        const float4 w = *reinterpret_cast<const float4 *>(WdC + (dy * KS + dx) * MBW_CE + 4 * c4);
                        a0 = __builtin_elementwise_fma((wf32x2){e.x, e.y}, (wf32x2){w.x, w.y}, a0);
                        a1 = __builtin_elementwise_fma((wf32x2){e.z, e.w}, (wf32x2){w.z, w.w}, a1);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one kernel row of loads in flight, not all of them
                }
                gelu_erf_fast4(a0, a1);
                if (PREC == 3) {
                    bh_f16x2 h0, l0, h1, l1;
                    bh_split2(a0[0], a0[1], h0, l0);
                    bh_split2(a1[0], a1[1], h1, l1);
                    *reinterpret_cast<wf16x4 *>(&DsH[px * MBW_DSH + 4 * c4]) = (wf16x4){h0[0], h0[1], h1[0], h1[1]};
                    *reinterpret_cast<wf16x4 *>(&DsL[px * MBW_DSH + 4 * c4]) = (wf16x4){l0[0], l0[1], l1[0], l1[1]};
                } else {
                    *reinterpret_cast<wf16x4 *>(&DsH[px * MBW_DSH + 4 * c4]) = (wf16x4){(_Float16)a0[0], (_Float16)a0[1], (_Float16)a1[0], (_Float16)a1[1]};
                }
            }
            mbw_lds_fence();
            // ---- P3: project, one 16-deep step per chunk ----
#pragma unroll
            for (int i = 0; i < MT; i++) {
                const wf16x4 ah = *reinterpret_cast<const wf16x4 *>(&DsH[(i * 16 + li) * MBW_DSH + 4 * kq]);
                wf16x4 al;
                if (PREC == 3) al = *reinterpret_cast<const wf16x4 *>(&DsL[(i * 16 + li) * MBW_DSH + 4 * kq]);
#pragma unroll
                for (int j = 0; j < NTO; j++) {
                    acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, wph[j], acco[i][j], 0, 0, 0);
                    if (PREC == 3) {
                        acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, wpl[j], acco[i][j], 0, 0, 0);
                        acco[i][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, wph[j], acco[i][j], 0, 0, 0);
                    }
                }
            }
            mbw_lds_fence();   // the next chunk's P1 rewrites Es, its P2 rewrites Ds
            if (ch + 1 < nchunks) {
#pragma unroll
                for (int g = 0; g < KG; g++) { weh[g] = nweh[g]; wel[g] = nwel[g]; }
#pragma unroll
                for (int j = 0; j < NTO; j++) { wph[j] = nwph[j]; wpl[j] = nwpl[j]; }
                be4 = nbe4;
            }
        }
        // ---- store (bias and residual are in the accumulators) ----
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NTO; j++) {
                const int col = 16 * j + li;
                if (col >= Cout) continue;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (opix[i][r] >= 0) d.Y[(size_t)opix[i][r] * Cout + col] = acco[i][j][r] * p_unscale;
            }
    }
}

struct MbwCfg {
    int KS, ST, KG, NTO, RT, MT, TWL, NW, OCC, PREC;
    void (*launch)(const MbDesc &, int, hipStream_t);
};

template <int KS, int ST, int KG, int NTO, int RT, int MT, int TWL, int NW, int OCC, int PREC>
void mbw_launch(const MbDesc &d, int n_seg, hipStream_t s) {
    auto kern = mbw_kernel<KS, ST, KG, NTO, RT, MT, TWL, NW, OCC, PREC>;
    static DeviceOnce attr_set;
    attr_set.run([&] { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    constexpr int TW = 1 << TWL, TH = (16 * MT) >> TWL;
    const long tiles = (long)((d.Wo + TW - 1) / TW) * ((d.Ho + TH - 1) / TH) * n_seg;
    const long per_cu = std::max<long>(1, std::min<long>(OCC, (160 * 1024) / (long)(d.lds_bytes + 256)));
    const long wgs = std::min<long>((tiles + NW - 1) / NW, per_cu * device_cu_count());
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(NW * 64), d.lds_bytes, s, d, n_seg);
}

#define MBW_ENTRY(KS, ST, KG, NTO, RT, MT, TWL, NW, OCC)                                                           \
    {KS, ST, KG, NTO, RT, MT, TWL, NW, OCC, 3, mbw_launch<KS, ST, KG, NTO, RT, MT, TWL, NW, OCC, 3>},               \
    {KS, ST, KG, NTO, RT, MT, TWL, NW, OCC, 1, mbw_launch<KS, ST, KG, NTO, RT, MT, TWL, NW, OCC, 1>}

//             KS ST KG NTO RT MT TWL NW OCC      tile rows x columns: (16 MT >> TWL) x (1 << TWL)
const MbwCfg kWaveCfgs[] = {
    MBW_ENTRY(3, 2, 1, 2, 6, 1, 3, 8, 2),    // 0/1: 16 -> 96 -> 24, 48x256 -> 24x128: tile 2x8, source 5x17 = 85 rows
    MBW_ENTRY(3, 1, 1, 2, 7, 4, 3, 8, 1),    // 2/3: 24 -> 144 -> 24 at 24x128: tile 8x8, source 10x10 = 100 rows
    MBW_ENTRY(5, 2, 1, 3, 9, 1, 3, 8, 2),    // 4/5: 24 -> 144 -> 40, 24x128 -> 12x64: tile 2x8, source 7x19 = 133 rows
    MBW_ENTRY(5, 1, 2, 3, 9, 4, 3, 8, 1),    // 6/7: 40 -> 240 -> 40 at 12x64: tile 8x8, source 12x12 = 144 rows
};
constexpr int kNWaveCfgs = (int)(sizeof(kWaveCfgs) / sizeof(kWaveCfgs[0]));

}  // namespace

// Picks a wave-private instantiation for this block (non-stem, GELU, f16 modes, large image) and fills the derived fields the
// weight packer (api.hip plan_fusion) and the launcher use.  d.cfg = -2 - index marks it.
bool mbw_plan(MbDesc &d) {
    // Measured slower than the cooperative kernel on every early block (16->96->24: 1 297 vs 1 063 us per 1 000 segments,
    // 24->144->24: 1 034 vs 744, 24->144->40 5x5: 1 610 vs 608): a wave-sized tile recomputes 1.5 - 1.75x the expanded pixels
    // (halo), and the expand GELU is the dominant VALU cost.  Opt-in: BIRDA_HIP_MB_WAVE=1.
    const char *e = getenv("BIRDA_HIP_MB_WAVE");
    if (!e || e[0] != '1') return false;
    if (d.stem || d.prec == 0 || d.act_e != ACT_GELU_ERF || d.act_d != ACT_GELU_ERF || d.act_p != ACT_NONE) return false;
    if (d.Cin % 4 || d.Ho * d.Wo < 512) return false;   // the early blocks only: later ones have too few pixels per weight byte
    for (int ci = 0; ci < kNWaveCfgs; ci++) {
        const MbwCfg &c = kWaveCfgs[ci];
        if (c.KS != d.KS || c.ST != d.ST || c.PREC != d.prec || (d.Cin + 31) / 32 != c.KG || (d.Cout + 15) / 16 != c.NTO) continue;
        const int nchunks = (d.Cexp + MBW_CE - 1) / MBW_CE;
        const size_t wd = (size_t)c.KS * c.KS * MBW_CE + MBW_CE;
        const size_t per_wave = (size_t)(c.RT * 16 + 1) * MBW_CES + (size_t)c.MT * 16 * MBW_DSH;
        const size_t lds = ((((size_t)nchunks * wd + 3) & ~(size_t)3) + (size_t)c.NW * per_wave) * sizeof(float);
        if (lds > 160 * 1024) continue;
        d.cfg = -2 - ci;
        d.CE = MBW_CE; d.KG = c.KG; d.NTOP = c.NTO; d.nchunks = nchunks; d.S = 1;
        d.TH = (16 * c.MT) >> c.TWL;
        d.tiles_x = (d.Wo + (1 << c.TWL) - 1) >> c.TWL; d.tiles_y = (d.Ho + d.TH - 1) / d.TH;
        d.IH = (d.TH - 1) * c.ST + c.KS; d.IW = ((1 << c.TWL) - 1) * c.ST + c.KS;
        d.mpad_max = c.RT * 16; d.ring = 0;
        d.lds_bytes = lds;
        return true;
    }
    return false;
}

void launch_mbwave(const MbDesc &d, int n_seg, hipStream_t s) { kWaveCfgs[-2 - d.cfg].launch(d, n_seg, s); }

int mbw_config_name(int cfg, char *out, size_t cap) {
    const int ci = -2 - cfg;
    if (ci < 0 || ci >= kNWaveCfgs) return 0;
    const MbwCfg &c = kWaveCfgs[ci];
    return snprintf(out, cap, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", c.KS, c.ST, c.KG, c.NTO, c.RT, c.MT, c.TWL, c.NW, c.OCC, c.PREC);
}

}  // namespace bh
