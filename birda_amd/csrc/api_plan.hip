// api_plan.hip -- host-side planning at classifier create (split out of api.hip in round 4, api_internal.hpp): the folded STFT x mel
// operator of a front-end branch, the liveness plan of the activation arena, and the fused MBConv blocks -- which layer triples run
// as one launch (reference: the Conv / activation / Add nodes birdnet_onnx::Classifier hands to ONNX Runtime,
// src/inference/classifier.rs:478-488), their tile configuration and their weights re-laid as per-chunk LDS-DMA blocks.
#include "api_internal.hpp"

using namespace bhi;

namespace bhi {


uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

// Gf for one branch (see kernels_frontend.hip): double precision on the host, once.
// prec 0: f32 fragment-major; prec 3: f16 hi / lo planes for the split MFMA (same byte count).
std::vector<float> build_gf(const bh::BranchRec &b, const float *W, int nm_pad, int prec, int *scale_exp) {
    const int L = (int)b.frame_length, K = L / 2, nb = (int)b.n_bins, nm = (int)b.n_mels;
    std::vector<double> ct(L);
    for (int i = 0; i < L; i++) ct[i] = std::cos(2.0 * M_PI * (double)i / (double)L);
    std::vector<int> rows;
    for (int k = 0; k < nb; k++) {
        bool nz = false;
        for (int m = 0; m < nm && !nz; m++) nz = W[(size_t)k * nm + m] != 0.0f;
        if (nz) rows.push_back(k);
    }
    std::vector<float> gf((size_t)K * nm_pad, 0.0f);
    std::vector<double> acc(nm);
    for (int j = 0; j < K; j++) {
        const int n = j + 1;
        const double wn = 0.5 - 0.5 * ct[n % L];
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int k : rows) {
            const double cv = ct[(size_t)((long long)k * n % L)];
            const float *wr = W + (size_t)k * nm;
            for (int m = 0; m < nm; m++) acc[m] += cv * (double)wr[m];
        }
        const double scale = (j == K - 1) ? 0.5 * wn : wn;  // the centre sample is added to itself
        for (int m = 0; m < nm; m++) gf[(size_t)j * nm_pad + m] = (float)(scale * acc[m]);
    }
    // f16 planes hold Gf * 2^s (kernels.hpp f16_scale_exponent); the kernel's power law undoes it (BranchParams::log2_bias)
    *scale_exp = 0;
    if (prec != 0) {
        float mx = 0.0f;
        for (float v : gf) mx = std::max(mx, std::fabs(v));
        *scale_exp = bh::f16_scale_exponent(mx);
        for (float &v : gf) v = std::ldexp(v, *scale_exp);
    }
    // MFMA-fragment-major relayout (kernels.hpp BranchParams::gf)
    const int mt_n = nm_pad / 16;
    std::vector<float> frag((size_t)K * nm_pad);
    if (prec == 32) {  // mel32_kernel: [step of 16 k][mel tile of 32][plane hi, lo][64 lanes][8 halves]; within a chunk of 64 k the
        // staged Y rows pair k with k + 32 in one dword, so element jj of step s holds k = 64 (s / 4) + 8 (s % 4) + 4 (lane >> 5) + jj / 2 + 32 (jj % 2)
        uint16_t *h = reinterpret_cast<uint16_t *>(frag.data());
        const int mt32 = nm_pad / 32;
        for (int st = 0; st < K / 16; st++)
            for (int mt = 0; mt < mt32; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int jj = 0; jj < 8; jj++) {
                        const float v = gf[(size_t)(64 * (st >> 2) + 8 * (st & 3) + 4 * (lane >> 5) + (jj >> 1) + 32 * (jj & 1)) * nm_pad + 32 * mt + (lane & 31)];
                        const uint16_t hi = f32_to_f16(v);
                        const size_t base = (((size_t)st * mt32 + mt) * 2) * 64 * 8;
                        h[base + (size_t)lane * 8 + jj] = hi;
                        h[base + 64 * 8 + (size_t)lane * 8 + jj] = f32_to_f16(v - f16_to_f32(hi));
                    }
        return frag;
    }
    if (prec != 0) {  // [step of 32 k][mel tile][plane hi, lo][64 lanes][8 halves]: k = 32 s + 4 jj + (lane >> 4)
        // (the k of a step are dealt to the four lane groups round-robin, not in runs of 8: the kernel's frame-strided LDS
        //  reads of the matching samples then fall on distinct banks -- kernels_frontend.hip, mel_kernel)
        uint16_t *h = reinterpret_cast<uint16_t *>(frag.data());
        for (int st = 0; st < K / 32; st++)
            for (int mt = 0; mt < mt_n; mt++)
                for (int lane = 0; lane < 64; lane++)
                    for (int jj = 0; jj < 8; jj++) {
                        const float v = gf[(size_t)(32 * st + 4 * jj + (lane >> 4)) * nm_pad + 16 * mt + (lane & 15)];
                        const uint16_t hi = f32_to_f16(v);
                        const size_t base = (((size_t)st * mt_n + mt) * 2) * 64 * 8;
                        h[base + (size_t)lane * 8 + jj] = hi;
                        h[base + 64 * 8 + (size_t)lane * 8 + jj] = f32_to_f16(v - f16_to_f32(hi));
                    }
        return frag;
    }
    for (int g = 0; g < K / 16; g++)
        for (int mt = 0; mt < mt_n; mt++)
            for (int lane = 0; lane < 64; lane++)
                for (int c = 0; c < 4; c++) {
                    const int k = 16 * g + 4 * (lane >> 4) + c, mel = 16 * mt + (lane & 15);
                    frag[(((size_t)g * mt_n + mt) * 64 + lane) * 4 + c] = gf[(size_t)k * nm_pad + mel];
                }
    return frag;
}
// liveness-based arena plan: tensor t is born at step t (tensor 0 = front-end) and dies after
// the last layer that reads it; the embedding tensor and the logits live to the end.
void plan_arena(const bh::Model &m, const std::vector<int> &fused_at, const std::vector<bh_classifier::SeInfo> &se, const std::vector<char> &head_gap, size_t max_batch,
                bool keep, std::vector<size_t> &off, size_t &total) {
    const size_t nt = m.layers.size() + 1;
    std::vector<size_t> last(nt, 0), sz(nt);
    for (size_t t = 0; t < nt; t++) { last[t] = t; sz[t] = align_up(m.tensor_floats[t] * max_batch, 64); }
    for (size_t i = 0; i < m.layers.size(); i++) {
        const auto &L = m.layers[i];
        last[L.in_tensor] = std::max(last[L.in_tensor], i + 1);
        if (L.res_tensor != bh::NO_TENSOR) last[L.res_tensor] = std::max(last[L.res_tensor], i + 1);
    }
    if (!keep)
        for (size_t i = 0; i < fused_at.size(); i++) {
            if (fused_at[i] >= 0 && (size_t)fused_at[i] < se.size() && se[fused_at[i]].iP != 0) {
                // a squeeze-excite block = pass A (reads the block input; writes the depthwise output D and, into the slot of the
                // OP_SCALE output, the per-tile channel sums), the gate launch (sums -> gate tensor) and the gated project GEMM
                // (D, gate, residual -> the block output) at the project layer's step.  Everything those launches touch stays
                // live until that step: the planner gives a tensor its bytes at the step its LAYER would have written it, and the
                // sums are written earlier than that (with pass A) -- the block input must not have been handed on by then.
                const auto &S = se[fused_at[i]];
                const size_t end = (size_t)S.iP + 1;
                last[m.layers[i].in_tensor] = std::max(last[m.layers[i].in_tensor], end);
                last[S.iD + 1] = std::max(last[S.iD + 1], end);
                last[S.iGap + 1] = std::max(last[S.iGap + 1], end);     // (the pooled sums and the hidden layer of the gate launches:
                last[S.iPw1 + 1] = std::max(last[S.iPw1 + 1], end);     //  written while the per-tile sums -- born "later" -- are read)
                last[S.iPw2 + 1] = std::max(last[S.iPw2 + 1], end);
                last[S.iScale + 1] = std::max(last[S.iScale + 1], end);
                if (S.iD != i) sz[i + 1] = 0;                                  // the expanded tensor stays in LDS
                sz[S.iScale + 1] = align_up(S.part_floats * max_batch, 64);  // the scaled tensor never exists
            } else
            if (fused_at[i] >= 0) {
                // one launch reads the block input while it writes the block's last tensor (i + 3; i + 2 for a block without an
                // expand convolution: depthwise -> project); the tensors in between stay in LDS and take no arena space
                const size_t len = (i + 2 < m.layers.size() && m.layers[i].op != bh::OP_DWCONV) ? 3 : 2;
                last[m.layers[i].in_tensor] = std::max(last[m.layers[i].in_tensor], i + len);
                for (size_t k = 1; k < len; k++) sz[i + k] = 0;
            }
        }
    if (!keep)
        for (size_t i = 0; i + 1 < head_gap.size(); i++)
            if (head_gap[i]) {
                // head conv + pool in one launch: workgroups still read the conv's input while finished ones store
                // pooled rows (tensor i+2), so the input lives through step i+2; the conv's output never exists
                last[m.layers[i].in_tensor] = std::max(last[m.layers[i].in_tensor], i + 2);
                sz[i + 1] = 0;
            }
    last[m.h.embedding_tensor] = nt;
    last[nt - 1] = nt;
    off.assign(nt, 0);
    total = 0;
    if (keep) {
        for (size_t t = 0; t < nt; t++) { off[t] = total; total += sz[t]; }
        return;
    }
    struct Live { size_t off, size, last; };
    std::vector<Live> live;
    for (size_t t = 0; t < nt; t++) {
        // tensors whose last reader ran before step t are dead (step t writes tensor t while
        // reading tensors with last >= t)
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live &l) { return l.last < t; }), live.end());
        std::sort(live.begin(), live.end(), [](const Live &a, const Live &b) { return a.off < b.off; });
        size_t pos = 0;
        for (const auto &l : live) {
            if (pos + sz[t] <= l.off) break;
            pos = std::max(pos, l.off + l.size);
        }
        off[t] = pos;
        live.push_back({pos, sz[t], last[t]});
        total = std::max(total, pos + sz[t]);
    }
}
// IEEE binary16 round-to-nearest-even of a float, and back (host side of the hi/lo operand split)
uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));   // overflow / nan
    if (x < 0x38800000u) {                                                                     // subnormal half
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 113 - (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t half = m >> (shift + 13), rem = m & ((1u << (shift + 13)) - 1), mid = 1u << (shift + 12);
        return (uint16_t)(sign | (half + ((rem > mid || (rem == mid && (half & 1))) ? 1 : 0)));
    }
    const uint32_t e = ((x >> 23) - 112) << 10, m = (x >> 13) & 0x3ffu, rem = x & 0x1fffu;
    uint32_t h = e | m;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int k = 0; uint32_t mm = m; while (!(mm & 0x400u)) { mm <<= 1; k++; } x = sign | ((uint32_t)(113 - k) << 23) | ((mm & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}

// The block that starts at layer i as a fused launch: expand(1x1, or the stem conv) -> depthwise -> project(1x1) whose
// intermediates have no other reader, described and handed to the tile planner.  Host logic only (no device): plan_fusion uses
// it per block at create, bh_plan_fused_blocks walks a model file with it.
// The squeeze-excite block that starts at layer i, if there is one: [expand 1x1 | stem conv] -> depthwise -> GlobalAveragePool ->
// 1x1 (C -> Cr) -> 1x1 (Cr -> C) -> OP_SCALE (depthwise output x gate) -> project 1x1, the depthwise output read by the pool and
// the scale only, every other intermediate by its successor only (EfficientNet's MBConv as tf2onnx / torch exporters spell it:
// onnx_conv.hpp, convert.py).
bool match_se_block(const bh::Model &m, const std::vector<int> &readers, size_t i, bh_classifier::SeInfo &S, bool &noexp) {
    const size_t nl = m.layers.size();
    noexp = m.layers[i].op == bh::OP_DWCONV;
    const size_t iD = noexp ? i : i + 1;
    if (iD + 5 >= nl) return false;
    const auto &D = m.layers[iD], &G = m.layers[iD + 1], &A1 = m.layers[iD + 2], &A2 = m.layers[iD + 3], &SC = m.layers[iD + 4], &P = m.layers[iD + 5];
    if (D.op != bh::OP_DWCONV || G.op != bh::OP_GAP || A1.op != bh::OP_PWCONV || A2.op != bh::OP_PWCONV || SC.op != bh::OP_SCALE || P.op != bh::OP_PWCONV) return false;
    const uint32_t tD = (uint32_t)iD + 1;   // the depthwise output tensor
    if (G.in_tensor != tD || A1.in_tensor != iD + 2 || A2.in_tensor != iD + 3 || SC.in_tensor != tD || SC.res_tensor != iD + 4 || P.in_tensor != iD + 5) return false;
    if (readers[tD] != 2 || readers[iD + 2] != 1 || readers[iD + 3] != 1 || readers[iD + 4] != 1 || readers[iD + 5] != 1) return false;
    if (!noexp && (m.layers[i + 1].in_tensor != i + 1 || readers[i + 1] != 1)) return false;
    if (G.res_tensor != bh::NO_TENSOR || A1.res_tensor != bh::NO_TENSOR || A2.res_tensor != bh::NO_TENSOR || D.res_tensor != bh::NO_TENSOR) return false;
    if (A1.in_h * A1.in_w != 1 || A2.in_h * A2.in_w != 1 || A1.cin != D.cout || A2.cout != D.cout || A1.cout != A2.cin || P.cin != D.cout) return false;
    if (!bh::se_gate_supports((int)D.cout, (int)A1.cout)) return false;
    S = bh_classifier::SeInfo{};
    S.iD = (uint32_t)iD; S.iGap = (uint32_t)iD + 1; S.iPw1 = (uint32_t)iD + 2; S.iPw2 = (uint32_t)iD + 3; S.iScale = (uint32_t)iD + 4; S.iP = (uint32_t)iD + 5;
    return true;
}

// The tile planner in the block's precision, then in the next more exact one that has an entry for its shape: a plain-f16 block
// (BH_FLAG_F16) without a plain-f16 instantiation runs on the split-f16 one -- more exact AND faster than the f32 MFMA -- and only
// then, as a split-f16 block without an entry does, on the f32 MFMA.
static bool plan_in_some_precision(bh::MbDesc &d, int force_cfg) {
    const int order[3] = {d.prec, d.prec == 1 ? 3 : 0, 0};
    for (int k = 0; k < 3; k++) {
        if (k > 0 && order[k] == order[k - 1]) continue;
        d.prec = order[k];
        if (bh::mb_plan(d, force_cfg)) return true;
    }
    return false;
}

bool describe_fused_block(const bh::Model &m, const std::vector<int> &readers, size_t i, int precision, int force_cfg, bh::MbDesc &d) {
    const size_t nl = m.layers.size();
    {
        bh_classifier::SeInfo S;
        bool noexp = false;
        static const bool se_off = [] { const char *e = getenv("BIRDA_HIP_FUSE_SE"); return e && e[0] == '0'; }();
        if (!se_off && match_se_block(m, readers, i, S, noexp)) {
            const auto &E = m.layers[i], &D = m.layers[S.iD], &P = m.layers[S.iP];
            const bool stem = !noexp && E.op == bh::OP_CONV && E.in_layout == 1 && E.kh == E.kw && E.sh == E.sw && E.in_tensor == 0;
            if (!noexp && E.op != bh::OP_PWCONV && !stem) return false;
            if (!noexp && E.res_tensor != bh::NO_TENSOR) return false;
            if (D.kh != D.kw || D.sh != D.sw || (!noexp && E.cout != D.cout) || D.in_layout != 0) return false;
            // (the gated project GEMM has no activation on the f16 MFMA; project convolutions of MBConv blocks have none)
            if (P.act != bh::ACT_NONE) return false;
            d = bh::MbDesc{};
            d.se = 1;
            d.dblk = 0;      // (set per forward: api.hip forward_slice)
            d.noexp = noexp ? 1 : 0;
            d.H = (int)(noexp ? D.in_h : E.in_h); d.W = (int)(noexp ? D.in_w : E.in_w);
            d.Cin = (int)(noexp ? D.cout : E.cin); d.Cexp = (int)D.cout; d.Cout = (int)P.cout;
            if (stem) {
                d.stem = 1; d.stem_c = (int)E.cin; d.stem_h = (int)E.in_h; d.stem_w = (int)E.in_w; d.stem_k = (int)E.kh;
                d.stem_s = (int)E.sh; d.stem_pt = (int)E.pad_t; d.stem_pl = (int)E.pad_l;
                d.H = (int)E.out_h; d.W = (int)E.out_w; d.Cin = (int)(E.kh * E.kw * E.cin);
            }
            d.Ho = (int)D.out_h; d.Wo = (int)D.out_w; d.pad_t = (int)D.pad_t; d.pad_l = (int)D.pad_l;
            d.KS = (int)D.kh; d.ST = (int)D.sh;
            d.act_e = (int)(noexp ? D.act : E.act); d.act_d = (int)D.act; d.act_p = (int)P.act;
            d.prec = precision;
            return plan_in_some_precision(d, force_cfg);
        }
    }
    if (i + 1 < nl && m.layers[i].op == bh::OP_DWCONV && m.layers[i + 1].op == bh::OP_PWCONV) {
        // depthwise -> project (+ residual) WITHOUT an expand convolution (the expand-ratio-1 blocks of EfficientNet after the
        // first): fused with the block input standing in for the expanded tensor (MbDesc::noexp)
        const auto &D = m.layers[i], &P = m.layers[i + 1];
        if (P.in_tensor != i + 1 || readers[i + 1] != 1 || D.res_tensor != bh::NO_TENSOR) return false;
        if (D.kh != D.kw || D.sh != D.sw || D.cout != P.cin || D.in_layout != 0) return false;
        d = bh::MbDesc{};
        d.noexp = 1;
        d.H = (int)D.in_h; d.W = (int)D.in_w; d.Cin = (int)D.cout; d.Cexp = (int)D.cout; d.Cout = (int)P.cout;
        d.Ho = (int)D.out_h; d.Wo = (int)D.out_w; d.pad_t = (int)D.pad_t; d.pad_l = (int)D.pad_l;
        d.KS = (int)D.kh; d.ST = (int)D.sh;
        d.act_e = (int)D.act; d.act_d = (int)D.act; d.act_p = (int)P.act;   // (no expand activation: the kernel template is keyed on one)
        if (const char *dbg = BH_XENV("BIRDA_HIP_MB_DBG")) d.dbg = atoi(dbg);
        d.prec = precision;
        return plan_in_some_precision(d, force_cfg);
    }
    if (i + 2 >= nl) return false;
    const auto &E = m.layers[i], &D = m.layers[i + 1], &P = m.layers[i + 2];
    const bool stem = E.op == bh::OP_CONV && E.in_layout == 1 && E.kh == E.kw && E.sh == E.sw && E.in_tensor == 0;
    if ((E.op != bh::OP_PWCONV && !stem) || D.op != bh::OP_DWCONV || P.op != bh::OP_PWCONV) return false;
    if (D.in_tensor != i + 1 || P.in_tensor != i + 2 || readers[i + 1] != 1 || readers[i + 2] != 1) return false;
    if (E.res_tensor != bh::NO_TENSOR || D.res_tensor != bh::NO_TENSOR) return false;
    if (D.kh != D.kw || D.sh != D.sw || E.cout != D.cout || D.cout != P.cin) return false;
    d = bh::MbDesc{};
    d.H = (int)E.in_h; d.W = (int)E.in_w; d.Cin = (int)E.cin; d.Cexp = (int)E.cout; d.Cout = (int)P.cout;
    if (stem) {  // the depthwise conv sees the stem's output image; the stem itself is gathered
        d.stem = 1; d.stem_c = (int)E.cin; d.stem_h = (int)E.in_h; d.stem_w = (int)E.in_w; d.stem_k = (int)E.kh;
        d.stem_s = (int)E.sh; d.stem_pt = (int)E.pad_t; d.stem_pl = (int)E.pad_l;
        d.H = (int)E.out_h; d.W = (int)E.out_w; d.Cin = (int)(E.kh * E.kw * E.cin);
    }
    d.Ho = (int)D.out_h; d.Wo = (int)D.out_w; d.pad_t = (int)D.pad_t; d.pad_l = (int)D.pad_l;
    d.KS = (int)D.kh; d.ST = (int)D.sh;
    d.act_e = (int)E.act; d.act_d = (int)D.act; d.act_p = (int)P.act;
    if (const char *dbg = BH_XENV("BIRDA_HIP_MB_DBG")) d.dbg = atoi(dbg);
    d.prec = precision;
    // The split-f16 MFMA and the f32 MFMA agree to ~1e-7 of sum|a b|; measured (profiles/), f16x3 is
    // the faster one on every block, the stem's 18-column im2col GEMM included.
    if (d.stem && precision == 3 && BH_XENV("BIRDA_HIP_STEM_F32")) d.prec = 0;   // A/B aid
    return plan_in_some_precision(d, force_cfg);
}

std::vector<int> tensor_readers(const bh::Model &m) {
    std::vector<int> readers(m.layers.size() + 1, 0);
    for (const auto &L : m.layers) {
        readers[L.in_tensor]++;
        if (L.res_tensor != bh::NO_TENSOR) readers[L.res_tensor]++;
    }
    readers[m.h.embedding_tensor]++;
    return readers;
}

// Prepares a fused launch for every block describe_fused_block accepts (weights re-laid fragment-major for the picked tile config).
int plan_fusion(bh_classifier *c) {
    const auto &m = c->model;
    const size_t nl = m.layers.size();
    c->fused_at.assign(nl, -1);
    const char *fuse_env = getenv("BIRDA_HIP_FUSE");
    if (fuse_env && fuse_env[0] == '0') return BH_OK;
    const char *cfg_env = getenv("BIRDA_HIP_MB_CFG");
    const int force_cfg = cfg_env ? atoi(cfg_env) : -1;
    const std::vector<int> readers = tensor_readers(m);
    for (size_t i = 0; i + 2 < nl; i++) {
        bh::MbDesc d{};
        if (!describe_fused_block(m, readers, i, c->precision, force_cfg, d)) continue;
        // (a no-expand block is layers i = depthwise, i + 1 = project; E then only lends the code below a valid layer to name)
        bh_classifier::SeInfo S;
        if (d.se) { bool ne = false; if (!match_se_block(m, readers, i, S, ne)) continue; }
        const auto &E = m.layers[i], &D = m.layers[d.noexp ? i : i + 1], &P = m.layers[d.se ? S.iP : d.noexp ? i + 1 : i + 2];
        // per-chunk weight blocks (kernels.hpp MbDesc)
        const int CE = d.CE, NTE = CE / 16, KG = d.KG, NTOP = d.NTOP, nch = d.nchunks, KK = d.KS * d.KS;
        const float *We = m.blob.data() + E.w_off, *Wp = m.blob.data() + P.w_off, *Wd = m.blob.data() + D.w_off;
        const float *be = m.blob.data() + E.b_off, *bd = m.blob.data() + D.b_off;
        const bool h16 = d.prec != 0;
        // f16 operand planes hold We * 2^se and Wp * 2^sp (kernels.hpp f16_scale_exponent); be / bp are multiplied alike
        int se = 0, sp = 0;
        if (h16) {
            float me = 0.0f, mp = 0.0f;
            for (size_t q = 0; !d.noexp && q < (size_t)d.Cin * d.Cexp; q++) me = std::max(me, std::fabs(We[q]));
            for (size_t q = 0; q < (size_t)d.Cexp * d.Cout; q++) mp = std::max(mp, std::fabs(Wp[q]));
            se = bh::f16_scale_exponent(me);
            sp = bh::f16_scale_exponent(mp);
        }
        // GELU blocks of the f16 modes: the expand GELU runs on the SCALED accumulator with coefficients c_k 2^(-k se) and the 2^-se
        // moves into the depthwise taps (kernels.hpp gelu_erf_fast4_scaled).  c_5 2^(-5 se) must stay a normal f32 on both sides:
        // |se| <= 21 (weights 2^7 away from the usual He-normal sizes still land within 2^-8 of the top of the f16 range).
        d.e_fold = 0;
        d.gelu = bh::GeluScaled{0.f, 0.f, 0.f, 0.f, 0.f};
#if BH_GELU_DEGREE == 5
        if (d.noexp) se = 0;
        if (h16 && d.act_e == bh::ACT_GELU_ERF && !d.noexp) {
            se = std::max(-21, std::min(21, se));
            d.e_fold = 1;
            float gc[5];
            for (int k = 1; k <= 5; k++) gc[k - 1] = std::ldexp(bh::kGeluCoef[k - 1], -k * se);
            d.gelu = bh::GeluScaled{gc[0], gc[1], gc[2], gc[3], gc[4]};
        }
#endif
        // ... and both GELUs of such a block leave TWICE their value (gelu2x_fast4, kernels.hpp): the expand one's factor joins the
        // 2^-se in the depthwise taps (x2e), the depthwise one's raises the exponent the project accumulators live at (x2d).
        int x2e = 0, x2d = 0;
#if BH_GELU_DEGREE == 5 && BH_GELU_2X
        if (h16 && d.act_d == bh::ACT_GELU_ERF) x2d = 1;
        x2e = d.e_fold;
#endif
        const int spa = sp + x2d;   // the project accumulators hold 2^spa times the output
        d.e_unscale = std::ldexp(1.0f, -se); d.p_scale = std::ldexp(1.0f, spa); d.p_unscale = std::ldexp(1.0f, -spa);
        const size_t frag = h16 ? 512 : 256, psteps = h16 ? (CE + 31) / 32 : NTE;
        const bool p16 = h16 && CE == 16;   // project GEMM as one 16-deep step: [column tile]{hi, lo}[64 lanes][4 halves]
        const size_t we_fl = (size_t)KG * NTE * frag + CE, wp_fl = p16 ? (size_t)NTOP * 256 : psteps * NTOP * frag, wd_fl = (size_t)KK * CE + CE;
        std::vector<float> wef(nch * we_fl, 0.0f), wpf(nch * wp_fl, 0.0f), wdf(nch * wd_fl, 0.0f);
        // (stem block in the f16 modes: the kernel gathers its im2col columns by memory runs, kernels_mbconv.hip -- position
        //  8 kq + 3 q + dx of the one 32-deep step is tap (dy, dx) of channel ch with run 2 kq + q = 3 ch + dy)
        auto we_at = [&](int k, int n) {
            if (d.stem && h16) {
                const int g = k >> 5, kq = (k & 31) >> 3, jj = k & 7, r = 8 * g + 2 * kq + jj / 3, dx = jj % 3;
                if (jj >= 6 || r >= 3 * d.stem_c) return 0.0f;
                const int ch = r / 3, dy = r - 3 * ch;
                k = (dy * 3 + dx) * d.stem_c + ch;
            }
            return (!d.noexp && k < d.Cin && n < d.Cexp) ? std::ldexp(We[(size_t)k * d.Cexp + n], se) : 0.0f;
        };
        auto wp_at = [&](int k, int n) { return (k < d.Cexp && n < d.Cout) ? std::ldexp(Wp[(size_t)k * d.Cout + n], sp) : 0.0f; };
        // f16: element jj of lane's 8-half fragment = k = 32 g + 8 (lane >> 4) + jj; hi plane then lo plane
        auto put16 = [&](std::vector<float> &dst, size_t base_fl, int plane, int lane, int jj, float v) {
            uint16_t *h = reinterpret_cast<uint16_t *>(dst.data() + base_fl) + ((size_t)plane * 64 + lane) * 8 + jj;
            const uint16_t hi = f32_to_f16(v);
            *h = plane == 0 ? hi : f32_to_f16(v - f16_to_f32(hi));
        };
        for (int ch = 0; ch < nch; ch++) {
            for (int g = 0; g < KG; g++)
                for (int j = 0; j < NTE; j++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int n = ch * CE + 16 * j + (lane & 15);
                        if (h16) {
                            for (int jj = 0; jj < 8; jj++) {
                                const float v = we_at(32 * g + 8 * (lane >> 4) + jj, n);
                                const size_t base = ch * we_fl + ((size_t)g * NTE + j) * 512;
                                put16(wef, base, 0, lane, jj, v);
                                put16(wef, base, 1, lane, jj, v);
                            }
                        } else {
                            for (int cc = 0; cc < 4; cc++)
                                wef[ch * we_fl + (((size_t)g * NTE + j) * 64 + lane) * 4 + cc] = we_at(16 * g + 4 * (lane >> 4) + cc, n);
                        }
                    }
            for (int n = 0; n < CE; n++) wef[ch * we_fl + (size_t)KG * NTE * frag + n] = (!d.noexp && ch * CE + n < d.Cexp) ? std::ldexp(be[ch * CE + n], se) : 0.0f;
            if (p16) {
                for (int j = 0; j < NTOP; j++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int jj = 0; jj < 4; jj++) {
                            const float v = wp_at(ch * CE + 4 * (lane >> 4) + jj, 16 * j + (lane & 15));
                            uint16_t *h = reinterpret_cast<uint16_t *>(wpf.data() + ch * wp_fl + (size_t)j * 256);
                            const uint16_t hi = f32_to_f16(v);
                            h[(size_t)lane * 4 + jj] = hi;
                            h[(size_t)(64 + lane) * 4 + jj] = f32_to_f16(v - f16_to_f32(hi));
                        }
            }
            for (int g = 0; g < (int)psteps && !p16; g++)
                for (int j = 0; j < NTOP; j++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int n = 16 * j + (lane & 15);
                        if (h16) {
                            for (int jj = 0; jj < 8; jj++) {
                                const int kk = 32 * g + 8 * (lane >> 4) + jj;          // k inside the chunk (zero padding past CE)
                                const float v = kk < CE ? wp_at(ch * CE + kk, n) : 0.0f;
                                const size_t base = ch * wp_fl + ((size_t)g * NTOP + j) * 512;
                                put16(wpf, base, 0, lane, jj, v);
                                put16(wpf, base, 1, lane, jj, v);
                            }
                        } else {
                            for (int cc = 0; cc < 4; cc++)
                                wpf[ch * wp_fl + (((size_t)g * NTOP + j) * 64 + lane) * 4 + cc] = wp_at(ch * CE + 16 * g + 4 * (lane >> 4) + cc, n);
                        }
                    }
            for (int tap = 0; tap < KK; tap++)
                for (int n = 0; n < CE; n++)
                    wdf[ch * wd_fl + (size_t)tap * CE + n] = ch * CE + n < d.Cexp ? std::ldexp(Wd[(size_t)tap * d.Cexp + ch * CE + n], d.e_fold ? -se - x2e : 0) : 0.0f;
            for (int n = 0; n < CE; n++) wdf[ch * wd_fl + (size_t)KK * CE + n] = ch * CE + n < d.Cexp ? bd[ch * CE + n] : 0.0f;
        }
        float *dwe = nullptr, *dwp = nullptr, *dwd = nullptr;
        int rc = upload(wef.data(), wef.size() * sizeof(float), &dwe);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(dwe);
        rc = upload(wpf.data(), wpf.size() * sizeof(float), &dwp);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(dwp);
        rc = upload(wdf.data(), wdf.size() * sizeof(float), &dwd);
        if (rc != BH_OK) return rc;
        c->d_owned.push_back(dwd);
        d.We = dwe; d.Wp = dwp; d.Wd = dwd;
        d.bp = c->d_blob + P.b_off;
        if (spa != 0) {   // bp * 2^spa: the project accumulators start there
            std::vector<float> bps(d.Cout);
            for (int n = 0; n < d.Cout; n++) bps[n] = std::ldexp(m.blob[P.b_off + n], spa);
            float *dbp = nullptr;
            rc = upload(bps.data(), bps.size() * sizeof(float), &dbp);
            if (rc != BH_OK) return rc;
            c->d_owned.push_back(dbp);
            d.bp = dbp;
        }
        c->fused_at[i] = (int)c->mb.size();
        c->mb.push_back(d);
        {
            bh::MbDesc tw{};
            tw.cfg = -1;
            static const bool no_twin = BH_XENV("BIRDA_HIP_MB_TWIN") && BH_XENV("BIRDA_HIP_MB_TWIN")[0] == '0';   // (A/B aid)
            // (squeeze-excite blocks keep ONE tiling -- the pooled sums are added tile by tile, and a segment's logits must not depend
            //  on the size of the launch it ran in -- except where the one-segment twin adds them in the very same order:
            //  mb_twin_sums_match, the 4x16 stages' whole-image tiles)
            if (force_cfg < 0 && !no_twin && bh::mb_plan_twin(d, tw) && (!d.se || bh::mb_twin_sums_match(d, tw))) c->mb_small.push_back(tw);
            else { tw.cfg = -1; c->mb_small.push_back(tw); }
            bh::MbDesc nw{};
            nw.cfg = -1;
            if (!(force_cfg < 0 && !no_twin && !d.se && bh::mb_plan_narrow(d, nw))) nw.cfg = -1;
            c->mb_narrow.push_back(nw);
            // (squeeze-excite: the slot of the per-tile channel sums holds whichever of the three tilings has the most tiles)
            size_t tiles = (size_t)d.tiles_x * d.tiles_y;
            if (c->mb_small.back().cfg >= 0) tiles = std::max(tiles, (size_t)c->mb_small.back().tiles_x * c->mb_small.back().tiles_y);
            if (nw.cfg >= 0) tiles = std::max(tiles, (size_t)nw.tiles_x * nw.tiles_y);
            S.part_floats = d.se ? tiles * (size_t)d.Cexp : 0;
            c->se.push_back(d.se ? S : bh_classifier::SeInfo{});
        }
        if (d.se) { i = S.iP; continue; }
        i += d.noexp ? 1 : 2;
    }
    return BH_OK;
}

}  // namespace bhi
