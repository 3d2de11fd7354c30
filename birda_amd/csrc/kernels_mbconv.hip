// Planner of the fused inverted-residual (MBConv) blocks: picks, per block, one of the tile configurations of mbconv_cfgs.inc
// (the kernel itself is mbconv_kernel.hpp, instantiated per activation in kernels_mbconv_gelu.hip / _swish.hip / _relu6.hip).
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>

#include "kernels.hpp"
#include "mbconv_cfg.hpp"

namespace bh {

namespace {

// The list (mbconv_cfgs.inc: 115 tile configurations, indices as documented there) is instantiated once per activation:
// entry ci + k * kNBase is configuration ci with the k-th activation of kActs.
constexpr int kActs[] = {ACT_GELU_ERF, ACT_SWISH, ACT_RELU6};
constexpr int kNActs = (int)(sizeof(kActs) / sizeof(kActs[0]));
struct CfgTables {
    std::vector<MbCfg> t[kNActs];     // (each activation's list, put together from its three translation units)
    int n_base;
    CfgTables() {
        typedef const MbCfg *(*Part)(int *);
        static const Part parts[kNActs][3] = {{mb_table_gelu_p0, mb_table_gelu_p1, mb_table_gelu_p2},
                                              {mb_table_swish_p0, mb_table_swish_p1, mb_table_swish_p2},
                                              {mb_table_relu6_p0, mb_table_relu6_p1, mb_table_relu6_p2}};
        for (int a = 0; a < kNActs; a++)
            for (int k = 0; k < 3; k++) {
                int n = 0;
                const MbCfg *p = parts[a][k](&n);
                t[a].insert(t[a].end(), p, p + n);
            }
        n_base = (t[0].size() == t[1].size() && t[1].size() == t[2].size()) ? (int)t[0].size() : 0;   // the three copies are the same list
    }
    const MbCfg &operator[](int ci) const { return t[ci / n_base][ci % n_base]; }
};
const CfgTables &cfg_tables() { static const CfgTables T; return T; }
#define kCfgs cfg_tables()
#define kNBase (cfg_tables().n_base)
#define kNCfgs (kNBase * kNActs)

// fills the derived fields for entry `ci` with tile height `th`; returns the estimated MFMA work
// per segment (in 16x16x4 steps), or -1 when the entry cannot run this block at that height
// (why an entry was refused, counted per reason while BIRDA_HIP_MB_WHY is set: tools/plan_coverage.py --why)
thread_local int g_why[16];
#define MB_NO(r) do { g_why[r]++; return -1; } while (0)
double mb_try_th(MbDesc &d, int ci, int th, bool relax_kg = false) {
    const MbCfg &c = kCfgs[ci];
    // any Cexp: the last chunk's missing channels are zero weights and biases (every activation here maps 0 to 0)
    if (c.KS != d.KS || c.ST != d.ST || (!d.stem && d.Cin % 4) || d.Cexp % 4) return -1;
    if (c.PREC != d.prec) return -1;
    if (d.act_e != c.ACT || d.act_d != c.ACT || (!d.se && d.act_p != ACT_NONE)) MB_NO(1);   // (se: the project conv is another launch)
    if (c.STEM != (d.stem ? d.stem_c : 0)) MB_NO(2);
    // (the kernel's per-lane offsets are 24-bit products and 32-bit element offsets within one workgroup's segments)
    if ((long)c.S * d.H * d.W * std::max(d.Cin, 1) >= (1L << 24) || (long)c.S * d.Ho * d.Wo * d.Cout >= (1L << 24) ||
        (d.stem && (long)d.stem_c * d.stem_h * d.stem_w >= (1L << 24))) return -1;
    if (d.stem && d.stem_k != 3) MB_NO(3);
    if (d.se && (!c.launch_se || c.PERSIST)) MB_NO(4);   // pass A of a squeeze-excite block: only where it is instantiated (mbconv_kernel.hpp MB_WITH_SE)
    if (c.KG == 0) { if (!d.noexp || d.stem) return -1; }   // the no-expand entries serve the no-expand blocks, and only them
    else {
        // k steps of the expand GEMM: the entry's own number, or (relax_kg: a block no entry was shaped for) any number up to it --
        // the columns past Cin are zero weights (plan_fusion) against zeroed operands (the kernel masks its loads at Cin)
        // (stem block in the f16 modes: the im2col columns are packed by memory runs, 8 runs of three taps to a step)
        const int need = (d.stem && c.PREC) ? (3 * d.stem_c + 7) / 8 : (d.Cin + (c.PREC ? 31 : 15)) / (c.PREC ? 32 : 16);
        if (d.noexp) return -1;
        if (need != c.KG && !(relax_kg && need < c.KG && !d.stem)) MB_NO(5);
    }
    if (c.COLTH) {   // column tasks: the tile is the whole image, COLTH rows high, symmetric padding, one task per thread
        // (relaxed: any image up to COLTH rows high -- the rows below it are grid rows that stay zero, exactly the padding a column
        //  task skips at compile time above and below the image; their outputs are computed and never stored)
        if (th != c.COLTH || d.Ho > c.COLTH || d.H > c.COLTH || (!relax_kg && d.H != c.COLTH) || d.ST != 1 || d.pad_t != (c.KS - 1) / 2) MB_NO(6);
        if (c.S * (1 << c.TWL) * (c.CE / (c.WM * c.WN == 8 ? 2 : 4)) > 64 * c.WM * c.WN) MB_NO(6);   // one task per thread
    }
    const int nto = (d.Cout + 15) / 16;
    if (nto > c.WN * c.NT_W) MB_NO(7);
    const int TW = 1 << c.TWL;
    const int ces = c.CE + 4, pout_pad = c.WM * c.MT_W * 16;
    if (th < 1 || c.S * th * TW > pout_pad) MB_NO(8);
    MbDesc t = d;
    t.cfg = ci; t.TH = th; t.S = c.S;
    t.tiles_y = (d.Ho + th - 1) / th; t.tiles_x = (d.Wo + TW - 1) / TW;
    {
        auto magic = [](unsigned dv) { const unsigned long long q = (1ull << 32) / dv + 1; return (unsigned)std::min<unsigned long long>(q, 0xffffffffull); };
        t.rcp_tiles_x = magic((unsigned)t.tiles_x); t.rcp_tiles_xy = magic((unsigned)(t.tiles_x * t.tiles_y));
    }
    t.IH = (th - 1) * c.ST + c.KS; t.IW = (TW - 1) * c.ST + c.KS;
    t.KG = c.KG; t.nchunks = (d.Cexp + c.CE - 1) / c.CE; t.NTOP = c.WN * c.NT_W; t.CE = c.CE;
    const int mseg = std::min(t.IH, d.H) * std::min(t.IW, d.W);
    t.mpad_max = (c.S * mseg + 15) / 16 * 16;
    if (t.mpad_max / 16 > c.RT_W * (c.WM * c.WN / c.NCS)) MB_NO(9);  // a wave keeps all its rows of X in registers
    const size_t frag = c.PREC ? 512 : 256, psteps = c.PREC ? (c.CE + 31) / 32 : c.CE / 16;
    const bool p16 = c.PREC && c.CE == 16;   // one 16-deep project step, half-size fragments and D rows
    const size_t we_fl = (size_t)c.KG * (c.CE / 16) * frag + c.CE, wp_fl = p16 ? (size_t)t.NTOP * 256 : psteps * t.NTOP * frag;
    const size_t wd_fl = (size_t)c.KS * c.KS * c.CE + c.CE;
    const size_t ds_fl = c.PREC ? (size_t)pout_pad * (p16 ? 24 : psteps * 32 + 8) : (size_t)pout_pad * ces;
    if (c.PERSIST == 2) {   // strip-walking workgroups: more than one tile row to walk, one segment, a halo the step inherits
        if (t.tiles_y < 2 || c.S != 1 || c.KS <= c.ST || th * c.ST < c.KS - c.ST || d.pad_t > th * c.ST) return -1;
    }
    const size_t halo_bytes = c.PERSIST == 2 ? (size_t)t.nchunks * (c.KS - c.ST) * t.IW * ces * 4 : 0;   // [chunk][KS - ST rows][IW][ces]
    const size_t lds_base = (((size_t)c.S * t.IH * t.IW + 1) * ces + ds_fl) * 4 + (size_t)pout_pad * 4 + halo_bytes;
    t.lds_bytes = lds_base + (we_fl + wp_fl + wd_fl) * 4 * (c.PERSIST == 1 ? (size_t)t.nchunks : 1);   // persistent: every chunk resident
    if (t.lds_bytes > 160 * 1024) MB_NO(10);
    if (c.PERSIST == 1 && t.lds_bytes > 80 * 1024) return -1;   // one workgroup per CU cannot hide its own set-up
    // weight ring (We x 2, Wp x 3, Wd x 2, a whole chunk of prefetch distance): for 16-channel chunks, when the workgroups the
    // entry's register budget allows per CU still fit in LDS with it.  BIRDA_HIP_MB_RING=0/1 forces it off / on where it fits.
    t.ring = 0;
    {
        const size_t lds_ring = lds_base + (2 * we_fl + 3 * wp_fl + 2 * wd_fl) * 4;
        const char *re = BH_XENV("BIRDA_HIP_MB_RING");
        // Measured (profiles/r2 notes): the ring removes the wait in front of B1 (22 % -> 3 % of a wave's
        // cycles) but the time moves to B2 and the launch does not get shorter, and wherever it costs a workgroup per CU it is
        // slower (7.31 -> 8.02 us per segment over all blocks when forced).  Off unless BIRDA_HIP_MB_RING=1.
        const bool want = !c.PERSIST && re && re[0] == '1' && c.CE == 16 && t.nchunks > 2 && lds_ring <= 160 * 1024;
        if (want) { t.ring = 1; t.lds_bytes = lds_ring; }
    }
    d = t;
    const double tiles = (double)t.tiles_y * t.tiles_x / c.S;
    // (relaxed pass: the weights every workgroup streams count too -- 1 / 192 of an MFMA step per byte: 64 bytes a clock into a CU's
    //  LDS against four SIMDs each a split-f16 step in 48.  Whole-image tiles of a late block then beat narrow ones that stream
    //  its 3-6 MB of weights two to four times an image.)
    if (relax_kg)
        return tiles * ((double)t.mpad_max / 16 * t.KG * 4 * (d.Cexp / 16) + (double)pout_pad / 16 * t.NTOP * (d.Cexp / 4) +
                        (double)t.nchunks * (double)(we_fl + wp_fl + wd_fl) * 4.0 / 192.0);
    if (c.PERSIST == 2)   // (the halo rows are not expanded again: TH * ST new rows per step)
        return tiles * ((double)(c.S * std::min(th * c.ST, d.H) * std::min(t.IW, d.W) + 15) / 16 * t.KG * 4 * (d.Cexp / 16) + (double)pout_pad / 16 * t.NTOP * (d.Cexp / 4));
    return tiles * ((double)t.mpad_max / 16 * t.KG * 4 * (d.Cexp / 16) + (double)pout_pad / 16 * t.NTOP * (d.Cexp / 4));
}

// Index of configuration `base` (0 .. kNBase - 1: the documented indices) instantiated for this block's activation, -1 when the
// block's expand / depthwise activation has no instantiation.
int mb_act_index(const MbDesc &d, int base) {
    if (base < 0 || base >= kNBase) return -1;
    for (int k = 0; k < kNActs; k++)
        if (kActs[k] == d.act_e) return base + k * kNBase;
    return -1;
}

// the entry's own tile height if it fits this block's image, else the tallest one that does
double mb_try(MbDesc &d, int ci, bool relax_kg = false) {
    if (kCfgs[ci].COLTH) {   // column tasks: the entry's height or nothing (a lower image leaves the rows below it unused)
        MbDesc t = d;
        const double w = mb_try_th(t, ci, kCfgs[ci].COLTH, relax_kg);
        if (w >= 0) d = t;
        return w;
    }
    for (int th = std::min(kCfgs[ci].TH, std::max(d.Ho, 1)); th >= 1; th--) {
        MbDesc t = d;
        const double w = mb_try_th(t, ci, th, relax_kg);
        if (w >= 0) { d = t; return w; }
    }
    return -1;
}

}  // namespace

int mb_config_count() { return kNCfgs; }

// the instantiation's template arguments as rocprofv3 prints them ("mbconv_kernel<...>")
int mb_config_name(int ci, char *out, size_t cap) {
    if (ci < 0 || ci >= kNCfgs) return 0;
    const MbCfg &c = kCfgs[ci];
    // the first 19 template arguments, as a profiler prints them (the last three: persistent instantiation, activation, column tasks);
    // the twentieth, SE, belongs to the launch, not the table entry: the caller appends it (classifier.py fused_kernel_name)
    return snprintf(out, cap, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", c.KS, c.ST, c.CE, c.KG, c.RT_W, c.NCS, c.WM,
                    c.WN, c.MT_W, c.NT_W, c.TWL, c.XBL, c.S, c.OCC, c.STEM, c.PREC, c.PERSIST, c.ACT, c.COLTH);
}

bool mb_plan(MbDesc &d, int force_cfg) {
    d.cfg = -1;
    if (force_cfg >= 0) {   // (a base index: the block's own activation selects the copy)
        const int ci = mb_act_index(d, force_cfg);
        MbDesc t = d;
        if (ci < 0) return false;
        if (mb_try(t, ci) < 0) { t = d; if (mb_try(t, ci, true) < 0) return false; }
        d = t;
        return true;
    }
    if (const char *pref = getenv("BIRDA_HIP_MB_PREFER")) {  // tuning aid: first valid entry of a comma list
        for (const char *q = pref; *q;) {
            const int ci = mb_act_index(d, atoi(q));
            MbDesc t = d;
            if (ci >= 0 && mb_try(t, ci) >= 0) { d = t; return true; }
            t = d;
            if (ci >= 0 && mb_try(t, ci, true) >= 0) { d = t; return true; }
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
    }
    // measured on MI355X (profiles/): 16-channel chunks (2-4 workgroups per CU) win wherever an
    // instantiation exists; the 192-channel 3x16 blocks need two column-split waves and stay at 32
    static const int kPreferred[] = {11, 12, 13, 14, 15, 16, 17, 18, 19, 9, 10, 20, 21};
    // split-f16 with 16-channel chunks (two or more workgroups per CU): the early blocks, and the 6x32
    // blocks whose depthwise phase is light enough (3x3, and the stride-2 5x5)
    // 79..84: persistent twins of 48, 49, 65, 66, 50, 51.  Measured (us per 1000 segments, same box): the two 5x5 blocks gain
    // (741 -> 686, 576 -> 536); the stem, 16 -> 96 -> 24 and 24 -> 144 -> 24 blocks LOSE (928 -> 1161, 1032 -> 1504, 737 -> 953):
    // with every chunk's weights resident they fit two workgroups per CU instead of four, and without a prefetch of the next
    // tile's rows nothing hides a tile's set-up.  Off by default; BIRDA_HIP_MB_PERSIST=1 all six, =2 the 5x5 pair.
    static const int kPreferred16a[] = {79, 80, 81, 82, 83, 84, 48, 49, 50, 51, 52, 113, 55, 58, 65, 66, 99, 101, 103, 105, 107, 109, 111, 85, 87, 89, 91, 95, 97};
    static const int kPreferred16p[] = {83, 84, 48, 49, 50, 51, 52, 113, 55, 58, 65, 66, 99, 101, 103, 105, 107, 109, 111, 85, 87, 89, 91, 95, 97};
    static const int kPreferred16n[] = {48, 49, 50, 51, 52, 113, 55, 58, 65, 66, 99, 101, 103, 105, 107, 109, 111, 85, 87, 89, 91, 95, 97};
    // 99..110: the 8-wave twins of 85..92 and 95..98 (two waves per SIMD inside the one workgroup a CU holds): 307 -> 257,
    // 426 -> 391, 258 -> 224, 292 -> 254 us per 1 000 segments (old and new library in one run, tools/ab_lib.sh); Perch-shaped
    // model 410 -> 256, 407 -> 298
    static const int kPreferred1[] = {100, 102, 104, 106, 108, 110, 112, 114, 86, 88, 90, 92, 96, 98};   // plain f16: the column-task twins where they apply, else the work rule
    const char *pe = BH_XENV("BIRDA_HIP_MB_PERSIST");
    const int persist_mode = !pe ? 0 : pe[0] == '1' ? 2 : pe[0] == '2' ? 1 : 0;
    if (d.prec == 0)
        for (int base : kPreferred) {  // at the entry's own tile height: the shapes it was measured on
            const int ci = mb_act_index(d, base);
            MbDesc t = d;
            if (ci >= 0 && mb_try_th(t, ci, kCfgs[ci].TH) >= 0) { d = t; return true; }
        }
    if (d.prec == 1)
        for (int base : kPreferred1) {
            const int ci = mb_act_index(d, base);
            MbDesc t = d;
            if (ci >= 0 && mb_try_th(t, ci, kCfgs[ci].TH) >= 0) { d = t; return true; }
        }
    if (d.prec == 3) {
        const int *list = persist_mode == 2 ? kPreferred16a : persist_mode == 1 ? kPreferred16p : kPreferred16n;
        const int nlist = persist_mode == 2 ? (int)(sizeof kPreferred16a / sizeof(int))
                        : persist_mode == 1 ? (int)(sizeof kPreferred16p / sizeof(int)) : (int)(sizeof kPreferred16n / sizeof(int));
        for (int q = 0; q < nlist; q++) {
            const int ci = mb_act_index(d, list[q]);
            MbDesc t = d;
            if (ci >= 0 && mb_try_th(t, ci, kCfgs[ci].TH) >= 0) { d = t; return true; }
        }
    }
    // ... then any entry that can run the block, the one with the least MFMA work: first among the entries whose k steps are the
    // block's own (the shapes of this repo's plans land here or above, unchanged since round 4), then -- a stack nobody tiled by
    // hand -- among those with MORE k steps than the block needs, the surplus zero-padded
    static const bool why = getenv("BIRDA_HIP_MB_WHY") != nullptr;
    for (int relax = 0; relax < 2; relax++) {
        double best = -1;
        MbDesc bestd = d;
        if (why) std::fill(g_why, g_why + 16, 0);
        for (int ci = 0; ci < kNCfgs; ci++) {
            if (kCfgs[ci].PERSIST) continue;   // persistent entries only through the preferred list (measured shapes)
            // (the generic entries, mbconv_cfgs.inc from 211 on, belong to the relaxed pass whatever their k steps: there a hand-shaped
            //  entry that fits the block by relaxation competes with them on cost, weights streamed included)
            if (!relax && ci % kNBase >= 211) continue;
            MbDesc t = d;
            const double w = mb_try(t, ci, relax != 0);
            if (why && getenv("BIRDA_HIP_MB_WHY")[0] == '2' && kCfgs[ci].ACT == d.act_e && kCfgs[ci].KS == d.KS && kCfgs[ci].ST == d.ST && kCfgs[ci].PREC == d.prec)
                fprintf(stderr, "  relax %d entry %d (KG %d COLTH %d): %g\n", relax, ci % kNBase, kCfgs[ci].KG, kCfgs[ci].COLTH, w);
            if (w < 0) continue;
            if (best < 0 || w < best) { best = w; bestd = t; }
        }
        if (best >= 0) {
            // f32 MFMA, relaxed pass: a block whose best entry issues more than 2.5 times the MFMA steps the block itself holds (k steps
            // and project tiles padded far beyond its widths, a halo several times the tile) stays layer by layer -- on the f32 MFMA the
            // padded steps are the kernel's time, and the GEMMs of the layer path do not pad (measured, profiles/r6_c_plan_coverage_f32.txt:
            // 72 -> 216 -> 72 5x5 at 32x69 on entry 6 0.44 of the layer path's speed at 2.9 times the steps, 96 -> 576 -> 160 stride 2 0.52
            // at 3.1; everything below 2.5 within 0.8-1.5).  Split-f16 entries never lost to the layer path (r6_c_plan_coverage_f16x3.txt).
            if (relax && d.prec == 0) {
                const MbCfg &c = kCfgs[bestd.cfg];
                const double tiles = (double)bestd.tiles_y * bestd.tiles_x / c.S, pout_pad = c.WM * c.MT_W * 16;
                const double steps = tiles * ((double)bestd.mpad_max / 16 * bestd.KG * 4 * (d.Cexp / 16.0) + pout_pad / 16 * bestd.NTOP * (d.Cexp / 4.0));
                const double own = (double)d.H * d.W / 16 * ((d.noexp ? 0 : d.Cin) / 4.0) * (d.Cexp / 16.0) + (double)d.Ho * d.Wo / 16 * (d.Cout / 16.0) * (d.Cexp / 4.0);
                static const double pad_max = [] { const char *e = BH_XENV("BIRDA_HIP_MB_F32_PAD"); return e ? atof(e) : 2.5; }();   // (A/B aid)
                if (steps > pad_max * own && d.Cin >= 64 && d.Cexp >= 192) {   // (narrow blocks pad as much on the layer path's GEMM tiles)
                    if (why)
                        fprintf(stderr, "mb_plan: %d -> %d -> %d k%d s%d %dx%d left to the layer kernels on the f32 MFMA: best entry %d issues %.1f x the block's own MFMA steps\n",
                                d.Cin, d.Cexp, d.Cout, d.KS, d.ST, d.H, d.W, bestd.cfg % kNBase, steps / own);
                    return false;
                }
            }
            d = bestd;
            return true;
        }
    }
    if (why)
        fprintf(stderr, "mb_plan: no entry for %s%d -> %d -> %d k%d s%d %dx%d -> %dx%d act %d prec %d se %d: refused by act %d stem %d stemk %d se %d kg %d colth %d nto %d pout %d rows %d lds %d\n",
                d.stem ? "stem " : d.noexp ? "noexp " : "", d.Cin, d.Cexp, d.Cout, d.KS, d.ST, d.H, d.W, d.Ho, d.Wo, d.act_e, d.prec, d.se,
                g_why[1], g_why[2], g_why[3], g_why[4], g_why[5], g_why[6], g_why[7], g_why[8], g_why[9], g_why[10]);
    return false;
}

// The small-batch twin of a planned block, if its configuration has one: the same chunk size, k steps, precision, activation and
// project-tile count (so the weights plan_fusion laid out for `d` are the twin's too) with ONE segment per workgroup instead of two.
bool mb_plan_twin(const MbDesc &d, MbDesc &twin) {
    if (d.cfg < 0 || d.cfg >= kNCfgs) return false;
    const MbCfg &c = kCfgs[d.cfg];
    if (c.S != 2 || !c.COLTH || c.WM * c.WN != 8) return false;
    const int k0 = (d.cfg / kNBase) * kNBase;   // the same activation's copy of the list
    for (int b = 0; b < kNBase; b++) {
        const MbCfg &q = kCfgs[k0 + b];
        if (q.S != 1 || q.COLTH != c.COLTH || q.KS != c.KS || q.ST != c.ST || q.CE != c.CE || q.KG != c.KG || q.PREC != c.PREC ||
            q.TWL != c.TWL || q.WM * q.WN != 8 || q.WN * q.NT_W != c.WN * c.NT_W || q.STEM != c.STEM || q.PERSIST) continue;
        MbDesc t = d;
        // (relaxed: the block may have come to its entry through the relaxed pass -- fewer k steps than the entry's, a lower image)
        if (mb_try_th(t, k0 + b, q.TH, true) >= 0 && t.NTOP == d.NTOP && t.nchunks == d.nchunks && t.KG == d.KG && t.CE == d.CE) { twin = t; return true; }
    }
    return false;
}

// Pass A of a squeeze-excite block on its one-segment twin: the pooled sums must come out bit for bit as from the two-segment tile
// (a segment's logits do not depend on the launch it ran in).  The store phase gives thread tid the pixels tid / C4N + k NTH / C4N
// of the workgroup's segments and adds up per thread, per row of 16 lanes, then over (wave, row) in order: when a segment's pixels
// (whole image = one tile) are a multiple of NTH / C4N, segment slot 1 of the two-segment tile maps its pixels to threads exactly
// as slot 0 does, and as the twin's only slot does -- the same additions in the same order.
bool mb_twin_sums_match(const MbDesc &d, const MbDesc &tw) {
    if (d.cfg < 0 || tw.cfg < 0 || d.cfg >= kNCfgs || tw.cfg >= kNCfgs) return false;
    const MbCfg &c = kCfgs[d.cfg], &q = kCfgs[tw.cfg];
    if (!q.launch_se || d.tiles_x * d.tiles_y != 1 || tw.tiles_x * tw.tiles_y != 1) return false;
    if (c.CE != q.CE || c.WM * c.WN != q.WM * q.WN || c.TWL != q.TWL || d.TH != tw.TH || q.S != 1) return false;
    const int nth = 64 * c.WM * c.WN, step = nth / (c.CE / 4), thtw = d.TH << c.TWL;
    return step > 0 && thtw % step == 0;
}

// The few-segment twin of a planned WHOLE-IMAGE block, if its configuration has one: the same chunk size, k steps, precision,
// activation and project-tile count (the block's weights serve it too), one segment per workgroup, and a NARROWER tile -- two or
// four workgroups per image, each expanding the halo columns of its own tile again.  A launch of a few dozen segments leaves most of
// the chip without a workgroup and lasts as long as one workgroup's walk through the chunks; narrower tiles shorten that walk.
// (A pixel's sums do not depend on the tile it is computed in: the results are bit-identical to the wide tiles'.)
bool mb_plan_narrow(const MbDesc &d, MbDesc &narrow) {
    if (d.cfg < 0 || d.cfg >= kNCfgs) return false;
    const MbCfg &c = kCfgs[d.cfg];
    if (c.PERSIST || d.tiles_x * d.tiles_y != 1) return false;
    const int k0 = (d.cfg / kNBase) * kNBase;   // the same activation's copy of the list
    int best_twl = 99;
    for (int b = 0; b < kNBase; b++) {
        const MbCfg &q = kCfgs[k0 + b];
        if (q.S != 1 || q.COLTH != c.COLTH || q.KS != c.KS || q.ST != c.ST || q.CE != c.CE || q.KG != c.KG || q.PREC != c.PREC ||
            q.TWL >= c.TWL || q.TWL >= best_twl || q.WM * q.WN != c.WM * c.WN || q.WN * q.NT_W != c.WN * c.NT_W || q.STEM != c.STEM || q.PERSIST) continue;
        MbDesc t = d;
        if (mb_try_th(t, k0 + b, q.TH, true) >= 0 && t.NTOP == d.NTOP && t.nchunks == d.nchunks && t.KG == d.KG && t.CE == d.CE && t.tiles_y == 1 && t.tiles_x >= 2) {
            narrow = t; best_twl = q.TWL;
        }
    }
    return best_twl != 99;
}

__global__ __launch_bounds__(256) void mb_reduce_partials_kernel(const float4 *__restrict__ partial, float4 *__restrict__ Y, int ksplit, size_t count4, float unscale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    float4 a = partial[i];
    for (int k = 1; k < ksplit; k++) {      // (index order: the regime's one summation order)
        const float4 b = partial[(size_t)k * count4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    Y[i] = make_float4(a.x * unscale, a.y * unscale, a.z * unscale, a.w * unscale);
}

void launch_mb_reduce_partials(const float *partial, float *Y, int ksplit, size_t count, float unscale, hipStream_t s) {
    const size_t count4 = count / 4;      // (Cout % 4 == 0: every fused block's width is)
    hipLaunchKernelGGL(mb_reduce_partials_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, s, (const float4 *)partial, (float4 *)Y, ksplit, count4, unscale);
}

void launch_mbconv(const MbDesc &d, int n_seg, hipStream_t s) {
    if (d.se) kCfgs[d.cfg].launch_se(d, n_seg, s);
    else kCfgs[d.cfg].launch(d, n_seg, s);
}

}  // namespace bh
