// One row of the fused MBConv kernel's table of tile configurations (mbconv_cfgs.inc), and the three per-activation copies of
// the table: each is built by its own translation unit (kernels_mbconv_gelu.hip / _swish.hip / _relu6.hip) so that hipcc compiles
// them in parallel; the planner (kernels_mbconv.hip) reads them through these accessors.  (The tables themselves have internal
// linkage: a `const` array with external linkage is emitted into the DEVICE code object as well, where its host function
// pointers do not link.)
#pragma once
#include "kernels.hpp"

namespace bh {

struct MbCfg {
    int KS, ST, CE, KG, RT_W, NCS, WM, WN, MT_W, NT_W, TWL, TH, S, STEM, XBL, OCC, PREC, PERSIST, ACT, COLTH;
    void (*launch)(const MbDesc &, int, hipStream_t);   // nullptr: not part of this build (KS = 0 matches no block)
    void (*launch_se)(const MbDesc &, int, hipStream_t);   // pass A of a squeeze-excite block (MbDesc::se); nullptr: not instantiated for this activation / entry
};

// (three parts of the list per activation, one translation unit each: kernels_mbconv_<act>[_p1 | _p2].hip)
#define BH_MB_TABLE_DECL(act) const MbCfg *mb_table_##act##_p0(int *n); const MbCfg *mb_table_##act##_p1(int *n); const MbCfg *mb_table_##act##_p2(int *n);
BH_MB_TABLE_DECL(gelu) BH_MB_TABLE_DECL(swish) BH_MB_TABLE_DECL(relu6)
#undef BH_MB_TABLE_DECL

}  // namespace bh
