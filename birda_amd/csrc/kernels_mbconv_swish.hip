// The fused MBConv kernel's tile configurations (mbconv_cfgs.inc) instantiated for ONE activation: ACT_SWISH.
// ... and, swish being the activation of the stacks that carry squeeze-excite gates (EfficientNet: Perch v2's backbone), every
// entry a second time as pass A of such a block (MB_WITH_SE, mbconv_kernel.hpp SE = 1).
#define MB_WITH_SE 1
#include "mbconv_kernel.hpp"

namespace bh {

namespace {
#define MB_A ACT_SWISH
const MbCfg kTable[] = {
#include "mbconv_cfgs.inc"
};
#undef MB_A
}  // namespace

const MbCfg *mb_table_swish(int *n) {
    if (n) *n = (int)(sizeof(kTable) / sizeof(kTable[0]));
    return kTable;
}

}  // namespace bh
