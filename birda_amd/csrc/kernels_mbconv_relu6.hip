// The fused MBConv kernel's tile configurations (mbconv_cfgs.inc) instantiated for ONE activation: ACT_RELU6.
// (round 6: every entry also as pass A of a squeeze-excite block, as the swish unit since round 5 -- gated GELU / ReLU6 stacks)
#define MB_WITH_SE 1
#include "mbconv_kernel.hpp"

namespace bh {

namespace {
#define MB_A ACT_RELU6
const MbCfg kTable[] = {
#include "mbconv_cfgs.inc"
};
#undef MB_A
}  // namespace

const MbCfg *mb_table_relu6(int *n) {
    if (n) *n = (int)(sizeof(kTable) / sizeof(kTable[0]));
    return kTable;
}

}  // namespace bh
