// The fused MBConv kernel's tile configurations (mbconv_cfgs.inc, part 2 of 3) instantiated for ONE activation: ACT_RELU6,
// every entry a second time as pass A of a squeeze-excite block (MB_WITH_SE, mbconv_kernel.hpp SE = 1: swish since round 5 --
// EfficientNet's, Perch v2's backbone -- GELU and ReLU6 since round 6).
#define MB_WITH_SE 1
#include "mbconv_kernel.hpp"

namespace bh {

namespace {
#define MB_A ACT_RELU6
#define MB_PART 2
const MbCfg kTable[] = {
#include "mbconv_cfgs.inc"
};
#undef MB_PART
#undef MB_A
}  // namespace

const MbCfg *mb_table_relu6_p2(int *n) {
    if (n) *n = (int)(sizeof(kTable) / sizeof(kTable[0]));
    return kTable;
}

}  // namespace bh
