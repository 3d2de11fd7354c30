// The fused MBConv kernel's tile configurations (mbconv_cfgs.inc) instantiated for ONE activation: ACT_GELU_ERF.
#include "mbconv_kernel.hpp"

namespace bh {

namespace {
#define MB_A ACT_GELU_ERF
const MbCfg kTable[] = {
#include "mbconv_cfgs.inc"
};
#undef MB_A
}  // namespace

const MbCfg *mb_table_gelu(int *n) {
    if (n) *n = (int)(sizeof(kTable) / sizeof(kTable[0]));
    return kTable;
}

}  // namespace bh
