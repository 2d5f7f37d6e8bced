// internal launcher declarations (xsplit.hip): split-precision (bf16x3) MFMA kernels for the backward GEMMs of the many-pixel dense blocks
#pragma once
#include <hip/hip_runtime.h>
#include "rdm_common.h"
namespace rdm {
bool xs_wgrad1x1_supported(const WgradArgs& a);
int launch_xs_wgrad1x1(const WgradArgs& a, hipStream_t s);      // same operands and meaning as launch_conv_wgrad for a 1x1 / stride 1 convolution
}  // namespace rdm
