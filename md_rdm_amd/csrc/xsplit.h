// internal launcher declarations (xsplit.hip): split-precision (bf16x3) MFMA kernels for the backward GEMMs of the many-pixel dense blocks
#pragma once
#include <hip/hip_runtime.h>
#include "rdm_common.h"
namespace rdm {
bool xs_wgrad1x1_supported(const WgradArgs& a);
int launch_xs_wgrad1x1(const WgradArgs& a, hipStream_t s);      // same operands and meaning as launch_conv_wgrad for a 1x1 / stride 1 convolution
// 3x3 / stride 1 / pad 1 input gradient with 48 gradient channels (FwdArgs as launch_conv_fwd takes them for a dgrad: A = output gradient, Wt = packed
// weights [tap][48][N], out = input gradient [M][N]); epi = EPI_STORE or EPI_MASK_STATS; ws = xs_dgrad3x3_workspace_bytes(N) bytes of scratch
bool xs_dgrad3x3_supported(const FwdArgs& a);
size_t xs_dgrad3x3_workspace_bytes(int Cb);
int launch_xs_dgrad3x3(const FwdArgs& a, Epilogue epi, void* ws, size_t ws_bytes, hipStream_t s, int np = 3, bool prepacked = false);      // np: bf16 MFMAs per product - 3 = split, 1 = operands rounded to bf16
// 1x1 / stride 1 input gradient (FwdArgs as for a dgrad: A = output gradient [M][C = contracted channels], Wt = weights [C][N], out = [M][N])
bool xs_dgrad1x1_supported(const FwdArgs& a);
size_t xs_dgrad1x1_workspace_bytes(int K, int C);
int launch_xs_dgrad1x1(const FwdArgs& a, Epilogue epi, void* ws, size_t ws_bytes, hipStream_t s, int np = 3, bool prepacked = false);
// 3x3 / stride 1 / pad 1 weight gradient with <= 48 output channels: operands and meaning of launch_conv_wgrad (dW pre-zeroed, accumulated)
// frame image of the gradient operand (a.g_frame = 1, a.G = the image): split rows [padded position][48] with zeros on the frames' borders
size_t xs_frame_rows_bytes(int B, int H, int W);
int launch_frame_split_rows(const float* G, int ldg, int N, int B, int H, int W, void* dst, hipStream_t s);
bool xs_wgrad3x3_supported(const WgradArgs& a);
int launch_xs_wgrad3x3(const WgradArgs& a, hipStream_t s);
// 1x1 / stride 1 FORWARD with a three-way split (six bf16 MFMAs per product: float32-equivalent); FwdArgs as launch_conv_fwd takes them
// (A = input [M][C] with the BatchNorm + ReLU prologue, Wt = weights [N][C], out = [M][N]); epi = EPI_STORE or EPI_STORE_STATS
bool xs_fwd1x1_supported(const FwdArgs& a);
size_t xs_fwd1x1_workspace_bytes(int K, int N);
int launch_xs_fwd1x1(const FwdArgs& a, Epilogue epi, void* ws, size_t ws_bytes, hipStream_t s, int np = 6, bool prepacked = false);       // 6 = three-way split, 1 = operands rounded to bf16
// the weight re-formatting launches of the three kernels above on their own (prepacked = true then skips them: a plan packs every layer's weights
// once per step on its side stream, off the dependent chain); ws as the kernel's workspace query sizes it
int launch_xs_pack_w3_dgrad(const float* w, long wtap, int ldw, int Cb, void* ws, hipStream_t s);
int launch_xs_pack_w1_dgrad(const float* w, int ldw, int K, int C, void* ws, hipStream_t s);
int launch_xs_pack_w1_fwd(const float* w, int ldw, int N, int K, void* ws, hipStream_t s);
}  // namespace rdm
