// internal launcher declarations (wino.hip): Winograd F(2x2, 3x3) for the 48-output 3x3 convolutions of the dense layers
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
namespace rdm {
struct WinoConv {
  const float* A; int lda; int C;                 // raw NHWC input (pre BatchNorm), contracted channels (multiple of 16)
  const float* a_scale; const float* a_shift;     // consumer BatchNorm + ReLU, per contracted channel (NULL: identity)
  const float* U;                                 // launch_wino_weight output, 16 * 48 * C floats
  float* out; int ldc; int N;                     // output slice (pixel stride ldc), N <= 48 channels
  int B, H, W;
  int split;                                      // 0: launcher's choice; > 1 needs `partial`
  float* partial; size_t partial_floats;          // scratch for the per-split partial outputs, split * B*H*W * 48 floats
  double* stat0; double* stat1;                   // optional: += sum / sum of squares of the output per channel
  int x6;                                         // 1: the bf16x6 kernel (float32-equivalent, bf16 matrix pipe); U then in launch_wino_weight(..., x6 = true) form
};
struct WinoWgrad {                                // dW[tap][n][c] = sum_m G[m][n] * f(A[pix(m, tap)][c]), written (not accumulated)
  const float* G; int ldg; int N;                 // output gradient, N <= 48 channels
  const float* A; int lda; int C;                 // forward input (pre BatchNorm), Cb channels
  const float* a_scale; const float* a_shift;
  float* dW; long wtap; int ldw;                  // packed gradient [tap][n][c]
  float* Vy; size_t vy_floats;                    // scratch: transformed gradient (wino_wgrad_vy_floats)
  float* part; size_t part_floats;                // scratch: per-split partial sums (wino_wgrad_part_floats)
  int B, H, W;
};
size_t wino_wgrad_vy_floats(int B, int H, int W);
size_t wino_wgrad_part_floats(int B, int H, int W, int C);
int launch_conv3x3_wino_wgrad(const WinoWgrad& a, hipStream_t s);
size_t wino_u_bytes(int C, bool x6);                                       // transformed-weight image of one layer
size_t wino_fwd_workspace_bytes(int C, long M, int split, bool x6 = false);
int wino_pick_split(int tiles, int nslab, bool x6 = false);
int launch_wino_weight(const float* w_packed, long wtap, int ldw, int N, int C, float* U, hipStream_t s, bool x6 = false);
int launch_conv3x3_wino_fwd(const WinoConv& a, hipStream_t s);
}  // namespace rdm
