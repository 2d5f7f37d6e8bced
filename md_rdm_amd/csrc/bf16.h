// internal declarations of the bf16 forward kernels (bf16.hip); not part of the C ABI
#pragma once
#include <hip/hip_runtime.h>
namespace rdm {

struct GemmBf16Args {        // out[m][n] = bias[n] + sum_k f(X[m][k]) * W[n][k],  f = relu(x*scale[k]+shift[k]) when scale != NULL
  const void* X; int ldx; int K;           // bf16 activations [M][ldx] (k-contiguous), contracted extent (multiple of 8)
  const float* scale; const float* shift;  // optional eval-mode BatchNorm affine + ReLU over k
  const void* W; int ldw;                  // bf16 weights [N][ldw]
  void* out; int ldc; int M, N;            // bf16 (or f32) [M][ldc]
  const float* bias;                       // optional f32[N]
  const float* oscale; const float* oshift; // optional f32[N]: out = relu(out*oscale[n] + oshift[n]) before the bf16 rounding (the CONSUMER's eval-mode BatchNorm + ReLU)
  float* partial; size_t partial_floats;   // optional f32 scratch for a K-split ([split][M][N] partial sums + one tiny reduction launch)
  unsigned x_bytes, w_bytes, p_bytes, o_bytes;   // set by the launcher (buffer descriptors)
  int abl;                                 // development builds: ablation bits (0 in the shipped build)
};

struct Conv3Bf16Args {       // out[m][n] = sum_{tap,c} relu(Y[pix(m,tap)][c]*scale[c]+shift[c]) * Wt[tap][n][c], zero padding, n < 48
  const void* Y; int ldy; int C;           // bf16 [M][ldy], C multiple of 8
  const float* scale; const float* shift;
  const void* Wt; long wtap; int ldw;      // bf16 [9][48][ldw]
  unsigned short* out; int ldc;            // bf16 [M][ldc] (a 48-channel slice of the block buffer)
  int B, H, W, M;
  float* partial; size_t partial_floats;   // optional f32 scratch for the K-split: [split][M][48] partial sums, reduced by a second (tiny) launch
  int split;                               // set by the launcher
  int slots;                               // set by the launcher: zero-padded LDS image slots (pixels) the kernel may use, multiple of 8
  unsigned y_bytes, w_bytes, p_bytes;
};

struct Conv3ActArgs {        // out[m][n] = sum_{tap,c} Y[pix(m,tap)][c] * w[n][c][tap] on an ALREADY ACTIVATED input, zero padding, n < 48
  const void* Y; int ldy; int C;           // bf16 [B*H*W][ldy]; C multiple of 32 (pad channels: finite values, their weights are zero)
  const void* Wimg;                        // fragment-order weight image (launch_pack_w3_frag_bf16): C/32 slabs of 27 KiB
  unsigned short* out; int ldc;            // bf16 [B*H*W][ldc]
  int B, H, W;
  float* partial; size_t partial_floats;   // optional f32 scratch for the K-split ([tile][split][BM][48])
  unsigned* counters; int n_counters;      // one per tile, ZERO on entry, left zero (needed when partial != NULL)
  int split, slots, tiles_per_img, rect;   // set by the launcher (rect: 64-column rectangular tiles for rows too wide for whole-row tiles)
  unsigned y_bytes, w_bytes;
  int abl;                                 // development builds: ablation bits (0 in the shipped build)
};

int launch_gemm_bf16(const GemmBf16Args& a, bool out_f32, hipStream_t s);
int launch_conv3x3_act_bf16(const Conv3ActArgs& a, hipStream_t s);      // RDM_ERR_UNSUPPORTED when the geometry does not fit the LDS (nothing launched)
size_t conv3x3_act_partial_floats(int C, int B, int H, int W);          // scratch the heuristic would like
int conv3x3_act_tiles(int B, int H, int W);                             // upper bound of the tile count (counters needed)
bool conv3x3_act_fits(int B, int H, int W);                             // a tile's padded rows fit the LDS image
int launch_pack_w3_frag_bf16(const float* w, void* img, int C, int Cpad, int packed, hipStream_t s);
int launch_conv3x3_bf16(const Conv3Bf16Args& a, hipStream_t s);
int launch_f32_to_bf16_rows(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, int cols_pad, hipStream_t s);
int launch_pack_w_bf16(const float* w_oihw, void* w_packed, int O, int I, int ld, int T, hipStream_t s);
int launch_im2col_stem_bf16(const float* x, void* patches, int B, int H, int W, hipStream_t s);
int launch_maxpool3s2_bf16(const void* X, void* Y, int ldy, int B, int H, int W, int C, hipStream_t s);
int launch_trans_pool_bf16(const void* X, int ldx, const float* sc, const float* sh, void* P, int B, int H, int W, int C, hipStream_t s);

}  // namespace rdm
