/* librdm_hip.so - C ABI of the MI355X-native (gfx950) hot path of az16/MD_RDM.
 *
 * The reference is pure Python and has no FFI layer; the stable surface it
 * offers is `network.RDM_Net.DepthEstimationNet` + `network.computations`.  This header is the
 * INNER boundary a maintainer binds that surface to (see INTEGRATION.md for the ctypes stub):
 * every entry point below names the reference code it replaces (file:line in /root/reference).
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless marked host.
 *  - the caller (PyTorch) owns every buffer; the library never allocates or frees device
 *    memory and keeps no reference after return.  Workspace needs are answered by *_bytes().
 *  - all work is enqueued on the caller's `hipStream_t`; no entry point synchronises the host.
 *  - return 0 (RDM_OK) or a negative rdm_status; rdm_last_error_string() (thread-local) says why.
 *    Nothing throws or aborts across the ABI.
 *  - activations inside the conv stack are NHWC ("pixel-major"): element (b,y,x,c) of a tensor
 *    with pixel stride `ld` lives at ((b*H + y)*W + x)*ld + c.  A tensor that is a channel slice
 *    of a wider buffer simply has ld > channels (concat-free DenseNet blocks).
 *  - packed conv weights are [tap = r*kw + s][out_channel][in_channel] (float32).
 */
#ifndef RDM_HIP_H_
#define RDM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* rdm_stream_t; /* == hipStream_t */

typedef enum rdm_status {
  RDM_OK = 0,
  RDM_ERR_BAD_ARGUMENT = -1,
  RDM_ERR_WORKSPACE_TOO_SMALL = -2,
  RDM_ERR_HIP = -3,
  RDM_ERR_UNSUPPORTED = -4
} rdm_status;

const char* rdm_last_error_string(void);
int rdm_version(void);

/* Per-launch HIP-event timing of the MFMA conv kernels (bench.py's roofline leg).  While enabled,
 * every conv launch is bracketed by events on its own stream; rdm_profile_read() synchronises on
 * them (HOST sync - never call it inside a graph capture), returns the summed kernel time (ms), the
 * length of the UNION of the kernels' intervals (weight-gradient kernels overlap the dgrad chain on the
 * library's side stream, so the sum over-counts), the FLOPs the launches executed and their count,
 * and clears the record. */
void rdm_profile_enable(int32_t on);
int rdm_profile_read(double* conv_ms_sum, double* conv_ms_union, double* conv_flops, int32_t* launches);
/* Per-kernel breakdown of the LAST rdm_profile_read(): kind = 0..18 (returns RDM_ERR_BAD_ARGUMENT beyond; 0-5 the direct f32 MFMA kernels, 13-16 the split-precision bf16x3 gradient kernels - FLOPs = the float32 product's, i.e. algorithmic,
 * 7-8 and 11-12 the bf16 forward kernels, 9-10 the Winograd f32 forward / weight-gradient kernels - their FLOPs are the DIRECT convolution's, i.e. algorithmic), *name = static string naming the kernel, summed duration (ms), executed FLOPs and launch count. */
int rdm_profile_kind(int32_t kind, const char** name, double* ms_sum, double* flops, int32_t* launches);
/* algorithmic HBM bytes (operands read once + result written once) of those launches; kept for the bf16 kernels (0 for the f32 kinds) */
double rdm_profile_kind_bytes(int32_t kind);
/* (The development A/B switch of earlier rounds is NOT part of this ABI: it is declared in include/rdm_dev.h and exists only in libraries
 * built with RDM_DEV_VARIANTS=1; the shipped library does not export it.) */
/* number of kernel-launching calls the library has made in this process (monotonic; bench.py reports the per-step difference;
 * a K-split launcher that also enqueues its zero-fill or reduction counts once per enqueued kernel family) */
int64_t rdm_launch_count(void);

/* Launch census (test instrumentation): while enabled, every MFMA conv launcher records which kernel VARIANT it picked - e.g.
 * "conv3x3_halo_kernel/dgrad/px256/hl6/MASK_STATS", "xs_dgrad3x3_kernel/MASK_STATS", "conv1x1_dma256_kernel/STORE_STATS/bn1" -
 * so tests/test_gpu_conv.py can assert that the variants the headline geometry (B=16, 228x304) selects are the ones its parity cases
 * ran.  rdm_census_count() snapshots the table and returns its size; rdm_census_entry(i) reads entry i of that snapshot (the name
 * stays valid until the next rdm_census_count()). */
void rdm_census_enable(int32_t on);
void rdm_census_reset(void);
int32_t rdm_census_count(void);
int rdm_census_entry(int32_t i, const char** name, int64_t* launches);

/* ------------------------------------------------------------------------------------------
 * Convolution family (fp32 MFMA implicit GEMM).  Replaces the nn.Conv2d / torchvision
 * _DenseLayer / _Transition convolutions of network/RDM_Net.py:144,146-147,524-531 and the WSM
 * convolutions of :163-236, plus their autograd-generated backward passes.
 * ------------------------------------------------------------------------------------------ */
typedef struct rdm_conv_desc {
  int32_t batch, in_h, in_w;     /* input extent */
  int32_t in_c, in_ld;           /* contracted channels (multiple of 16) and input pixel stride */
  int32_t out_c, out_ld;         /* output channels and output pixel stride */
  int32_t kh, kw, stride_h, stride_w, pad_h, pad_w;
} rdm_conv_desc;

/* y[b,oy,ox,n] = bias[n] + sum_{r,s,c} f(x[b, oy*sh-ph+r, ox*sw-pw+s, c]) * w[r*kw+s][n][c]
 * f = identity, or relu(x*bn_scale[c] + bn_shift[c]) when bn_scale != NULL (the BatchNorm+ReLU
 * that precedes the conv in torchvision's _DenseLayer; zero padding applies AFTER f).
 * stat_sum/stat_sq (optional, f64[out_c], pre-zeroed): += sum_m y and sum_m y^2 per channel. */
int rdm_conv2d_fwd(const rdm_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                   const float* bn_scale, const float* bn_shift, float* y, double* stat_sum, double* stat_sq,
                   rdm_stream_t stream);

/* dx[b,y,x,c] = sum_{r,s,n} dy[b, y+ph-r, x+pw-s, n] * w[r*kw+s][n][c]      (stride 1 only)
 * If mask_x != NULL the result is gated by relu'(mask_x*mask_scale+mask_shift) (the ReLU that fed
 * the conv) and stat_a += sum_m dx, stat_b += sum_m dx*mask_x (BatchNorm-backward reductions). */
int rdm_conv2d_dgrad(const rdm_conv_desc* d, const float* dy, const float* w_packed, float* dx, int32_t dx_ld,
                     const float* mask_x, int32_t mask_ld, const float* mask_scale, const float* mask_shift,
                     double* stat_a, double* stat_b, rdm_stream_t stream);

/* dw[r*kw+s][n][c] += sum_{b,oy,ox} dy[b,oy,ox,n] * f(x[b, oy*sh-ph+r, ox*sw-pw+s, c]);  dw pre-zeroed. */
int rdm_conv2d_wgrad(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale,
                     const float* bn_shift, float* dw_packed, rdm_stream_t stream);

/* rdm_conv2d_fwd with the BatchNorm + ReLU prologue given as the RAW training-mode channel sums (sum x, sum x^2 over bn_count elements,
 * gamma, beta; eps 1e-5): the kernel forms scale = gamma / sqrt(var + eps), shift = beta - mean * scale itself, with the arithmetic of
 * rdm_bn_finalize, so the finalisation launch leaves the dependent chain.  Built for the few-pixel blocks only (dense_e4 / decoder d_1 in the
 * native plan): 1x1 convs on 128 x 48 tiles (<= 32 768 pixels) and 3x3 / pad 1 convs with <= 48 outputs on 128-pixel tiles (<= 8 192 pixels,
 * width <= 63), in_c <= 768; anything else returns RDM_ERR_BAD_ARGUMENT. */
int rdm_conv2d_fwd_bnsums(const rdm_conv_desc* d, const float* x, const float* w_packed, const double* bn_sum, const double* bn_sumsq, double bn_count,
                          const float* bn_gamma, const float* bn_beta, float* y, double* stat_sum, double* stat_sq, int32_t split_k,
                          rdm_stream_t stream);

/* The 3x3 / pad 1 convolution of a few-pixel dense layer as rdm_net_forward runs it in training (round 5): raw-BatchNorm prologue as above, the result
 * ADDED into y (the caller zeroes the 48-channel slice once per block; any K split, f32 atomics), and the channel statistics of the FINISHED y taken
 * in the same launch - tile_tickets: one zeroed uint32 per 128-pixel tile (ceil(pixels / 128)), left zero; the last K split of a tile to arrive
 * reduces the tile into stat_sum / stat_sq (f64, pre-zeroed, +=).  RDM_ERR_BAD_ARGUMENT outside the raw form's limits (<= 8 192 pixels, width <= 63,
 * in_c <= 768, out_c <= 48, y rows 16-byte aligned). */
int rdm_conv3x3_fwd_bnsums_acc(const rdm_conv_desc* d, const float* x, const float* w_packed, const double* bn_sum, const double* bn_sumsq, double bn_count,
                               const float* bn_gamma, const float* bn_beta, float* y, double* stat_sum, double* stat_sq, uint32_t* tile_tickets,
                               int32_t split_k, rdm_stream_t stream);

/* The same three operators with the K-split chosen by the caller: split_k = 0 keeps the launcher's own choice (what the plain entry
 * points do), 1 forbids splitting (no atomics: one workgroup owns every output element, bit-reproducible), n > 1 asks for n partial
 * sums added with f32 atomics into a zeroed output (the launcher clamps n to the number of K slabs; statistics / bias epilogues
 * that are not linear in the partials always run unsplit). */
int rdm_conv2d_fwd_ex(const rdm_conv_desc* d, const float* x, const float* w_packed, const float* bias, const float* bn_scale,
                      const float* bn_shift, float* y, double* stat_sum, double* stat_sq, int32_t split_k, rdm_stream_t stream);
int rdm_conv2d_dgrad_ex(const rdm_conv_desc* d, const float* dy, const float* w_packed, float* dx, int32_t dx_ld, const float* mask_x,
                        int32_t mask_ld, const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, int32_t split_k,
                        rdm_stream_t stream);
int rdm_conv2d_wgrad_ex(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift,
                        float* dw_packed, int32_t split_k, rdm_stream_t stream);

/* Split-precision ("bf16x3") gradient kernels (csrc/xsplit.hip) for the autograd-generated weight / input gradients of the dense layers with
 * many pixels (torchvision _DenseLayer.conv1 / conv2 reached from network/RDM_Net.py:526,528).  Same operands and meaning as the _ex entry
 * points above; every float32 operand is split into two bf16 pieces at staging time and each product is formed as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the bf16 matrix cores with float32 accumulation (error ~5e-6 of the result's maximum; the exact-f32
 * MFMA kernels: ~1e-6).  Gradients only - the forward pass never goes through these.  RDM_ERR_UNSUPPORTED for shapes without such a kernel
 * (1x1: 96 <= in_c <= 2304, in_c a multiple of 48; 3x3 / stride 1 / pad 1: out_c <= 48, rows of <= 93 pixels). */
/* `products` (all four entry points): bf16 MFMAs per float32 product - 0 = the split arithmetic (3; 6 for the forward), 1 = operands simply ROUNDED to
 * bf16 (one MFMA, float32 accumulation): the arithmetic of a mixed-precision (AMP) step, ~2e-3 of the result's maximum; RDM_NET_OPT_GEMM_BF16. */
/* SPLIT ROWS (round 5): a float32 row of C values (C a multiple of 4) stored in the same 4 C bytes as, per four consecutive values,
 * [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] bf16 (hi = bf16(x), lo = bf16(x - hi)) - the two halves the split kernels feed their MFMAs.  An operand handed
 * over in this form is staged verbatim: its conversion is paid once by the producer (rdm_split_rows_f32; rdm_bn_bwd with accumulate = 2) instead of
 * once per consumer tile.  OR these flags into `products` (split arithmetic only, 1x1 kernels only): the output-gradient operand `dy` / the
 * activation operand `x` (then already activated: bn_scale = bn_shift = NULL) is given as split rows. */
#define RDM_X3_DY_SPLIT_ROWS 0x10
#define RDM_X3_X_SPLIT_ROWS 0x20
/* 3x3 weight gradient only: `dy` is the FRAME IMAGE rdm_frame_split_rows_f32 wrote - split rows [padded position of the (h + 2) x (w + 2) frames][48]
 * with zeros on the frames' borders (rdm_frame_split_rows_bytes bytes).  The kernel contracts over padded positions, so a 32-position slab of the
 * gradient is 6 KB of consecutive bytes, staged verbatim by each of its (column block, K split) workgroups. */
#define RDM_X3_DY_FRAME_ROWS 0x40
/* rdm_conv1x1_dgrad_x3 with a mask only: the epilogue ADDS mask_scale[c] * (gated dz) into dx (the block gradient, dx_ld = its row stride) instead
 * of storing the gated dz - the a * dz term of the norm1 backward taken by the input gradient itself (mask_scale = gamma * rstd is the forward's
 * BatchNorm scale); the b * x + c terms follow through rdm_bn_bwd_defer.  network/RDM_Net.py:526 (autograd of torchvision _DenseLayer.norm1). */
#define RDM_X3_ACC_SCALED 0x80
size_t rdm_frame_split_rows_bytes(int32_t batch, int32_t h, int32_t w);
int rdm_frame_split_rows_f32(const float* dy, int32_t dy_ld, int32_t channels, int32_t batch, int32_t h, int32_t w, void* dst, rdm_stream_t stream);
/* dst (split rows, row stride dst_ld floats' worth of bytes) = split(ReLU(bn_scale * src + bn_shift)), or split(src) with bn_scale = bn_shift = NULL:
 * relu1(norm1(x)) of a dense layer (torchvision _DenseLayer reached from network/RDM_Net.py:526-530) as the operand of its conv1 weight gradient. */
int rdm_split_rows_f32(const float* src, int32_t src_ld, const float* bn_scale, const float* bn_shift, void* dst, int32_t dst_ld, int64_t rows,
                       int32_t channels, rdm_stream_t stream);
int rdm_conv2d_wgrad_x3(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift,
                        float* dw_packed, int32_t split_k, int32_t products, rdm_stream_t stream);
/* 3x3 / stride 1 / pad 1 input gradient with out_c = 48 (the dense layers' conv2): operands and meaning of rdm_conv2d_dgrad (gate + BatchNorm-backward
 * sums when mask_x is given).  The workspace receives the split weights in MFMA-fragment order (re-formed on every call). */
/* The dense layers' conv1 FORWARD (1x1 / stride 1; torchvision _DenseLayer.conv1 reached from network/RDM_Net.py:526-530) with a THREE-way split -
 * six bf16 MFMAs per product, float32-equivalent accuracy (the two-way split of the gradient kernels is not accurate enough for the forward's 1e-4
 * parity): operands and meaning of rdm_conv2d_fwd without bias (BatchNorm + ReLU prologue on x, optional per-channel sums of y and y^2). */
size_t rdm_conv1x1_fwd_x6_workspace_bytes(int32_t in_c, int32_t out_c);
int rdm_conv1x1_fwd_x6(const rdm_conv_desc* d, const float* x, const float* w_packed, const float* bn_scale, const float* bn_shift, float* y,
                       double* stat_sum, double* stat_sq, void* workspace, size_t workspace_bytes, int32_t products, rdm_stream_t stream);
size_t rdm_conv1x1_dgrad_x3_workspace_bytes(int32_t out_c, int32_t in_c);
int rdm_conv1x1_dgrad_x3(const rdm_conv_desc* d, const float* dy, const float* w_packed, float* dx, int32_t dx_ld, const float* mask_x,
                         int32_t mask_ld, const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, void* workspace,
                         size_t workspace_bytes, int32_t products, rdm_stream_t stream);      /* 1x1 / stride 1 (the dense layers' conv1): w_packed = [out_c][in_c] */
size_t rdm_conv3x3_dgrad_x3_workspace_bytes(int32_t in_c);
int rdm_conv3x3_dgrad_x3(const rdm_conv_desc* d, const float* dy, const float* w_packed, float* dx, int32_t dx_ld, const float* mask_x,
                         int32_t mask_ld, const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, void* workspace,
                         size_t workspace_bytes, int32_t products, rdm_stream_t stream);

/* The 3x3 / stride 1 / pad 1 convolution with <= 48 outputs (torchvision _DenseLayer.conv2 reached from network/RDM_Net.py:144,526-530)
 * as Winograd F(2x2, 3x3) on the f32 MFMA path: 2.25x fewer multiply-adds than rdm_conv2d_fwd for the same result up to float32
 * rounding (tests/test_gpu_wino.py holds it to the same 2e-5 against a float64 evaluation).  Same operands and meaning as
 * rdm_conv2d_fwd (w_packed [9][out_c][in_c], BatchNorm + ReLU prologue, zero padding after it, optional channel statistics); the
 * workspace (rdm_conv3x3_wino_workspace_bytes, 256-byte aligned) holds the transformed weights and, for a K-split, the per-split
 * partial outputs, which are summed in a FIXED order (no atomics: the result is bit-reproducible).  split_k = 0: the launcher's choice. */
size_t rdm_conv3x3_wino_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w, int32_t split_k);
int rdm_conv3x3_wino_fwd(const rdm_conv_desc* d, const float* x, const float* w_packed, const float* bn_scale, const float* bn_shift, float* y,
                         double* stat_sum, double* stat_sq, void* workspace, size_t workspace_bytes, int32_t split_k, rdm_stream_t stream);

/* The same convolution on the bf16 matrix pipe with float32-EQUIVALENT arithmetic (round 5): both transformed operands split three ways
 * (v = v0 + v1 + v2 in bf16, 24 significant bits), six bf16 products per float32 product, float32 accumulation - the arithmetic of
 * rdm_conv1x1_fwd_x6 applied to the 16 position GEMMs of F(2x2, 3x3).  Same operands, meaning and tolerance as rdm_conv3x3_wino_fwd
 * (tests/test_gpu_wino.py holds both to 2e-5 of the output's maximum against float64, this one also to 5e-6).  Measured at dense_e2 / dense_e3
 * sizes: no faster than the f32 kernel - both are bound by the transform producers and the per-workgroup prologue / epilogue of the K split, not by
 * the matrix pipe (profiles/r05_wino_x6_ablation.txt) - so the plan keeps the f32 kernel by default (RDM_NET_OPT_WINO_X6). */
size_t rdm_conv3x3_wino_x6_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w, int32_t split_k);
int rdm_conv3x3_wino_fwd_x6(const rdm_conv_desc* d, const float* x, const float* w_packed, const float* bn_scale, const float* bn_shift, float* y,
                            double* stat_sum, double* stat_sq, void* workspace, size_t workspace_bytes, int32_t split_k, rdm_stream_t stream);
/* ... and the weight gradient of that convolution as Winograd F(3x3, 2x2): same operands and meaning as rdm_conv2d_wgrad (BatchNorm + ReLU
 * prologue on x), except that dw_packed [9][out_c][in_c] is WRITTEN, not accumulated, and that the K-split partial sums are combined
 * in a fixed order (no atomics: bit-reproducible).  The workspace holds the transformed gradient and the per-split partial sums. */
size_t rdm_conv3x3_wino_wgrad_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w);
int rdm_conv3x3_wino_wgrad(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw_packed,
                           void* workspace, size_t workspace_bytes, rdm_stream_t stream);

/* [out][in][kh][kw] (PyTorch) <-> [tap][out_pad][in] (packed); rows >= out_c are zero-filled. */
int rdm_pack_conv_weight(const float* w_oihw, float* w_packed, int32_t out_c, int32_t in_c, int32_t kh, int32_t kw,
                         int32_t out_c_padded, rdm_stream_t stream);
int rdm_unpack_conv_weight(const float* w_packed, float* w_oihw, int32_t out_c, int32_t in_c, int32_t kh, int32_t kw,
                           int32_t out_c_padded, rdm_stream_t stream);

/* bf16 operators of the reduced-precision forward (rdm_net_forward_bf16 below enqueues the same kernels; reference: the
 * mixed-precision convolutions of train.py:11,57-58 over network/RDM_Net.py:144,524-531).  bf16 = the upper 16 bits of an IEEE
 * float32, round-to-nearest-even; v_mfma_f32_16x16x32_bf16 with f32 accumulation.
 *   rdm_gemm_bf16     out[m][n] = bias[n] + sum_k f(x[m][k]) * w[n][k], f = relu(x*scale[k]+shift[k]) (rounded to bf16) when scale != NULL.
 *                     x (M, ldx) bf16, w (N, ldw) bf16, out (M, ldc) bf16 or f32 (out_f32 != 0); K, ldx, ldw multiples of 8; N, ldc of 4.
 *                     workspace (optional f32 scratch, 256-byte aligned, up to 8*M*N floats are used): lets a few-row / long-K product
 *                     (M <= 1024) split K over the grid; partial sums are reduced in a fixed order (deterministic).
 *   rdm_conv3x3_bf16  3x3 / stride 1 / pad 1 conv with 48 outputs: y (B,H,W, ldy) bf16 with the BN-ReLU prologue over its `channels`
 *                     (multiple of 8; scale = shift = NULL: y is already activated), w [9][48][channels] bf16, out (B*H*W, ldc) bf16 - typically a 48-channel slice of a wider buffer.
 *                     workspace (optional, 256-byte aligned): f32 scratch for a K-split that fills the chip when B*H*W is small -
 *                     partial sums are stored per split and reduced in a fixed order (deterministic); any size is accepted, the split is
 *                     sized to it (rdm_conv3x3_bf16_workspace_bytes gives the amount the heuristic would like). */
int rdm_gemm_bf16(const void* x, int32_t ldx, int32_t k, const float* scale, const float* shift, const void* w, int32_t ldw, const float* bias,
                  void* out, int32_t ldc, int32_t m, int32_t n, int32_t out_f32, void* workspace, size_t workspace_bytes, rdm_stream_t stream);
/* rdm_gemm_bf16 with the CONSUMER's eval-mode BatchNorm + ReLU in the epilogue: out = bf16(relu(acc * out_scale[n] + out_shift[n])) - the
 * dense layer's 1x1 in rdm_net_forward_bf16 (the 3x3 behind it then reads an already activated tensor: rdm_conv3x3_act_bf16).
 * Many rows x many outputs x K <= 352 (dense_e2) run a persistent panel kernel (activations in registers, weight tiles by LDS-DMA). */
int rdm_gemm_bf16_act(const void* x, int32_t ldx, int32_t k, const float* scale, const float* shift, const void* w, int32_t ldw, const float* out_scale,
                      const float* out_shift, void* out, int32_t ldc, int32_t m, int32_t n, void* workspace, size_t workspace_bytes, rdm_stream_t stream);
size_t rdm_conv3x3_bf16_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w);
int rdm_conv3x3_bf16(const void* y, int32_t ldy, int32_t channels, const float* scale, const float* shift, const void* w_packed, void* out,
                     int32_t ldc, int32_t batch, int32_t h, int32_t w, void* workspace, size_t workspace_bytes, rdm_stream_t stream);
/* The same 3x3 on an ALREADY ACTIVATED input (the form rdm_net_forward_bf16 runs: in eval mode the producing 1x1 applies BatchNorm +
 * ReLU in its epilogue - rdm_gemm_bf16_act - so the 3x3 has no prologue and both operands reach LDS by DMA, zero padding included):
 *   rdm_conv3x3_act_bf16_pack  w (48, channels, 3, 3) f32 OIHW -> the fragment-order bf16 image the kernel streams
 *                              (rdm_conv3x3_act_bf16_weight_bytes(channels) bytes; channels are padded to a multiple of 32 with zeros).
 *   rdm_conv3x3_act_bf16       y (B,H,W, ldy) bf16, `channels_padded` = the multiple of 32 the image was packed for (<= ldy; activations
 *                              behind the real channel count must be finite: their weights are zero), out (B*H*W, ldc) bf16.
 *                              workspace (optional, 256-byte aligned, rdm_conv3x3_act_bf16_workspace_bytes): tile counters + f32 partial
 *                              sums of a K-split; the last workgroup of a tile adds them in a fixed order inside the same launch (deterministic).
 *                              Rows too wide for whole-row tiles (above ~200 pixels) take 64-column rectangular tiles. */
size_t rdm_conv3x3_act_bf16_weight_bytes(int32_t channels);
int rdm_conv3x3_act_bf16_pack(const float* w_oihw, int32_t channels, void* w_image, rdm_stream_t stream);
size_t rdm_conv3x3_act_bf16_workspace_bytes(int32_t channels_padded, int32_t batch, int32_t h, int32_t w);
int rdm_conv3x3_act_bf16(const void* y_act, int32_t ldy, int32_t channels_padded, const void* w_image, void* out, int32_t ldc, int32_t batch,
                         int32_t h, int32_t w, void* workspace, size_t workspace_bytes, rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm / pooling pieces of the conv stack as operators (the plan below enqueues the same kernels).
 * Reference call sites: torchvision _DenseLayer / _Transition BatchNorm2d + ReLU reached from
 * network/RDM_Net.py:144,526-531 (third party, restated in tests/golden/make_golden.py), max_e1 :525,
 * pad_br + trans_e* (ZeroPad2d((0,1,0,1)) -> BatchNorm -> ReLU -> conv -> AvgPool2d(2)) :527,529,531-532.
 * All tensors NHWC float32 with a pixel stride `ld` >= channels; channels a multiple of 4; 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
/* sum[c] += sum_m x[m][c], sumsq[c] += sum_m x[m][c]^2 over `rows` pixels (f64, pre-zeroed by the caller; sumsq may be NULL) */
int rdm_bn_stats(const float* x, int32_t ld, int64_t rows, int32_t channels, double* sum, double* sumsq, rdm_stream_t stream);
/* nn.BatchNorm2d bookkeeping from the sums: training != 0 -> batch mean / biased variance over `count` elements per channel,
 * running_mean / running_var (momentum 0.1, unbiased variance) and *num_batches_tracked updated; training == 0 -> running
 * statistics.  Outputs: scale = gamma * rstd and shift = beta - mean * scale (what a conv prologue applies), mean and
 * rstd = 1/sqrt(var + 1e-5) (what backward needs). */
int rdm_bn_finalize(const double* sum, const double* sumsq, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean, float* save_rstd,
                    int32_t channels, int32_t training, rdm_stream_t stream);
/* ReLU gate + the two BatchNorm-backward reductions: dz[m][c] = dy[m][c] * [x*scale+shift > 0] (in place),
 * sum_dz[c] += sum_m dz, sum_dz_x[c] += sum_m dz * x (f64, pre-zeroed) */
int rdm_bn_bwd_reduce(float* dz, int32_t dz_ld, const float* x, int32_t x_ld, const float* scale, const float* shift, int64_t rows,
                      int32_t channels, double* sum_dz, double* sum_dz_x, rdm_stream_t stream);
/* BatchNorm backward from the reductions: dx (=|+= when accumulate) gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)) in training,
 * gamma*rstd*dz in eval; dgamma[c] = sum dz*xhat, dbeta[c] = sum dz (either may be NULL).  accumulate: 0 = write, 1 = add, 2 = write dx as
 * SPLIT ROWS (see rdm_split_rows_f32): the form in which the norm2 backward hands dY to the split-precision conv1 gradient kernels. */
int rdm_bn_bwd(float* dx, int32_t dx_ld, const float* dz, int32_t dz_ld, const float* x, int32_t x_ld, const double* sum_dz,
               const double* sum_dz_x, double count, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma,
               float* dbeta, int64_t rows, int32_t channels, int32_t accumulate, int32_t training, rdm_stream_t stream);
/* Deferred norm1 backward of a dense block (the plan's default on the split kernels; RDM_NET_OPT_DEFER_NORM1).  The BatchNorm backward of layer
 * i adds a dz + b x + c over ALL its input channels; a dz is added by the 1x1 input gradient (RDM_X3_ACC_SCALED), and since x - the block buffer's
 * channel - is the same for every layer reading it, the (b, c) of the layers walked so far are SUMMED per channel and applied once, when a channel's
 * gradient is read next.  One call per layer, last layer first: from the layer's reductions (sum_dz, sum_dz_x as rdm_bn_bwd takes them) it forms
 * (b, c) with rdm_bn_bwd's arithmetic, writes b_out / c_out [channels] = b_in / c_in + (b, c) (ping-pong buffers; b_in = c_in = zeros at the block's
 * last layer), dgamma / dbeta (either may be NULL), and applies g[:, slice_c0 : slice_c0 + slice_n] += b_out x + c_out to the slice whose gradient
 * the caller reads next (the 48 channels the layer below produced; after the first layer the block's input channels). */
int rdm_bn_bwd_defer(float* g, int32_t g_ld, const float* x, int32_t x_ld, const double* sum_dz, const double* sum_dz_x, double count,
                     const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, const float* b_in,
                     const float* c_in, float* b_out, float* c_out, int64_t rows, int32_t channels, int32_t slice_c0, int32_t slice_n,
                     int32_t training, rdm_stream_t stream);
/* nn.MaxPool2d(3, stride 2, padding 1): x (B,H,W,C) contiguous -> y (B,Ho,Wo,C) with pixel stride y_ld; argmax (B,Ho,Wo,C) uint8 = winning
 * tap r*3+s (first maximum in scan order, as ATen).  Backward is the gather form: dx (B,H,W,C) contiguous, every element written. */
int rdm_maxpool3s2_fwd(const float* x, float* y, int32_t y_ld, uint8_t* argmax, int32_t batch, int32_t h, int32_t w, int32_t channels,
                       rdm_stream_t stream);
int rdm_maxpool3s2_bwd(const float* dy, int32_t dy_ld, const uint8_t* argmax, float* dx, int32_t batch, int32_t h, int32_t w, int32_t channels,
                       rdm_stream_t stream);
/* Transition front end: pooled (B,ceil(H/2),ceil(W/2),C) contiguous = AvgPool2d(2)(relu(pad_br(x) * scale + shift)) - the zero row /
 * column of pad_br is a BatchNorm INPUT (it contributes relu(shift)), and (scale, shift) must come from statistics over the padded
 * extent: rdm_bn_stats over the real pixels with count = B*(H+1)*(W+1).  The 1x1 conv of the transition is linear and is applied
 * to the pooled tensor (rdm_conv2d_fwd). */
int rdm_padavgpool2_fwd(const float* x, int32_t x_ld, const float* scale, const float* shift, float* pooled, int32_t batch, int32_t h,
                        int32_t w, int32_t channels, rdm_stream_t stream);
/* its backward incl. the BatchNorm: dx (B,H,W,C; pixel stride dx_ld; every element written) and dgamma / dbeta from dpooled.
 * workspace: rdm_padavgpool2_bwd_workspace_bytes(channels) bytes, 256-byte aligned (reductions + per-channel coefficients). */
size_t rdm_padavgpool2_bwd_workspace_bytes(int32_t channels);
int rdm_padavgpool2_bwd(const float* dpooled, const float* x, int32_t x_ld, const float* scale, const float* shift, const float* gamma,
                        const float* save_mean, const float* save_rstd, float* dx, int32_t dx_ld, float* dgamma, float* dbeta, int32_t batch,
                        int32_t h, int32_t w, int32_t channels, int32_t training, void* workspace, size_t workspace_bytes, rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The whole convolutional stack of DepthEstimationNet as one native plan:
 *   network/RDM_Net.py:73-94 (encoder) + :150-159 (Decoder d_1 up to conv2), forward and backward,
 *   train-mode (batch statistics + running-stat update) or eval-mode BatchNorm.
 * `tensors` is a HOST array of rdm_net_num_tensors() device pointers, one per state_dict entry
 * in the reference's registration order (rdm_net_tensor_name(i) gives the key); `grads` likewise
 * (NULL entries = not wanted / buffers).  The workspace carries the saved activations from
 * forward to backward, so the same workspace must be passed to both.
 * ------------------------------------------------------------------------------------------ */
typedef struct rdm_net rdm_net;

int rdm_net_num_tensors(void);
const char* rdm_net_tensor_name(int32_t i);
int64_t rdm_net_tensor_numel(int32_t i);
int rdm_net_tensor_is_param(int32_t i);          /* 1: float parameter, 0: buffer (running stats / counter) */

int rdm_net_create(int32_t batch, int32_t height, int32_t width, rdm_net** out);
void rdm_net_destroy(rdm_net* net);
size_t rdm_net_workspace_bytes(const rdm_net* net);
int rdm_net_output_hw(const rdm_net* net, int32_t* h, int32_t* w);
/* Plan options (all default 0 = `tensors` / `grads` exactly as the reference's state_dict lays them out, the plan zeroes what it
 * accumulates into):
 *   RDM_NET_OPT_PACKED_3X3       the 78 dense-layer 3x3 weights `...denselayerN.conv2.weight` - and their gradients - are handed over
 *                                PACKED [tap][out][in] instead of PyTorch's [out][in][kh][kw] (md_rdm_amd keeps them packed inside its
 *                                flat parameter buffer and exposes OIHW-shaped strided views): removes 78 pack launches per forward
 *                                and 78 unpack launches + 78 scratch fills per backward.
 *   RDM_NET_OPT_GRADS_PREZEROED  the caller guarantees that every gradient tensor is all zero when backward stage 0 starts (one fill
 *                                of a flat gradient buffer instead of ~160 per-tensor fills inside the plan).
 *   RDM_NET_OPT_DIRECT_3X3       every exact-f32 3x3 convolution on the direct implicit-GEMM kernels (default 0: blocks with >= 8 192 pixels run
 *                                the Winograd F(2x2, 3x3) forward, blocks with >= 12 288 pixels the Winograd F(3x3, 2x2) weight gradient -
 *                                dense_e2 / dense_e3 at the headline geometry; csrc/wino.hip; same result to float32 rounding,
 *                                tests/test_gpu_wino.py).  With RDM_NET_OPT_SPLIT_BWD the weight gradient of those blocks runs the
 *                                split-precision kernel instead and this option then only concerns the forward.
 *   RDM_NET_OPT_DETERMINISTIC    ordered reductions everywhere (for tests): no K split (so no f32 atomics with more than one contributor),
 *                                channel statistics and the dgrad gate / BatchNorm-backward sums as separate passes with a fixed summation
 *                                order, no layer pipelining.  Two runs on the same inputs then give bit-identical gradients; several
 *                                times slower than the default.
 *   RDM_NET_OPT_JOIN_PER_SEGMENT rdm_net_backward_stage orders the library's side stream (weight gradients) before the caller's stream at the
 *                                end of each of the 4 SEGMENTS only, not after every stage: for callers that do not consume gradients
 *                                stage by stage (no data-parallel exchange).  The gradients of a segment are complete on the caller's
 *                                stream once its last stage has returned.
 *   RDM_NET_OPT_SPLIT_BWD        the weight / input gradient GEMMs of the dense blocks with >= 1 024 pixels (all four blocks at the headline
 *                                geometry; the 3x3 weight gradient from 8 192 pixels) run the split-precision bf16x3 kernels (rdm_conv2d_*_x3, csrc/xsplit.hip) instead of the exact-f32
 *                                MFMA kernels: gradients agree to ~5e-6 of a tensor's maximum, the forward pass is untouched.  Ignored in
 *                                deterministic mode.
 *   RDM_NET_OPT_SPLIT_FWD        the 1x1 convolutions (conv1) of the dense blocks with >= 8 192 pixels run the three-way-split bf16x6 kernel
 *                                (rdm_conv1x1_fwd_x6) in rdm_net_forward: float32-equivalent accuracy, write-bound instead of f32-MFMA-bound.
 *   RDM_NET_OPT_DEFER_NORM1      (default 1) blocks whose conv1 input gradient runs on the split kernels: that kernel's epilogue adds gamma * rstd * dz into
 *                                the block gradient itself, and the remaining terms of the norm1 BatchNorm backward (b * x + c per channel, which need the
 *                                kernel's channel sums) are summed over the layers and applied once per channel when its gradient is read next, instead of
 *                                one elementwise pass over all input channels per layer.  Same gradients to float32 rounding.
 *   RDM_NET_OPT_PREPACK          (default 1) the weight images of the split kernels (three-way split for the 1x1 forward, fragment order / transposed split
 *                                for the two input-gradient kernels) are formed for ALL layers once per training step on the library's side stream at the
 *                                start of rdm_net_forward, instead of by one small launch in front of every kernel on the dependent chain.
 *   RDM_NET_OPT_SPLIT_ROWS       (default 1) float32 arithmetic on the split kernels: the norm2 BatchNorm backward writes dY as split rows and relu1(norm1(x))
 *                                is activated + split once per layer (rdm_split_rows_f32) - the conv1 input / weight gradient kernels stage both operands
 *                                verbatim instead of converting and splitting them per tile.  Bit-identical gradients (the same two bf16 values per element).
 *   RDM_NET_OPT_WINO_X6          (default 0; with RDM_NET_OPT_SPLIT_FWD) the Winograd 3x3 forward of the blocks with >= 8 192 pixels runs the bf16x6 kernel
 *                                (rdm_conv3x3_wino_fwd_x6: float32-equivalent accuracy on the bf16 matrix pipe) instead of the f32 MFMA one.
 *   RDM_NET_OPT_FUSE_STATS3      (default 1) training forward of the few-pixel blocks (dense_e4, decoder d_1): the K-split 3x3 convolution of a layer takes the
 *                                channel statistics of its 48 outputs in the same launch - the last split of a pixel tile to arrive reduces the finished tile -
 *                                instead of a separate column reduction on the dependent chain (0: the separate pass; the values differ only in summation order).
 *   RDM_NET_OPT_GEMM_BF16        mixed-precision arithmetic (the reference's default --precision 16, train.py:11,57-58): every launch that
 *                                RDM_NET_OPT_SPLIT_BWD / RDM_NET_OPT_SPLIT_FWD route to the split kernels rounds its operands to bf16 instead (ONE bf16
 *                                MFMA per product, float32 accumulation).  value 1 = forward and gradient GEMMs, 2 = forward only, 3 = gradient GEMMs
 *                                only, 0 = off.  Activations, weights, BatchNorm statistics and the optimiser stay float32 in memory; with the gradient
 *                                GEMMs in this mode the per-layer dZ -> dY scratch tensor is kept as bf16.  Not the parity configuration: tolerance
 *                                stated in tests/test_gpu_mixed.py. */
typedef enum rdm_net_option { RDM_NET_OPT_PACKED_3X3 = 1, RDM_NET_OPT_GRADS_PREZEROED = 2, RDM_NET_OPT_DIRECT_3X3 = 3, RDM_NET_OPT_DETERMINISTIC = 4,
                              RDM_NET_OPT_JOIN_PER_SEGMENT = 5, RDM_NET_OPT_SPLIT_BWD = 6, RDM_NET_OPT_SPLIT_FWD = 7, RDM_NET_OPT_GEMM_BF16 = 8, RDM_NET_OPT_DEFER_NORM1 = 9, RDM_NET_OPT_PREPACK = 10, RDM_NET_OPT_SPLIT_ROWS = 11, RDM_NET_OPT_WINO_X6 = 12, RDM_NET_OPT_FUSE_STATS3 = 13 } rdm_net_option;
int rdm_net_set_option(rdm_net* net, int32_t option, int32_t value);

/* x: (B,3,H,W) float32 NCHW; logits: (B,180,h,w) float32 NCHW (conv2 output, RDM_Net.py:159). */
int rdm_net_forward(rdm_net* net, const float* x_nchw, void* const* tensors, void* workspace, size_t workspace_bytes,
                    float* logits_nchw, int32_t training, rdm_stream_t stream);

/* The encoder output (trans_e4, RDM_Net.py:94: what every decoder consumes, :103-125) of the last rdm_net_forward on `workspace`,
 * copied out as (B,1056,h,w) float32 NCHW - the input of the relative decoders d_6..d_10 (:57-61,106-125). */
int rdm_net_encoder_output(const rdm_net* net, const void* workspace, size_t workspace_bytes, float* out_nchw, rdm_stream_t stream);

/* Reduced-precision forward (BASELINE config 2, "batch=8 forward-only bf16"; the reference's default precision is mixed,
 * train.py:11,57-58): bf16 weights and activations in HBM, v_mfma_f32_16x16x32_bf16 with f32 accumulation, eval-mode BatchNorm
 * (running statistics) applied in f32 inside the conv staging, f32 bias and f32 logits.  Inference only: no statistics update, no
 * saved activations, no backward.
 *   rdm_net_bf16_prepare      converts every conv weight to bf16 (3x3: [tap][out][in]) and folds every BatchNorm to (scale, shift)
 *                             into `wbuf` (rdm_net_bf16_weight_bytes, 256-byte aligned, caller-owned); call again after a weight update.
 *   rdm_net_forward_bf16      x (B,3,H,W) f32 NCHW -> logits (B,180,h,w) f32 NCHW; `tensors` supplies the two f32 bias vectors.
 *   rdm_net_bf16_forward_bytes  algorithmic HBM bytes of one such forward (each activation written once and read once per consumer) */
size_t rdm_net_bf16_weight_bytes(const rdm_net* net);
size_t rdm_net_bf16_workspace_bytes(const rdm_net* net);
double rdm_net_bf16_forward_bytes(const rdm_net* net);
int rdm_net_bf16_prepare(rdm_net* net, void* const* tensors, void* wbuf, size_t wbuf_bytes, rdm_stream_t stream);
int rdm_net_forward_bf16(rdm_net* net, const float* x_nchw, void* const* tensors, const void* wbuf, size_t wbuf_bytes, void* workspace,
                         size_t workspace_bytes, float* logits_nchw, rdm_stream_t stream);

/* Backward in up to 4 segments so the caller can start reducing a segment's gradients (RCCL)
 * while the next one computes: 0 = decoder d_1, 1 = dense_e4+trans_e4, 2 = dense_e3+trans_e3,
 * 3 = dense_e2+trans_e2+conv_e1.  Segments must be run in order 0..3. */
int rdm_net_backward(rdm_net* net, const float* dlogits_nchw, void* const* tensors, void* const* grads, void* workspace,
                     size_t workspace_bytes, int32_t first_segment, int32_t last_segment, rdm_stream_t stream);
/* (first,last) tensor index range whose gradients segment `seg` produces */
int rdm_net_segment_range(int32_t seg, int32_t* first_tensor, int32_t* last_tensor);
/* The same backward at bucket granularity: the segments cut into STAGES of consecutive dense layers holding ~25 MB of gradients each
 * (the bucket size of the DistributedDataParallel the reference trains under, train.py:55), so the caller can start an all-reduce
 * every few layers and the last exchange is small.  Stages must be run in order 0 .. rdm_net_num_backward_stages()-1; the tensors
 * of stage k form the contiguous index range rdm_net_backward_stage_range gives (stages run from the LAST registered tensor backwards). */
int rdm_net_num_backward_stages(void);
int rdm_net_backward_stage_range(int32_t stage, int32_t* first_tensor, int32_t* last_tensor);
int rdm_net_backward_stage(rdm_net* net, const float* dlogits_nchw, void* const* tensors, void* const* grads, void* workspace,
                           size_t workspace_bytes, int32_t stage, rdm_stream_t stream);
/* test/debug access to the workspace layout: byte offset + float count of a named internal buffer
 * ("blk0".."blk3" block activations NHWC, "G0".."G3" their gradients, "logits", "Y<b>_<i>" bottlenecks ...) */
int rdm_net_buffer(const rdm_net* net, const char* name, int64_t* offset_bytes, int64_t* numel);
/* forward conv FLOPs (2*MAC) of one forward pass at this geometry, for roofline accounting */
double rdm_net_forward_flops(const rdm_net* net);
double rdm_net_backward_flops(const rdm_net* net);

/* Layout changes at the edges of the NHWC conv stack (the reference's tensors are NCHW: RDM_Net.py:70-135): src (B,C,HW) -> dst (B,HW,dst_ld) and
 * back.  nchw_to_nhwc writes EVERY element of every dst row: channels [0, C) from src, channels [C, dst_ld) as zeros - do not aim it at a slice
 * of a wider buffer whose other channels must survive (convert into a scratch of ld = C for that).  nhwc_to_nchw reads the channel prefix
 * [0, C) of rows of stride src_ld.  Used by the relative decoders d_6..d_10, which enter the stack with the encoder output. */
int rdm_layout_nchw_to_nhwc_f32(const float* src, float* dst, int32_t dst_ld, int32_t batch, int32_t channels, int32_t hw, rdm_stream_t stream);
int rdm_layout_nhwc_to_nchw_f32(const float* src, int32_t src_ld, float* dst, int32_t batch, int32_t channels, int32_t hw, rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DORN ordinal head + ordinal loss.  RDM_Net.py:313-345 (DornOrdinalRegression), loss.py:8-59.
 * ------------------------------------------------------------------------------------------ */
/* logits (B,2K,H,W) f32 NCHW -> ord (B,K,H,W) f64, decode (B,1,H,W) i64 */
int rdm_dorn_fwd(const float* logits, double* ord, int64_t* decode, int32_t batch, int32_t k, int32_t hw, rdm_stream_t stream);
/* dlogits = d/dlogits sum(ord * dord) */
int rdm_dorn_bwd(const float* logits, const double* dord, float* dlogits, int32_t batch, int32_t k, int32_t hw, rdm_stream_t stream);
/* loss (1 float, pre-zeroed) = -(sum_{k<=t} log P + sum_{k>t} log(1-P)) / (B*HW), float32 logs */
int rdm_ordinal_loss_fwd(const double* ord, const int32_t* target, float* loss, int32_t batch, int32_t k, int32_t hw, rdm_stream_t stream);
int rdm_ordinal_loss_bwd(const double* ord, const int32_t* target, const float* dloss, double* dord, int32_t batch, int32_t k, int32_t hw, rdm_stream_t stream);
/* utils.py:195-211 depth2label_sid on float64 depth: int32 labels.  A non-positive depth (a bicubic-resized target overshoots below zero next
 * to an invalid pixel) gives NaN before the reference's `.int()`, whose result depends on the DEVICE the reference ran on:
 *   RDM_SID_NAN_CPU   0x80000000 (x86 "integer indefinite") - what the parity fixtures, generated on the CPU, pin; the default
 *   RDM_SID_NAN_CUDA  0 - what the reference's production path (is_cuda = True, utils.py:205-211) computes on its GPU */
typedef enum rdm_sid_nan { RDM_SID_NAN_CPU = 0, RDM_SID_NAN_CUDA = 1 } rdm_sid_nan;
int rdm_depth2label_sid(const double* depth, int32_t* label, int64_t n, rdm_stream_t stream);      /* = _ex(..., RDM_SID_NAN_CPU, ...) */
int rdm_depth2label_sid_ex(const double* depth, int32_t* label, int64_t n, int32_t nan_semantics, rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * network/computations.py post-processing (float64 unless noted)
 * ------------------------------------------------------------------------------------------ */
/* computations.py:308-311 resize: bicubic (A=-0.75, align_corners=False), (n,h,w) -> (n,oh,ow) */
int rdm_resize_bicubic_f64(const double* src, double* dst, int32_t n, int32_t h, int32_t w, int32_t oh, int32_t ow, rdm_stream_t stream);
/* computations.py:244-255 quick_gm fused with the division that every caller applies:
 * dst[b,i] = src[b,i] / exp(exponent * sum_i log src[b,i]);  gm_out (optional) receives the means */
int rdm_gm_normalize_f64(const double* src, double* dst, double* gm_out, int32_t batch, int32_t n, double exponent, rdm_stream_t stream);
/* computations.py:368-392 decompose_depth_map: dn (B,S,S), S = 2^n <= 128 -> pyramid packed
 * smallest-first per sample: [d_0 (1x1) | F_1 (2x2) | ... | F_n (SxS)], (4^(n+1)-1)/3 doubles;
 * level k starts at (4^k-1)/3.  (relative_map=True callers simply ignore slot 0.) */
int rdm_decompose_f64(const double* dn, double* levels, int32_t batch, int32_t n, rdm_stream_t stream);
/* computations.py:423-484 + :512-528 with one candidate per level (the live graph, RDM_Net.py:126-133):
 * yhat_k = float32(log levels_k) * w[k], packed like `levels` (n_levels = n+1), float32 */
int rdm_fine_detail_pred_f32(const double* levels, const float* w, float* yhat, int32_t batch, int32_t n_levels, rdm_stream_t stream);
/* dw[k] = sum dyhat_k * float32(log levels_k)   (gradient of the 4 Weights scalars, RDM_Net.py:443-491) */
int rdm_fine_detail_pred_bwd(const double* levels, const float* dyhat, float* dw, int32_t batch, int32_t n_levels, rdm_stream_t stream);
/* computations.py:512-528 make_pred with several candidates per level (the relative decoders add rows to the matrix of a level,
 * RDM_Net.py:126-133): out[b][m] = sum_k float32(a[b][k][m]) * w[k], a (B,K,M) float64 (the log matrix), K <= 8; and the gradient of
 * the weights dw[k] = sum_{b,m} dout[b][m] * float32(a[b][k][m]). */
int rdm_candidates_matvec_f32(const double* a, const float* w, float* out, int32_t batch, int32_t k, int64_t m, rdm_stream_t stream);
int rdm_candidates_matvec_bwd(const double* a, const float* dout, float* dw, int32_t batch, int32_t k, int64_t m, rdm_stream_t stream);

/* computations.py:394-421 recombination: out (B,2^n_out,2^n_out) f64 = sum_{k>=first_level} nearest_up(yhat_k),
 * first_level = 0 when the list starts with d_0 (1x1), 1 for relative-only lists */
int rdm_recombine_f64(const float* yhat, double* out, int32_t batch, int32_t n_levels, int32_t n_out, int32_t first_level, rdm_stream_t stream);
int rdm_recombine_bwd(const double* dout, float* dyhat, int32_t batch, int32_t n_levels, int32_t n_out, int32_t first_level, rdm_stream_t stream);

/* metrics.py:48-128 (validation metrics) in one pass over the pixels with target > 0 (pred clamped to 1e-7):
 * out10 = [count, #delta1, #delta2, #delta3, sum sq err, sum abs err, sum |log10 p - log10 t|,
 *          sum |p-t|/t, sum (p-t)^2/t, sum sqrt((p-t)^2/t)]; the host divides by count (after an all-reduce under DP) */
int rdm_depth_metrics_f64(const double* pred, const double* target, int64_t n, double* out10, rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NYU input pipeline: dataloaders/nyu_dataloader.py:240-272 training_preprocess and
 * :274-287 validation_preprocess for a whole batch on the GPU, bit-exact with the Pillow arithmetic the
 * reference reaches through torchvision's PIL transforms.  The random draws of the reference are inputs:
 * one rdm_nyu_aug per sample, in DEVICE memory (md_rdm_amd/dataloaders/nyu.py fills it).
 *   rgb (B,in_h,in_w,3) uint8, depth (B,in_h,in_w) float32  ->  x (B,3,out_h,out_w), y (B,1,out_h,out_w) float32
 *   (resized_h, resized_w) = torchvision Resize(resize) of (in_h, in_w); validation = identity augmentation
 *   (depth_div 1, rot {65536,0,32768,0,65536,32768}, h2/w2 = resized size, ops -1, crop2 = {0,0,resized_h,resized_w}).
 * ------------------------------------------------------------------------------------------ */
typedef struct rdm_nyu_aug {
  float depth_div;      /* s: depth / s (:241-242) */
  int32_t rot[6];       /* Image.rotate(angle) as Geometry.c affine_fixed 16.16 coefficients a0,a1,a2,a3,a4,a5 (:252-254) */
  int32_t h2, w2;       /* size after Resize(int(resize * s)) (:256-258) */
  int32_t top, left;    /* CenterCrop origin in the (h2, w2) image (:260-262) */
  int32_t flip;         /* hflip (:264-266) */
  int32_t op[3];        /* ColorJitter order: 0 brightness, 1 contrast, 2 saturation, -1 none (:247) */
  float factor[3];      /* and the enhancement factors */
  int32_t crop2[4];     /* window {top, left, height, width} of the rotated image that the second Resize reads: the whole image for
                           train / val; test_preprocess (:289-307) = Resize(500), CenterCrop((480, 640)) as crop2, Resize(output_size) */
} rdm_nyu_aug;
size_t rdm_nyu_preprocess_workspace_bytes(int32_t batch, int32_t in_h, int32_t in_w, int32_t resized_h, int32_t resized_w, int32_t out_w);
int rdm_nyu_preprocess(const uint8_t* rgb, const float* depth, const rdm_nyu_aug* aug, int32_t batch, int32_t in_h, int32_t in_w, int32_t resized_h,
                       int32_t resized_w, int32_t out_h, int32_t out_w, float* x, float* y, void* workspace, size_t workspace_bytes,
                       rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Relative decoders (dormant in the reference graph, live as operators):
 * ratio grid + Lloyd quantisation (RDM_Net.py:244-311, computations.py:269-295) and rank-1 ALS
 * (computations.py:38-85,95-155,175-193), paging (:201-238).
 * ------------------------------------------------------------------------------------------ */
/* dense: R[b,i,j] = lloyd(d[b,i] * (1/d[b,j])), float32, S*S x S*S (sparse_comparison_v1) */
int rdm_ratio_grid_lloyd_dense(const float* d, float* R, int32_t batch, int32_t n, const double* quant40, const double* inv41, rdm_stream_t stream);
/* paged: dn (B,S,S) f32, dn_1 (B,S/2,S/2) f64 -> R (P,B,256,64) f64 per 16x16 page (P=(S/16)^2) */
int rdm_ratio_grid_lloyd_paged(const float* dn, const double* dn_1, double* R, int32_t batch, int32_t s, const double* quant40,
                               const double* inv41, int32_t quantize, rdm_stream_t stream);
/* rank-1 ALS on `groups` independent calls of `batch` matrices (rows x cols, f64 or f32 input):
 * p (groups,batch,rows) f32 = first-arg-min-rmse iterate / gm.  Two passes, no iterate history: all iterations record the squared error
 * (and the first 8 iterates); a winner beyond those is recomputed by re-running that group to k* (same operations, same bits). */
size_t rdm_als_workspace_bytes(int32_t groups, int32_t batch, int32_t rows, int32_t cols, int32_t limit);
int rdm_als_rank1(const void* R, int32_t r_is_f64, float* p_out, int32_t groups, int32_t batch, int32_t rows, int32_t cols,
                  int32_t limit, void* workspace, size_t workspace_bytes, rdm_stream_t stream);

/* The paged head in one call (RDM_Net.py:259-311 + computations.py:95-155 for every 16x16 page): rdm_ratio_grid_lloyd_paged fused INTO
 * the ALS load - each ALS thread forms its own 64-entry row of the quantised grid from dn and the 3x3 window of dn_1, so the float64
 * grid (134 MB at d_10, B = 16) is never written or read; p (P,B,256) float32 is bit-identical with rdm_ratio_grid_lloyd_paged ->
 * rdm_als_rank1(r_is_f64 = 1).  Workspace: rdm_als_workspace_bytes((s/16)^2, batch, 256, 64, limit). */
int rdm_als_rank1_paged(const float* dn, const double* dn_1, float* p_out, int32_t batch, int32_t s, const double* quant40, const double* inv41,
                        int32_t limit, void* workspace, size_t workspace_bytes, rdm_stream_t stream);

/* computations.py:201-216 split_matrix: (B,S,S) -> (P,B,page,page) row-major pages, P = (S/page)^2 */
int rdm_page_split_f32(const float* src, float* pages, int32_t batch, int32_t s, int32_t page, rdm_stream_t stream);
/* computations.py:218-238 reconstruct, bug-as-spec: out[b,y,x] = pages[y/page][b, y%page, x%page] */
int rdm_page_reconstruct_f32(const float* pages, float* out, int32_t batch, int32_t s, int32_t page, rdm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser: torch.optim.AdamW step of network/module.py:41 over flat buffers (one launch).
 * ------------------------------------------------------------------------------------------ */
int rdm_adamw_fused(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int32_t step, float grad_scale, rdm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RDM_HIP_H_ */
