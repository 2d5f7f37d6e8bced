// internal launcher declarations (elementwise.hip)
#pragma once
#include <algorithm>
#include <hip/hip_runtime.h>
namespace rdm {
int launch_colstats(const float* V, int ldv, int M, int C, double* sum, double* sq, hipStream_t s);
int launch_mask_stats(float* V, int ldv, const float* X, int ldx, const float* xs, const float* xt, int M, int C, double* s0, double* s1, hipStream_t s);
int launch_bn_finalize(const double* sum, const double* sq, double count, const float* gamma, const float* beta, float* rm, float* rv,
                       long long* nbt, float* scale, float* shift, float* save_mean, float* save_rstd, int C, int training, hipStream_t s);
// batched training-mode finalisation (k_bn_finalize_batch): entry = channels [0, C) of one BatchNorm; out = [scale | shift | mean | rstd] with
// row stride Cout (the BatchNorm's full width when only a channel range is finalised)
struct BnBatchEntry { const double* sum; const double* sq; const float* gamma; const float* beta; float* rm; float* rv; long long* nbt; float* out; double count; int C; int Cout; };
constexpr int BN_BATCH = 24;
struct BnBatch { BnBatchEntry e[BN_BATCH]; };
int launch_bn_finalize_batch(const BnBatch& bb, int count, int max_c, hipStream_t s);
int launch_bn_bwd_coeffs(const double* s0, const double* s1, double count, const float* gamma, const float* mean, const float* rstd, float* A,
                         float* Bc, float* Cc, float* dgamma, float* dbeta, int C, int training, hipStream_t s);
int launch_bn_bwd_defer(float* G, int ldg, const float* x, int ldx, const double* s0, const double* s1, double count, const float* gamma, const float* mean,
                        const float* rstd, float* dgamma, float* dbeta, const float* b_in, const float* c_in, float* b_out, float* c_out, int M, int C,
                        int slice_c0, int slice_n, int training, hipStream_t s);
int launch_bn_bwd_apply(float* dst, int ldd, const float* dz, int ldz, const float* x, int ldx, const double* s0, const double* s1, double count,
                        const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta, int M, int C, bool accumulate,
                        int training, hipStream_t s, bool bf16_rows = false, bool split_rows = false);
// dst (split rows, xsplit_dev.h; row stride ld_dst in floats) = split(ReLU(scale * src + shift)) - or split(src) without scale / shift
int launch_split_rows(const float* src, int ld_src, const float* scale, const float* shift, void* dst, int ld_dst, long M, int C, hipStream_t s);
int launch_zero_rows(float* p, long rows, long row_floats, long ld, hipStream_t s);
int launch_trans_pool(const float* X, int ldx, const float* sc, const float* sh, float* P, int B, int H, int W, int C, hipStream_t s);
int launch_trans_pool_bwd_reduce(const float* dP, const float* X, int ldx, const float* sc, const float* sh, int B, int H, int W, int C,
                                 double* s0, double* s1, hipStream_t s);
int launch_trans_pool_bwd_apply(const float* dP, const float* X, int ldx, const float* sc, const float* sh, const float* A, const float* Bc,
                                const float* Cc, float* G, int ldg, int B, int H, int W, int C, hipStream_t s);
int launch_im2col_stem(const float* x, float* patches, int B, int H, int W, hipStream_t s);
int launch_maxpool3s2(const float* X, float* Y, int ldy, unsigned char* arg, int B, int H, int W, int C, hipStream_t s);
int launch_maxpool3s2_bwd(const float* Gy, int ldg, const unsigned char* arg, float* Gx, int B, int H, int W, int C, hipStream_t s);
int launch_pack_w(const float* w, float* wp, int O, int I, int T, int Opad, hipStream_t s);
int launch_unpack_w(const float* wp, float* w, int O, int I, int T, int Opad, hipStream_t s);
int launch_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int C, int HW, hipStream_t s);
int launch_nchw_to_nhwc(const float* src, float* dst, int ld, int B, int C, int HW, hipStream_t s);
int launch_f64_to_f32(const double* src, float* dst, int n, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                 float gscale, hipStream_t s);
}  // namespace rdm
