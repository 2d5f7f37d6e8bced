// C-ABI plumbing: thread-local error string, argument validation and the op-level conv entry
// points declared in include/rdm_hip.h.  No torch types cross this boundary.
#include <stdarg.h>

#include "rdm_common.h"
#include "elementwise.h"

namespace rdm {
extern int g_variant;
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int geom_from_desc(const rdm_conv_desc* d, ConvGeom* g) {
  RDM_CHECK_ARG(d != nullptr, "conv desc is NULL");
  RDM_CHECK_ARG(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->in_c > 0 && d->out_c > 0, "conv desc: non-positive extent");
  RDM_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0, "conv desc: bad filter geometry");
  RDM_CHECK_ARG(d->in_ld >= d->in_c && d->out_ld >= d->out_c, "conv desc: pixel stride smaller than channel count");
  const int ho = (d->in_h + 2 * d->pad_h - d->kh) / d->stride_h + 1, wo = (d->in_w + 2 * d->pad_w - d->kw) / d->stride_w + 1;
  RDM_CHECK_ARG(ho > 0 && wo > 0, "conv desc: empty output");
  *g = ConvGeom{d->batch, d->in_h, d->in_w, ho, wo, d->kh, d->kw, d->stride_h, d->stride_w, d->pad_h, d->pad_w, 1};
  return 0;
}
}  // namespace rdm

using namespace rdm;

extern "C" {

const char* rdm_last_error_string(void) { return g_err; }
int rdm_version(void) { return 100; }

void rdm_debug_variant(int32_t v) { rdm::g_variant = v; }
void rdm_profile_enable(int32_t on) { profile_enable(on != 0); }
int rdm_profile_read(double* conv_ms_sum, double* conv_ms_union, double* conv_flops, int32_t* launches) {
  int n = 0;
  int rc = profile_read(conv_ms_sum, conv_ms_union, conv_flops, &n);
  if (launches) *launches = n;
  return rc;
}

int rdm_profile_kind(int32_t kind, const char** name, double* ms_sum, double* flops, int32_t* launches) {
  int n = 0;
  int rc = profile_kind(kind, name, ms_sum, flops, &n);
  if (launches) *launches = n;
  return rc;
}

size_t rdm_nyu_preprocess_workspace_bytes(int32_t batch, int32_t in_h, int32_t in_w, int32_t resized_h, int32_t resized_w, int32_t out_w) {
  return nyu_preprocess_workspace_bytes(batch, in_h, in_w, resized_h, resized_w, out_w);
}
int rdm_nyu_preprocess(const uint8_t* rgb, const float* depth, const rdm_nyu_aug* aug, int32_t batch, int32_t in_h, int32_t in_w, int32_t resized_h,
                       int32_t resized_w, int32_t out_h, int32_t out_w, float* x, float* y, void* workspace, size_t workspace_bytes,
                       rdm_stream_t stream) {
  return launch_nyu_preprocess(rgb, depth, aug, batch, in_h, in_w, resized_h, resized_w, out_h, out_w, x, y, workspace, workspace_bytes,
                               static_cast<hipStream_t>(stream));
}

int rdm_conv2d_fwd(const rdm_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale, const float* bn_shift,
                   float* y, double* stat_sum, double* stat_sq, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(x && w && y, "conv2d_fwd: NULL operand");
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "conv2d_fwd: bn_scale and bn_shift go together");
  RDM_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "conv2d_fwd: stat_sum and stat_sq go together");
  RDM_CHECK_ARG(!(stat_sum && bias), "conv2d_fwd: statistics epilogue and bias are mutually exclusive");
  FwdArgs a{};
  a.g = g; a.A = x; a.lda = d->in_ld; a.C = d->in_c; a.a_scale = bn_scale; a.a_shift = bn_shift;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = y; a.ldc = d->out_ld; a.M = g.B * g.Ho * g.Wo; a.N = d->out_c; a.bias = bias;
  a.stat0 = stat_sum; a.stat1 = stat_sq;
  rc = launch_conv_fwd(a, false, stat_sum ? EPI_STORE_STATS : EPI_STORE, stream);
  return rc < 0 ? rc : RDM_OK;
}

int rdm_conv2d_dgrad(const rdm_conv_desc* d, const float* dy, const float* w, float* dx, int32_t dx_ld, const float* mask_x, int32_t mask_ld,
                     const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && w && dx, "conv2d_dgrad: NULL operand");
  RDM_CHECK_ARG(d->stride_h == 1 && d->stride_w == 1, "conv2d_dgrad: stride 1 only");
  RDM_CHECK_ARG(d->out_c % 16 == 0, "conv2d_dgrad: out_c (%d) must be a multiple of 16 (pad the weight rows)", d->out_c);
  RDM_CHECK_ARG(!mask_x || (mask_scale && mask_shift && stat_a && stat_b), "conv2d_dgrad: mask needs scale, shift and both statistics");
  // the gradient wrt the input lives on the input grid; gather from the output grid
  ConvGeom gd{d->batch, g.Ho, g.Wo, d->in_h, d->in_w, d->kh, d->kw, 1, 1, d->pad_h, d->pad_w, -1};
  FwdArgs a{};
  a.g = gd; a.A = dy; a.lda = d->out_ld; a.C = d->out_c;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = dx; a.ldc = dx_ld; a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c;
  a.X = mask_x; a.ldx = mask_ld; a.x_scale = mask_scale; a.x_shift = mask_shift; a.stat0 = stat_a; a.stat1 = stat_b;
  rc = launch_conv_fwd(a, true, mask_x ? EPI_MASK_STATS : EPI_STORE, stream);
  return rc < 0 ? rc : RDM_OK;
}

int rdm_conv2d_wgrad(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw,
                     rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && x && dw, "conv2d_wgrad: NULL operand");
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "conv2d_wgrad: bn_scale and bn_shift go together");
  WgradArgs a{};
  a.g = g; a.G = dy; a.ldg = d->out_ld; a.N = d->out_c;
  a.Xs = x; a.ldx = d->in_ld; a.C = d->in_c; a.x_scale = bn_scale; a.x_shift = bn_shift;
  a.dW = dw; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  return launch_conv_wgrad(a, stream);
}

int rdm_pack_conv_weight(const float* w, float* wp, int32_t out_c, int32_t in_c, int32_t kh, int32_t kw, int32_t out_c_padded, rdm_stream_t stream) {
  RDM_CHECK_ARG(w && wp && out_c > 0 && in_c > 0 && kh > 0 && kw > 0 && out_c_padded >= out_c, "pack_conv_weight: bad argument");
  return launch_pack_w(w, wp, out_c, in_c, kh * kw, out_c_padded, stream);
}
int rdm_unpack_conv_weight(const float* wp, float* w, int32_t out_c, int32_t in_c, int32_t kh, int32_t kw, int32_t out_c_padded, rdm_stream_t stream) {
  RDM_CHECK_ARG(w && wp && out_c > 0 && in_c > 0 && kh > 0 && kw > 0 && out_c_padded >= out_c, "unpack_conv_weight: bad argument");
  return launch_unpack_w(wp, w, out_c, in_c, kh * kw, out_c_padded, stream);
}

int rdm_adamw_fused(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step, float grad_scale, rdm_stream_t stream) {
  RDM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "adamw_fused: bad argument");
  RDM_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "adamw_fused: buffers must be 16-byte aligned");
  if (n == 0) return RDM_OK;
  return launch_adamw(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

}  // extern "C"
