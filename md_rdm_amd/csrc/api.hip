// C-ABI plumbing: thread-local error string, argument validation and the op-level conv entry
// points declared in include/rdm_hip.h.  No torch types cross this boundary.
#include <stdarg.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rdm_common.h"
#ifdef RDM_DEV_VARIANTS
#include "../../include/rdm_dev.h"
#endif
#include "elementwise.h"
#include "bf16.h"
#include "wino.h"
#include "xsplit.h"

namespace rdm {
long long g_launches = 0;
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

thread_local bool t_deterministic = false;
bool g_census_on = false;
namespace {
std::mutex g_census_mu;
std::map<std::string, long long> g_census;
std::vector<std::pair<std::string, long long>> g_census_snapshot;      // stable storage for the strings rdm_census_entry hands out
}  // namespace
void census_hit(const char* fmt, ...) {
  char buf[192];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lk(g_census_mu);
  ++g_census[buf];
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

#ifdef RDM_DEV_VARIANTS
void rdm_debug_variant(int32_t v) { rdm::g_variant = v; }      // include/rdm_dev.h: development builds only
#endif
int64_t rdm_launch_count(void) { return rdm::g_launches; }
void rdm_census_enable(int32_t on) { rdm::g_census_on = on != 0; }
void rdm_census_reset(void) {
  std::lock_guard<std::mutex> lk(g_census_mu);
  g_census.clear();
}
int32_t rdm_census_count(void) {
  std::lock_guard<std::mutex> lk(g_census_mu);
  g_census_snapshot.assign(g_census.begin(), g_census.end());
  return (int32_t)g_census_snapshot.size();
}
int rdm_census_entry(int32_t i, const char** name, int64_t* launches) {
  std::lock_guard<std::mutex> lk(g_census_mu);
  RDM_CHECK_ARG(i >= 0 && i < (int)g_census_snapshot.size() && name && launches, "census_entry: index %d outside the last rdm_census_count()", (int)i);
  *name = g_census_snapshot[i].first.c_str();
  *launches = g_census_snapshot[i].second;
  return RDM_OK;
}
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

/* algorithmic HBM bytes (operands read once + result written once) of the launches of `kind` in the last rdm_profile_read(); 0 for the f32 kinds */
double rdm_profile_kind_bytes(int32_t kind) { return profile_kind_bytes(kind); }

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
  return rdm_conv2d_fwd_ex(d, x, w, bias, bn_scale, bn_shift, y, stat_sum, stat_sq, 0, stream);
}

int rdm_conv2d_fwd_ex(const rdm_conv_desc* d, const float* x, const float* w, const float* bias, const float* bn_scale, const float* bn_shift,
                      float* y, double* stat_sum, double* stat_sq, int32_t split_k, rdm_stream_t stream) {
  RDM_CHECK_ARG(split_k >= 0 && split_k <= 128, "conv2d_fwd: split_k (%d) must be 0 (auto) .. 128", (int)split_k);
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
  a.split_k = split_k;
  rc = launch_conv_fwd(a, false, stat_sum ? EPI_STORE_STATS : EPI_STORE, stream);
  return rc < 0 ? rc : RDM_OK;
}

int rdm_conv2d_fwd_bnsums(const rdm_conv_desc* d, const float* x, const float* w, const double* bn_sum, const double* bn_sumsq, double bn_count,
                          const float* bn_gamma, const float* bn_beta, float* y, double* stat_sum, double* stat_sq, int32_t split_k, rdm_stream_t stream) {
  RDM_CHECK_ARG(split_k >= 0 && split_k <= 128, "conv2d_fwd_bnsums: split_k (%d) must be 0 (auto) .. 128", (int)split_k);
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(x && w && y && bn_sum && bn_sumsq && bn_gamma && bn_beta && bn_count >= 1, "conv2d_fwd_bnsums: NULL operand");
  RDM_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "conv2d_fwd_bnsums: stat_sum and stat_sq go together");
  FwdArgs a{};
  a.g = g; a.A = x; a.lda = d->in_ld; a.C = d->in_c;
  a.a_sum = bn_sum; a.a_sq = bn_sumsq; a.a_gamma = bn_gamma; a.a_beta = bn_beta; a.a_count = bn_count;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = y; a.ldc = d->out_ld; a.M = g.B * g.Ho * g.Wo; a.N = d->out_c;
  a.stat0 = stat_sum; a.stat1 = stat_sq;
  a.split_k = split_k;
  rc = launch_conv_fwd(a, false, stat_sum ? EPI_STORE_STATS : EPI_STORE, stream);
  return rc < 0 ? rc : RDM_OK;
}

int rdm_conv3x3_fwd_bnsums_acc(const rdm_conv_desc* d, const float* x, const float* w, const double* bn_sum, const double* bn_sumsq, double bn_count,
                               const float* bn_gamma, const float* bn_beta, float* y, double* stat_sum, double* stat_sq, uint32_t* tile_tickets, int32_t split_k,
                               rdm_stream_t stream) {
  RDM_CHECK_ARG(split_k >= 0 && split_k <= 128, "conv3x3_fwd_bnsums_acc: split_k (%d) must be 0 (auto) .. 128", (int)split_k);
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(x && w && y && bn_sum && bn_sumsq && bn_gamma && bn_beta && bn_count >= 1 && stat_sum && stat_sq && tile_tickets, "conv3x3_fwd_bnsums_acc: NULL operand");
  RDM_CHECK_ARG(d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 && d->out_c <= 48 && ((uintptr_t)tile_tickets & 3) == 0,
                "conv3x3_fwd_bnsums_acc: a 3x3 / stride 1 / pad 1 convolution with <= 48 outputs");
  FwdArgs a{};
  a.g = g; a.A = x; a.lda = d->in_ld; a.C = d->in_c;
  a.a_sum = bn_sum; a.a_sq = bn_sumsq; a.a_gamma = bn_gamma; a.a_beta = bn_beta; a.a_count = bn_count;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = y; a.ldc = d->out_ld; a.M = g.B * g.Ho * g.Wo; a.N = d->out_c;
  a.stat0 = stat_sum; a.stat1 = stat_sq;
  a.split_k = split_k; a.accumulate = 1; a.tickets = tile_tickets;
  rc = launch_conv_fwd(a, false, EPI_STORE, stream);
  return rc < 0 ? rc : RDM_OK;
}

int rdm_conv2d_dgrad(const rdm_conv_desc* d, const float* dy, const float* w, float* dx, int32_t dx_ld, const float* mask_x, int32_t mask_ld,
                     const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, rdm_stream_t stream) {
  return rdm_conv2d_dgrad_ex(d, dy, w, dx, dx_ld, mask_x, mask_ld, mask_scale, mask_shift, stat_a, stat_b, 0, stream);
}

int rdm_conv2d_dgrad_ex(const rdm_conv_desc* d, const float* dy, const float* w, float* dx, int32_t dx_ld, const float* mask_x, int32_t mask_ld,
                        const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, int32_t split_k, rdm_stream_t stream) {
  RDM_CHECK_ARG(split_k >= 0 && split_k <= 128, "conv2d_dgrad: split_k (%d) must be 0 (auto) .. 128", (int)split_k);
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
  a.split_k = split_k;
  rc = launch_conv_fwd(a, true, mask_x ? EPI_MASK_STATS : EPI_STORE, stream);
  return rc < 0 ? rc : RDM_OK;
}

int rdm_conv2d_wgrad(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw,
                     rdm_stream_t stream) {
  return rdm_conv2d_wgrad_ex(d, dy, x, bn_scale, bn_shift, dw, 0, stream);
}

int rdm_conv2d_wgrad_ex(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw,
                        int32_t split_k, rdm_stream_t stream) {
  RDM_CHECK_ARG(split_k >= 0 && split_k <= 128, "conv2d_wgrad: split_k (%d) must be 0 (auto) .. 128", (int)split_k);
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && x && dw, "conv2d_wgrad: NULL operand");
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "conv2d_wgrad: bn_scale and bn_shift go together");
  WgradArgs a{};
  a.g = g; a.G = dy; a.ldg = d->out_ld; a.N = d->out_c;
  a.Xs = x; a.ldx = d->in_ld; a.C = d->in_c; a.x_scale = bn_scale; a.x_shift = bn_shift;
  a.dW = dw; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.split_k = split_k;
  return launch_conv_wgrad(a, stream);
}

size_t rdm_conv3x3_dgrad_x3_workspace_bytes(int32_t in_c) { return in_c > 0 && in_c % 16 == 0 ? xs_dgrad3x3_workspace_bytes(in_c) : 0; }

int rdm_conv3x3_dgrad_x3(const rdm_conv_desc* d, const float* dy, const float* w, float* dx, int32_t dx_ld, const float* mask_x, int32_t mask_ld,
                         const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, void* workspace, size_t workspace_bytes,
                         int32_t products, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && w && dx, "conv3x3_dgrad_x3: NULL operand");
  RDM_CHECK_ARG(products == 0 || products == 1 || products == 3, "conv3x3_dgrad_x3: products (%d) must be 0 / 3 (split precision) or 1 (bf16 operands)", (int)products);
  RDM_CHECK_ARG(!mask_x || (mask_scale && mask_shift && stat_a && stat_b), "conv3x3_dgrad_x3: mask needs scale, shift and both statistics");
  ConvGeom gd{d->batch, g.Ho, g.Wo, d->in_h, d->in_w, d->kh, d->kw, d->stride_h, d->stride_w, d->pad_h, d->pad_w, -1};
  FwdArgs a{};
  a.g = gd; a.A = dy; a.lda = d->out_ld; a.C = d->out_c;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = dx; a.ldc = dx_ld; a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c;
  a.X = mask_x; a.ldx = mask_ld; a.x_scale = mask_scale; a.x_shift = mask_shift; a.stat0 = stat_a; a.stat1 = stat_b;
  if (!xs_dgrad3x3_supported(a)) { set_error("conv3x3_dgrad_x3: no split-precision kernel for this convolution (3x3 / stride 1 / pad 1, out_c = 48, in_c a multiple of 48, rows <= ~330 pixels)"); return RDM_ERR_UNSUPPORTED; }
  return launch_xs_dgrad3x3(a, mask_x ? EPI_MASK_STATS : EPI_STORE, workspace, workspace_bytes, stream, products == 1 ? 1 : 3);
}

size_t rdm_conv1x1_fwd_x6_workspace_bytes(int32_t in_c, int32_t out_c) { return out_c > 0 && in_c > 0 ? xs_fwd1x1_workspace_bytes(in_c, out_c) : 0; }

int rdm_conv1x1_fwd_x6(const rdm_conv_desc* d, const float* x, const float* w, const float* bn_scale, const float* bn_shift, float* y, double* stat_sum,
                       double* stat_sq, void* workspace, size_t workspace_bytes, int32_t products, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(x && w && y, "conv1x1_fwd_x6: NULL operand");
  RDM_CHECK_ARG(products == 0 || products == 1 || products == 6, "conv1x1_fwd_x6: products (%d) must be 0 / 6 (three-way split) or 1 (bf16 operands)", (int)products);
  RDM_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "conv1x1_fwd_x6: both statistics or none");
  FwdArgs a{};
  a.g = g; a.A = x; a.lda = d->in_ld; a.C = d->in_c; a.a_scale = bn_scale; a.a_shift = bn_shift;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = y; a.ldc = d->out_ld; a.M = d->batch * g.Ho * g.Wo; a.N = d->out_c;
  a.stat0 = stat_sum; a.stat1 = stat_sq;
  if (!xs_fwd1x1_supported(a)) { set_error("conv1x1_fwd_x6: no split-precision kernel for this convolution (1x1 / stride 1, out_c a multiple of 16, in_c of 4)"); return RDM_ERR_UNSUPPORTED; }
  return launch_xs_fwd1x1(a, stat_sum ? EPI_STORE_STATS : EPI_STORE, workspace, workspace_bytes, stream, products == 1 ? 1 : 6);
}

size_t rdm_conv1x1_dgrad_x3_workspace_bytes(int32_t out_c, int32_t in_c) { return out_c > 0 && in_c > 0 ? xs_dgrad1x1_workspace_bytes(out_c, in_c) : 0; }

int rdm_conv1x1_dgrad_x3(const rdm_conv_desc* d, const float* dy, const float* w, float* dx, int32_t dx_ld, const float* mask_x, int32_t mask_ld,
                         const float* mask_scale, const float* mask_shift, double* stat_a, double* stat_b, void* workspace, size_t workspace_bytes,
                         int32_t products, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && w && dx, "conv1x1_dgrad_x3: NULL operand");
  const int dy_rows = (products & RDM_X3_DY_SPLIT_ROWS) != 0, acc_scaled = (products & RDM_X3_ACC_SCALED) != 0;
  products &= ~(RDM_X3_DY_SPLIT_ROWS | RDM_X3_ACC_SCALED);
  RDM_CHECK_ARG(products == 0 || products == 1 || products == 3, "conv1x1_dgrad_x3: products (%d) must be 0 / 3 (split precision) or 1 (bf16 operands)", (int)products);
  RDM_CHECK_ARG(!acc_scaled || mask_x, "conv1x1_dgrad_x3: the scaled accumulating epilogue belongs to the masked form");
  RDM_CHECK_ARG(!dy_rows || products != 1, "conv1x1_dgrad_x3: split rows are an operand of the split arithmetic (products 0 / 3)");
  RDM_CHECK_ARG(!mask_x || (mask_scale && mask_shift && stat_a && stat_b), "conv1x1_dgrad_x3: mask needs scale, shift and both statistics");
  ConvGeom gd{d->batch, g.Ho, g.Wo, d->in_h, d->in_w, d->kh, d->kw, d->stride_h, d->stride_w, d->pad_h, d->pad_w, -1};
  FwdArgs a{};
  a.g = gd; a.A = dy; a.lda = d->out_ld; a.C = d->out_c;
  a.Wt = w; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.out = dx; a.ldc = dx_ld; a.M = d->batch * d->in_h * d->in_w; a.N = d->in_c;
  a.X = mask_x; a.ldx = mask_ld; a.x_scale = mask_scale; a.x_shift = mask_shift; a.stat0 = stat_a; a.stat1 = stat_b;
  a.a_split = dy_rows;
  a.acc_scaled = acc_scaled;
  if (!xs_dgrad1x1_supported(a)) { set_error("conv1x1_dgrad_x3: no split-precision kernel for this convolution (1x1 / stride 1, in_c a multiple of 16 and <= 2304)"); return RDM_ERR_UNSUPPORTED; }
  return launch_xs_dgrad1x1(a, mask_x ? EPI_MASK_STATS : EPI_STORE, workspace, workspace_bytes, stream, products == 1 ? 1 : 3);
}

int rdm_conv2d_wgrad_x3(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw,
                        int32_t split_k, int32_t products, rdm_stream_t stream) {
  RDM_CHECK_ARG(split_k >= 0 && split_k <= 128, "conv2d_wgrad_x3: split_k (%d) must be 0 (auto) .. 128", (int)split_k);
  const int dy_rows = (products & RDM_X3_DY_SPLIT_ROWS) != 0, x_rows = (products & RDM_X3_X_SPLIT_ROWS) != 0, dy_frame = (products & RDM_X3_DY_FRAME_ROWS) != 0;
  products &= ~(RDM_X3_DY_SPLIT_ROWS | RDM_X3_X_SPLIT_ROWS | RDM_X3_DY_FRAME_ROWS);
  RDM_CHECK_ARG(!dy_frame || (products != 1 && d->kh == 3 && d->kw == 3), "conv2d_wgrad_x3: the frame image is the gradient operand of the 3x3 kernel in the split arithmetic (products 0 / 3)");
  RDM_CHECK_ARG(products == 0 || products == 1 || products == 3, "conv2d_wgrad_x3: products (%d) must be 0 / 3 (split precision) or 1 (bf16 operands)", (int)products);
  RDM_CHECK_ARG(!(dy_rows || x_rows) || (products != 1 && d->kh == 1 && d->kw == 1), "conv2d_wgrad_x3: split rows are operands of the 1x1 kernel in the split arithmetic (products 0 / 3)");
  RDM_CHECK_ARG(!x_rows || (bn_scale == nullptr && bn_shift == nullptr), "conv2d_wgrad_x3: split activation rows are already activated (rdm_split_rows_f32 applied BatchNorm + ReLU)");
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && x && dw, "conv2d_wgrad_x3: NULL operand");
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "conv2d_wgrad_x3: bn_scale and bn_shift go together");
  WgradArgs a{};
  a.g = g; a.G = dy; a.ldg = d->out_ld; a.N = d->out_c;
  a.Xs = x; a.ldx = d->in_ld; a.C = d->in_c; a.x_scale = bn_scale; a.x_shift = bn_shift;
  a.dW = dw; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.split_k = split_k;
  a.xsplit = products == 1 ? 1 : 3;
  a.g_split = dy_rows; a.x_split = x_rows; a.g_frame = dy_frame;
  if (d->kh == 1 && d->kw == 1) return launch_xs_wgrad1x1(a, stream);
  if (d->kh == 3 && d->kw == 3) return launch_xs_wgrad3x3(a, stream);
  set_error("conv2d_wgrad_x3: no split-precision kernel for a %dx%d convolution", d->kh, d->kw);
  return RDM_ERR_UNSUPPORTED;
}

size_t rdm_conv3x3_wino_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w, int32_t split_k) {
  if (channels <= 0 || channels % 16 || batch <= 0 || h <= 0 || w <= 0 || split_k < 0) return 0;
  const int T = batch * ((h + 1) / 2) * ((w + 1) / 2);
  const int split = split_k > 0 ? split_k : wino_pick_split(T, channels / 16);
  return wino_fwd_workspace_bytes(channels, (long)batch * h * w, split);
}

int rdm_conv3x3_wino_fwd(const rdm_conv_desc* d, const float* x, const float* w, const float* bn_scale, const float* bn_shift, float* y,
                         double* stat_sum, double* stat_sq, void* workspace, size_t workspace_bytes, int32_t split_k, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(x && w && y && workspace, "conv3x3_wino_fwd: NULL operand");
  RDM_CHECK_ARG(d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1, "conv3x3_wino_fwd: 3x3 / stride 1 / pad 1 only");
  RDM_CHECK_ARG(d->out_c <= 48 && d->in_c % 16 == 0, "conv3x3_wino_fwd: out_c (%d) <= 48 and in_c (%d) a multiple of 16", d->out_c, d->in_c);
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr) && (stat_sum == nullptr) == (stat_sq == nullptr), "conv3x3_wino_fwd: scale/shift and the two statistics go together");
  RDM_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && split_k >= 0, "conv3x3_wino_fwd: workspace must be 256-byte aligned, split_k >= 0");
  const long M = (long)d->batch * d->in_h * d->in_w;
  const size_t ub = (((size_t)16 * 48 * d->in_c * sizeof(float)) + 255) & ~(size_t)255;
  if (workspace_bytes < ub) { set_error("conv3x3_wino_fwd: workspace too small: %zu < %zu", workspace_bytes, ub); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  float* U = static_cast<float*>(workspace);
  if ((rc = launch_wino_weight(w, (long)d->out_c * d->in_c, d->in_c, d->out_c, d->in_c, U, stream))) return rc;
  WinoConv a{};
  a.A = x; a.lda = d->in_ld; a.C = d->in_c; a.a_scale = bn_scale; a.a_shift = bn_shift; a.U = U;
  a.out = y; a.ldc = d->out_ld; a.N = d->out_c; a.B = d->batch; a.H = d->in_h; a.W = d->in_w;
  a.split = split_k; a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + ub); a.partial_floats = (workspace_bytes - ub) / sizeof(float);
  if (a.partial_floats < (size_t)M * 48 * 2) { a.partial = nullptr; a.partial_floats = 0; }
  a.stat0 = stat_sum; a.stat1 = stat_sq;
  return launch_conv3x3_wino_fwd(a, stream);
}

size_t rdm_conv3x3_wino_x6_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w, int32_t split_k) {
  if (channels <= 0 || channels % 16 || batch <= 0 || h <= 0 || w <= 0 || split_k < 0) return 0;
  const int T = batch * ((h + 1) / 2) * ((w + 1) / 2);
  const int split = split_k > 0 ? split_k : wino_pick_split(T, channels / 16, true);
  return wino_fwd_workspace_bytes(channels, (long)batch * h * w, split, true);
}

int rdm_conv3x3_wino_fwd_x6(const rdm_conv_desc* d, const float* x, const float* w, const float* bn_scale, const float* bn_shift, float* y,
                            double* stat_sum, double* stat_sq, void* workspace, size_t workspace_bytes, int32_t split_k, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(x && w && y && workspace, "conv3x3_wino_fwd_x6: NULL operand");
  RDM_CHECK_ARG(d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1, "conv3x3_wino_fwd_x6: 3x3 / stride 1 / pad 1 only");
  RDM_CHECK_ARG(d->out_c <= 48 && d->in_c % 16 == 0, "conv3x3_wino_fwd_x6: out_c (%d) <= 48 and in_c (%d) a multiple of 16", d->out_c, d->in_c);
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr) && (stat_sum == nullptr) == (stat_sq == nullptr), "conv3x3_wino_fwd_x6: scale/shift and the two statistics go together");
  RDM_CHECK_ARG(((uintptr_t)workspace & 255) == 0 && split_k >= 0, "conv3x3_wino_fwd_x6: workspace must be 256-byte aligned, split_k >= 0");
  const long M = (long)d->batch * d->in_h * d->in_w;
  const size_t ub = (wino_u_bytes(d->in_c, true) + 255) & ~(size_t)255;
  if (workspace_bytes < ub) { set_error("conv3x3_wino_fwd_x6: workspace too small: %zu < %zu", workspace_bytes, ub); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  float* U = static_cast<float*>(workspace);
  if ((rc = launch_wino_weight(w, (long)d->out_c * d->in_c, d->in_c, d->out_c, d->in_c, U, stream, true))) return rc;
  WinoConv a{};
  a.A = x; a.lda = d->in_ld; a.C = d->in_c; a.a_scale = bn_scale; a.a_shift = bn_shift; a.U = U; a.x6 = 1;
  a.out = y; a.ldc = d->out_ld; a.N = d->out_c; a.B = d->batch; a.H = d->in_h; a.W = d->in_w;
  a.split = split_k; a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + ub); a.partial_floats = (workspace_bytes - ub) / sizeof(float);
  if (a.partial_floats < (size_t)M * 48 * 2) { a.partial = nullptr; a.partial_floats = 0; }
  a.stat0 = stat_sum; a.stat1 = stat_sq;
  return launch_conv3x3_wino_fwd(a, stream);
}

size_t rdm_conv3x3_wino_wgrad_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w) {
  if (channels <= 0 || batch <= 0 || h <= 0 || w <= 0) return 0;
  return (((wino_wgrad_vy_floats(batch, h, w) * sizeof(float)) + 255) & ~(size_t)255) + (((wino_wgrad_part_floats(batch, h, w, channels) * sizeof(float)) + 255) & ~(size_t)255);
}

int rdm_conv3x3_wino_wgrad(const rdm_conv_desc* d, const float* dy, const float* x, const float* bn_scale, const float* bn_shift, float* dw,
                           void* workspace, size_t workspace_bytes, rdm_stream_t stream) {
  ConvGeom g;
  int rc = geom_from_desc(d, &g);
  if (rc) return rc;
  RDM_CHECK_ARG(dy && x && dw && workspace, "conv3x3_wino_wgrad: NULL operand");
  RDM_CHECK_ARG(d->kh == 3 && d->kw == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1, "conv3x3_wino_wgrad: 3x3 / stride 1 / pad 1 only");
  RDM_CHECK_ARG((bn_scale == nullptr) == (bn_shift == nullptr), "conv3x3_wino_wgrad: bn_scale and bn_shift go together");
  RDM_CHECK_ARG(d->out_c >= 4 && d->out_c <= 48 && d->out_c % 4 == 0 && d->in_c >= 4 && d->in_c % 4 == 0, "conv3x3_wino_wgrad: out_c (%d) must be 4 .. 48 and, like in_c (%d), a multiple of 4", d->out_c, d->in_c);
  RDM_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "conv3x3_wino_wgrad: workspace must be 256-byte aligned");
  if (workspace_bytes < rdm_conv3x3_wino_wgrad_workspace_bytes(d->in_c, d->batch, d->in_h, d->in_w)) {
    set_error("conv3x3_wino_wgrad: workspace too small: %zu < %zu", workspace_bytes, rdm_conv3x3_wino_wgrad_workspace_bytes(d->in_c, d->batch, d->in_h, d->in_w));
    return RDM_ERR_WORKSPACE_TOO_SMALL;
  }
  WinoWgrad a{};
  a.G = dy; a.ldg = d->out_ld; a.N = d->out_c; a.A = x; a.lda = d->in_ld; a.C = d->in_c; a.a_scale = bn_scale; a.a_shift = bn_shift;
  a.dW = dw; a.wtap = (long)d->out_c * d->in_c; a.ldw = d->in_c;
  a.vy_floats = wino_wgrad_vy_floats(d->batch, d->in_h, d->in_w);
  a.Vy = static_cast<float*>(workspace);
  a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + (((a.vy_floats * sizeof(float)) + 255) & ~(size_t)255));
  a.part_floats = wino_wgrad_part_floats(d->batch, d->in_h, d->in_w, d->in_c);
  a.B = d->batch; a.H = d->in_h; a.W = d->in_w;
  return launch_conv3x3_wino_wgrad(a, stream);
}

int rdm_pack_conv_weight(const float* w, float* wp, int32_t out_c, int32_t in_c, int32_t kh, int32_t kw, int32_t out_c_padded, rdm_stream_t stream) {
  RDM_CHECK_ARG(w && wp && out_c > 0 && in_c > 0 && kh > 0 && kw > 0 && out_c_padded >= out_c, "pack_conv_weight: bad argument");
  return launch_pack_w(w, wp, out_c, in_c, kh * kw, out_c_padded, stream);
}
int rdm_unpack_conv_weight(const float* wp, float* w, int32_t out_c, int32_t in_c, int32_t kh, int32_t kw, int32_t out_c_padded, rdm_stream_t stream) {
  RDM_CHECK_ARG(w && wp && out_c > 0 && in_c > 0 && kh > 0 && kw > 0 && out_c_padded >= out_c, "unpack_conv_weight: bad argument");
  return launch_unpack_w(wp, w, out_c, in_c, kh * kw, out_c_padded, stream);
}

int rdm_gemm_bf16(const void* x, int32_t ldx, int32_t k, const float* scale, const float* shift, const void* w, int32_t ldw, const float* bias,
                  void* out, int32_t ldc, int32_t m, int32_t n, int32_t out_f32, void* workspace, size_t workspace_bytes, rdm_stream_t stream) {
  RDM_CHECK_ARG(x && w && out && m > 0 && n > 0 && k > 0 && ldx >= k && ldw >= k && ldc >= n, "gemm_bf16: bad argument");
  RDM_CHECK_ARG(!workspace || ((uintptr_t)workspace & 255) == 0, "gemm_bf16: workspace must be 256-byte aligned");
  GemmBf16Args a{};
  a.X = x; a.ldx = ldx; a.K = k; a.scale = scale; a.shift = shift; a.W = w; a.ldw = ldw; a.bias = bias; a.out = out; a.ldc = ldc; a.M = m; a.N = n;
  a.partial = static_cast<float*>(workspace); a.partial_floats = workspace ? workspace_bytes / sizeof(float) : 0;
  return launch_gemm_bf16(a, out_f32 != 0, stream);
}

int rdm_gemm_bf16_act(const void* x, int32_t ldx, int32_t k, const float* scale, const float* shift, const void* w, int32_t ldw, const float* out_scale,
                      const float* out_shift, void* out, int32_t ldc, int32_t m, int32_t n, void* workspace, size_t workspace_bytes, rdm_stream_t stream) {
  RDM_CHECK_ARG(x && w && out && out_scale && out_shift && m > 0 && n > 0 && k > 0 && ldx >= k && ldw >= k && ldc >= n, "gemm_bf16_act: bad argument");
  RDM_CHECK_ARG(!workspace || ((uintptr_t)workspace & 255) == 0, "gemm_bf16_act: workspace must be 256-byte aligned");
  GemmBf16Args a{};
  a.X = x; a.ldx = ldx; a.K = k; a.scale = scale; a.shift = shift; a.W = w; a.ldw = ldw; a.oscale = out_scale; a.oshift = out_shift; a.out = out; a.ldc = ldc; a.M = m; a.N = n;
  a.partial = static_cast<float*>(workspace); a.partial_floats = workspace ? workspace_bytes / sizeof(float) : 0;
  return launch_gemm_bf16(a, false, stream);
}

size_t rdm_conv3x3_bf16_workspace_bytes(int32_t channels, int32_t batch, int32_t h, int32_t w) {
  if (channels <= 0 || batch <= 0 || h <= 0 || w <= 0) return 0;
  const long M = (long)batch * h * w;
  const int split = std::min(std::max(cdiv(channels, 32) / 2, 1), 16);
  return split > 1 ? (size_t)split * M * 48 * sizeof(float) : 0;
}

int rdm_conv3x3_bf16(const void* y, int32_t ldy, int32_t channels, const float* scale, const float* shift, const void* w_packed, void* out,
                     int32_t ldc, int32_t batch, int32_t h, int32_t w, void* workspace, size_t workspace_bytes, rdm_stream_t stream) {
  RDM_CHECK_ARG(!workspace || ((uintptr_t)workspace & 255) == 0, "conv3x3_bf16: workspace must be 256-byte aligned");
  RDM_CHECK_ARG(y && w_packed && out && (scale == nullptr) == (shift == nullptr) && batch > 0 && h > 0 && w > 0 && channels > 0 && ldy >= channels && ldc >= 48, "conv3x3_bf16: bad argument");
  RDM_CHECK_ARG((long)batch * h * w < (1L << 30), "conv3x3_bf16: too many pixels for 32-bit indices");
  Conv3Bf16Args a{};
  a.Y = y; a.ldy = ldy; a.C = channels; a.scale = scale; a.shift = shift; a.Wt = w_packed; a.wtap = 48L * channels; a.ldw = channels;
  a.out = static_cast<unsigned short*>(out); a.ldc = ldc; a.B = batch; a.H = h; a.W = w; a.M = batch * h * w;
  a.partial = static_cast<float*>(workspace); a.partial_floats = workspace ? workspace_bytes / sizeof(float) : 0;
  return launch_conv3x3_bf16(a, stream);
}

size_t rdm_conv3x3_act_bf16_weight_bytes(int32_t channels) { return channels > 0 ? (size_t)cdiv(channels, 32) * 27 * 1024 : 0; }

int rdm_conv3x3_act_bf16_pack(const float* w_oihw, int32_t channels, void* w_image, rdm_stream_t stream) {
  RDM_CHECK_ARG(w_oihw && w_image && channels > 0 && ((uintptr_t)w_image & 15) == 0, "conv3x3_act_bf16_pack: bad argument");
  return launch_pack_w3_frag_bf16(w_oihw, w_image, channels, cdiv(channels, 32) * 32, 0, stream);
}

static const size_t kActCounterBytes = 16384;                // 4096 tile counters in front of the partial sums

size_t rdm_conv3x3_act_bf16_workspace_bytes(int32_t channels_padded, int32_t batch, int32_t h, int32_t w) {
  if (channels_padded <= 0 || batch <= 0 || h <= 0 || w <= 0) return 0;
  return kActCounterBytes + conv3x3_act_partial_floats(channels_padded, batch, h, w) * sizeof(float);
}

int rdm_conv3x3_act_bf16(const void* y_act, int32_t ldy, int32_t channels_padded, const void* w_image, void* out, int32_t ldc, int32_t batch,
                         int32_t h, int32_t w, void* workspace, size_t workspace_bytes, rdm_stream_t stream) {
  RDM_CHECK_ARG(!workspace || ((uintptr_t)workspace & 255) == 0, "conv3x3_act_bf16: workspace must be 256-byte aligned");
  RDM_CHECK_ARG(y_act && w_image && out && batch > 0 && h > 0 && w > 0 && channels_padded > 0 && ldy >= channels_padded && ldc >= 48, "conv3x3_act_bf16: bad argument");
  RDM_CHECK_ARG((long)batch * h * w < (1L << 30), "conv3x3_act_bf16: too many pixels for 32-bit indices");
  Conv3ActArgs a{};
  a.Y = y_act; a.ldy = ldy; a.C = channels_padded; a.Wimg = w_image; a.out = static_cast<unsigned short*>(out); a.ldc = ldc; a.B = batch; a.H = h; a.W = w;
  if (workspace && workspace_bytes > kActCounterBytes) {
    a.counters = static_cast<unsigned*>(workspace); a.n_counters = (int)(kActCounterBytes / 4);
    a.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + kActCounterBytes);
    a.partial_floats = (workspace_bytes - kActCounterBytes) / sizeof(float);
    RDM_HIP_OK(hipMemsetAsync(workspace, 0, kActCounterBytes, stream));        // the counters are self-resetting; a caller's buffer starts in an unknown state
  }
  return launch_conv3x3_act_bf16(a, stream);
}

static int check_nhwc(const void* p, int ld, int channels, const char* what) {
  RDM_CHECK_ARG(p != nullptr, "%s: NULL tensor", what);
  RDM_CHECK_ARG(channels > 0 && channels % 4 == 0 && ld >= channels && ld % 4 == 0, "%s: channels (%d) and pixel stride (%d) must be multiples of 4, stride >= channels", what, channels, ld);
  RDM_CHECK_ARG(((uintptr_t)p & 15) == 0, "%s: tensor must be 16-byte aligned", what);
  return 0;
}

int rdm_bn_stats(const float* x, int32_t ld, int64_t rows, int32_t channels, double* sum, double* sumsq, rdm_stream_t stream) {
  if (int rc = check_nhwc(x, ld, channels, "bn_stats")) return rc;
  RDM_CHECK_ARG(sum != nullptr && rows > 0 && rows < (1L << 31), "bn_stats: bad argument");
  return launch_colstats(x, ld, (int)rows, channels, sum, sumsq, stream);
}

int rdm_bn_finalize(const double* sum, const double* sumsq, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean, float* save_rstd,
                    int32_t channels, int32_t training, rdm_stream_t stream) {
  RDM_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && save_mean && save_rstd && channels > 0, "bn_finalize: NULL argument");
  RDM_CHECK_ARG(!training || (sum && sumsq && count >= 1), "bn_finalize: training mode needs the sums and a positive count");
  return launch_bn_finalize(sum, sumsq, count, gamma, beta, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), scale, shift,
                            save_mean, save_rstd, channels, training, stream);
}

int rdm_bn_bwd_reduce(float* dz, int32_t dz_ld, const float* x, int32_t x_ld, const float* scale, const float* shift, int64_t rows,
                      int32_t channels, double* sum_dz, double* sum_dz_x, rdm_stream_t stream) {
  if (int rc = check_nhwc(dz, dz_ld, channels, "bn_bwd_reduce(dz)")) return rc;
  if (int rc = check_nhwc(x, x_ld, channels, "bn_bwd_reduce(x)")) return rc;
  RDM_CHECK_ARG(scale && shift && sum_dz && sum_dz_x && rows > 0 && rows < (1L << 31), "bn_bwd_reduce: bad argument");
  RDM_CHECK_ARG((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0, "bn_bwd_reduce: scale / shift must be 16-byte aligned");
  return launch_mask_stats(dz, dz_ld, x, x_ld, scale, shift, (int)rows, channels, sum_dz, sum_dz_x, stream);
}

int rdm_bn_bwd(float* dx, int32_t dx_ld, const float* dz, int32_t dz_ld, const float* x, int32_t x_ld, const double* sum_dz,
               const double* sum_dz_x, double count, const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma,
               float* dbeta, int64_t rows, int32_t channels, int32_t accumulate, int32_t training, rdm_stream_t stream) {
  if (int rc = check_nhwc(dx, dx_ld, channels, "bn_bwd(dx)")) return rc;
  if (int rc = check_nhwc(dz, dz_ld, channels, "bn_bwd(dz)")) return rc;
  if (int rc = check_nhwc(x, x_ld, channels, "bn_bwd(x)")) return rc;
  RDM_CHECK_ARG(sum_dz && sum_dz_x && gamma && save_mean && save_rstd && rows > 0 && rows < (1L << 31) && count >= 1, "bn_bwd: bad argument");
  RDM_CHECK_ARG(accumulate >= 0 && accumulate <= 2, "bn_bwd: accumulate (%d) must be 0 (write), 1 (add) or 2 (write as split rows)", (int)accumulate);
  return launch_bn_bwd_apply(dx, dx_ld, dz, dz_ld, x, x_ld, sum_dz, sum_dz_x, count, gamma, save_mean, save_rstd, dgamma, dbeta, (int)rows, channels,
                             accumulate == 1, training, stream, false, accumulate == 2);
}

int rdm_bn_bwd_defer(float* g, int32_t g_ld, const float* x, int32_t x_ld, const double* sum_dz, const double* sum_dz_x, double count,
                     const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, const float* b_in,
                     const float* c_in, float* b_out, float* c_out, int64_t rows, int32_t channels, int32_t slice_c0, int32_t slice_n,
                     int32_t training, rdm_stream_t stream) {
  if (int rc = check_nhwc(g, g_ld, channels, "bn_bwd_defer(g)")) return rc;
  if (int rc = check_nhwc(x, x_ld, channels, "bn_bwd_defer(x)")) return rc;
  RDM_CHECK_ARG(sum_dz && sum_dz_x && gamma && save_mean && save_rstd && b_in && c_in && b_out && c_out && rows > 0 && rows < (1L << 31) && count >= 1,
                "bn_bwd_defer: bad argument");
  RDM_CHECK_ARG(b_in != b_out && c_in != c_out, "bn_bwd_defer: the running sums are ping-pong buffers (in != out)");
  return launch_bn_bwd_defer(g, g_ld, x, x_ld, sum_dz, sum_dz_x, count, gamma, save_mean, save_rstd, dgamma, dbeta, b_in, c_in, b_out, c_out, (int)rows,
                             channels, slice_c0, slice_n, training, stream);
}

int rdm_maxpool3s2_fwd(const float* x, float* y, int32_t y_ld, uint8_t* argmax, int32_t batch, int32_t h, int32_t w, int32_t channels,
                       rdm_stream_t stream) {
  if (int rc = check_nhwc(x, channels, channels, "maxpool3s2_fwd(x)")) return rc;
  if (int rc = check_nhwc(y, y_ld, channels, "maxpool3s2_fwd(y)")) return rc;
  RDM_CHECK_ARG(argmax && ((uintptr_t)argmax & 3) == 0 && batch > 0 && h > 0 && w > 0, "maxpool3s2_fwd: bad argument");
  return launch_maxpool3s2(x, y, y_ld, argmax, batch, h, w, channels, stream);
}

int rdm_maxpool3s2_bwd(const float* dy, int32_t dy_ld, const uint8_t* argmax, float* dx, int32_t batch, int32_t h, int32_t w, int32_t channels,
                       rdm_stream_t stream) {
  if (int rc = check_nhwc(dy, dy_ld, channels, "maxpool3s2_bwd(dy)")) return rc;
  if (int rc = check_nhwc(dx, channels, channels, "maxpool3s2_bwd(dx)")) return rc;
  RDM_CHECK_ARG(argmax && ((uintptr_t)argmax & 3) == 0 && batch > 0 && h > 0 && w > 0, "maxpool3s2_bwd: bad argument");
  return launch_maxpool3s2_bwd(dy, dy_ld, argmax, dx, batch, h, w, channels, stream);
}

int rdm_padavgpool2_fwd(const float* x, int32_t x_ld, const float* scale, const float* shift, float* pooled, int32_t batch, int32_t h,
                        int32_t w, int32_t channels, rdm_stream_t stream) {
  if (int rc = check_nhwc(x, x_ld, channels, "padavgpool2_fwd(x)")) return rc;
  if (int rc = check_nhwc(pooled, channels, channels, "padavgpool2_fwd(pooled)")) return rc;
  RDM_CHECK_ARG(scale && shift && (((uintptr_t)scale | (uintptr_t)shift) & 15) == 0 && batch > 0 && h > 0 && w > 0, "padavgpool2_fwd: bad argument");
  return launch_trans_pool(x, x_ld, scale, shift, pooled, batch, h, w, channels, stream);
}

static size_t pad256(size_t n) { return (n + 255) & ~(size_t)255; }
size_t rdm_padavgpool2_bwd_workspace_bytes(int32_t channels) {
  return channels > 0 ? pad256(2 * (size_t)channels * sizeof(double)) + 3 * pad256((size_t)channels * sizeof(float)) : 0;
}

int rdm_padavgpool2_bwd(const float* dpooled, const float* x, int32_t x_ld, const float* scale, const float* shift, const float* gamma,
                        const float* save_mean, const float* save_rstd, float* dx, int32_t dx_ld, float* dgamma, float* dbeta, int32_t batch,
                        int32_t h, int32_t w, int32_t channels, int32_t training, void* workspace, size_t workspace_bytes, rdm_stream_t stream) {
  if (int rc = check_nhwc(dpooled, channels, channels, "padavgpool2_bwd(dpooled)")) return rc;
  if (int rc = check_nhwc(x, x_ld, channels, "padavgpool2_bwd(x)")) return rc;
  if (int rc = check_nhwc(dx, dx_ld, channels, "padavgpool2_bwd(dx)")) return rc;
  RDM_CHECK_ARG(scale && shift && gamma && save_mean && save_rstd && batch > 0 && h > 0 && w > 0, "padavgpool2_bwd: bad argument");
  RDM_CHECK_ARG((((uintptr_t)scale | (uintptr_t)shift) & 15) == 0, "padavgpool2_bwd: scale / shift must be 16-byte aligned");
  RDM_CHECK_ARG(workspace && ((uintptr_t)workspace & 255) == 0, "padavgpool2_bwd: workspace must be 256-byte aligned");
  if (workspace_bytes < rdm_padavgpool2_bwd_workspace_bytes(channels)) {
    set_error("padavgpool2_bwd: workspace too small: %zu < %zu", workspace_bytes, rdm_padavgpool2_bwd_workspace_bytes(channels));
    return RDM_ERR_WORKSPACE_TOO_SMALL;
  }
  char* ws = static_cast<char*>(workspace);
  double* s0 = reinterpret_cast<double*>(ws);
  double* s1 = s0 + channels;
  const size_t coff = pad256(2 * (size_t)channels * sizeof(double)), cstep = pad256((size_t)channels * sizeof(float));
  float* cA = reinterpret_cast<float*>(ws + coff);
  float* cB = reinterpret_cast<float*>(ws + coff + cstep);
  float* cC = reinterpret_cast<float*>(ws + coff + 2 * cstep);
  RDM_HIP_OK(hipMemsetAsync(s0, 0, 2 * (size_t)channels * sizeof(double), stream));
  int rc = launch_trans_pool_bwd_reduce(dpooled, x, x_ld, scale, shift, batch, h, w, channels, s0, s1, stream);
  if (rc) return rc;
  const double count = (double)batch * (h + 1) * (w + 1);
  if ((rc = launch_bn_bwd_coeffs(s0, s1, count, gamma, save_mean, save_rstd, cA, cB, cC, dgamma, dbeta, channels, training, stream))) return rc;
  return launch_trans_pool_bwd_apply(dpooled, x, x_ld, scale, shift, cA, cB, cC, dx, dx_ld, batch, h, w, channels, stream);
}

int rdm_split_rows_f32(const float* src, int32_t src_ld, const float* bn_scale, const float* bn_shift, void* dst, int32_t dst_ld, int64_t rows, int32_t channels,
                       rdm_stream_t stream) {
  RDM_CHECK_ARG(src && dst && rows > 0 && channels > 0 && src_ld >= channels && dst_ld >= channels, "split_rows: bad argument");
  return launch_split_rows(src, src_ld, bn_scale, bn_shift, dst, dst_ld, rows, channels, stream);
}

size_t rdm_frame_split_rows_bytes(int32_t batch, int32_t h, int32_t w) { return batch > 0 && h > 0 && w > 0 ? xs_frame_rows_bytes(batch, h, w) : 0; }

int rdm_frame_split_rows_f32(const float* dy, int32_t dy_ld, int32_t channels, int32_t batch, int32_t h, int32_t w, void* dst, rdm_stream_t stream) {
  RDM_CHECK_ARG(dy && dst && batch > 0 && h > 0 && w > 0 && dy_ld >= channels, "frame_split_rows: bad argument");
  return launch_frame_split_rows(dy, dy_ld, channels, batch, h, w, dst, stream);
}

int rdm_layout_nchw_to_nhwc_f32(const float* src, float* dst, int32_t dst_ld, int32_t batch, int32_t channels, int32_t hw, rdm_stream_t stream) {
  RDM_CHECK_ARG(src && dst && batch > 0 && channels > 0 && hw > 0 && dst_ld >= channels, "layout_nchw_to_nhwc: bad argument");
  return launch_nchw_to_nhwc(src, dst, dst_ld, batch, channels, hw, stream);
}

int rdm_layout_nhwc_to_nchw_f32(const float* src, int32_t src_ld, float* dst, int32_t batch, int32_t channels, int32_t hw, rdm_stream_t stream) {
  RDM_CHECK_ARG(src && dst && batch > 0 && channels > 0 && hw > 0 && src_ld >= channels, "layout_nhwc_to_nchw: bad argument");
  return launch_nhwc_to_nchw(src, src_ld, dst, batch, channels, hw, stream);
}

int rdm_adamw_fused(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step, float grad_scale, rdm_stream_t stream) {
  RDM_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "adamw_fused: bad argument");
  RDM_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "adamw_fused: buffers must be 16-byte aligned");
  if (n == 0) return RDM_OK;
  return launch_adamw(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

}  // extern "C"
