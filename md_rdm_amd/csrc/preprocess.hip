// NYU input pipeline on the GPU: dataloaders/nyu_dataloader.py:240-287 (training_preprocess /
// validation_preprocess) for a whole batch, bit-exact with the Pillow arithmetic the reference reaches
// through torchvision's PIL transforms (oracle/preprocess_cpu.py restates it and is pinned to Pillow):
//
//   depth / s  ->  ColorJitter (ImageEnhance.Brightness / Contrast / Color = Blend.c over a degenerate image)
//   -> Resize(250)  (Resample.c bilinear with antialias support: horizontal pass, then vertical;
//                    8-bit images in 22-bit fixed point rounded to uint8 after EACH pass, mode-F depth in
//                    double accumulation rounded to float)
//   -> rotate(angle) (Geometry.c affine_fixed: NEAREST through 16.16 fixed-point coordinates, fill 0)
//   -> Resize(int(250 s)) -> CenterCrop -> hflip -> to_tensor (uint8 / 255, CHW float32).
//
// The uint8 rounding between stages is part of the reference's result, so the stages stay separate
// images (a few hundred KB per sample, L2/Infinity-Cache resident) instead of one fused resampling;
// the second resize only computes the centre-crop window, and crop + flip + to_tensor are folded into
// its vertical pass.  Everything is HBM/latency-bound byte work: one thread per output pixel, filter
// coefficients recomputed per thread in float64 exactly as precompute_coeffs does (<= 9 taps).
// Floating-point contraction is OFF in this file: Pillow's C is compiled without FMA.
#include <algorithm>

#include "rdm_common.h"

#pragma clang fp contract(off)

namespace rdm {

struct NyuAug {            // mirrors rdm_nyu_aug (include/rdm_hip.h)
  float depth_div;
  int rot[6];
  int h2, w2, top, left, flip;
  int op[3];
  float factor[3];
  int crop2[4];            // region of the rotated image the second Resize reads: top, left, height, width
};
static_assert(sizeof(NyuAug) == 88, "rdm_nyu_aug layout");

constexpr int KMAX = 9;    // taps of the antialiased triangle filter for a down-scale of up to 4

struct Coef { int xmin, n; double k[KMAX]; };

// Resample.c precompute_coeffs, bilinear_filter, box = the whole axis
__device__ __forceinline__ void bilinear_coeffs(int in_size, int out_size, int xx, Coef& c) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > KMAX) xmax = KMAX;               // unreachable: the launcher bounds the scale
  double ww = 0.0;
#pragma unroll
  for (int x = 0; x < KMAX; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w = a < 1.0 ? 1.0 - a : 0.0;
      ww += w;
    }
    c.k[x] = w;
  }
#pragma unroll
  for (int x = 0; x < KMAX; ++x)
    if (x < xmax && ww != 0.0) c.k[x] /= ww;
  c.xmin = xmin; c.n = xmax;
}

// normalize_coeffs_8bpc
__device__ __forceinline__ int fixed22(double k) { return (int)(k < 0.0 ? -0.5 + k * (double)(1 << 22) : 0.5 + k * (double)(1 << 22)); }
__device__ __forceinline__ unsigned char clip8(int v) { return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }

// ---------------------------------------------------------------------------------------------
// colour jitter
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int luma601(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // Convert.c rgb2l

// Blend.c ImagingBlend(degenerate, image, alpha): float32 arithmetic, truncating cast
__device__ __forceinline__ unsigned char blend8(int deg, int img, float alpha, bool interp) {
  const float t = (float)deg + alpha * (float)(img - deg);
  if (interp) return (unsigned char)(int)t;
  return t <= 0.0f ? 0 : t >= 255.0f ? 255 : (unsigned char)(int)t;
}

// round r: luma sum of every sample whose r-th op is Contrast (ImageStat.mean of convert("L"))
__global__ __launch_bounds__(256) void k_luma_sum(const unsigned char* __restrict__ img, long npix, const NyuAug* __restrict__ aug, int r,
                                                  unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  if (aug[b].op[r] != 1) return;
  const unsigned char* p = img + (long)b * npix * 3;
  unsigned long long s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) s += (unsigned)luma601(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  __shared__ unsigned long long sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[b], sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void k_jitter_apply(unsigned char* __restrict__ img, long npix, const NyuAug* __restrict__ aug, int r,
                                                      const unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const int op = aug[b].op[r];
  if (op < 0 || op > 2) return;
  const float alpha = aug[b].factor[r];
  const bool interp = alpha >= 0.0f && alpha <= 1.0f;
  int mean = 0;
  if (op == 1) mean = (int)((double)sums[b] / (double)npix + 0.5);
  unsigned char* p = img + (long)b * npix * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const int R = p[3 * i], G = p[3 * i + 1], B = p[3 * i + 2];
    const int deg = op == 0 ? 0 : op == 1 ? mean : luma601(R, G, B);
    p[3 * i] = blend8(deg, R, alpha, interp);
    p[3 * i + 1] = blend8(deg, G, alpha, interp);
    p[3 * i + 2] = blend8(deg, B, alpha, interp);
  }
}

// ---------------------------------------------------------------------------------------------
// resampling passes.  Images are (B, rows, cols[, 3]); `aug` == nullptr => uniform output size.
// ---------------------------------------------------------------------------------------------
// horizontal: out[b][y][c] for c in [0, ncols) <- output column col0_b + c of an (in_w -> out_w_b) resize
template <bool SECOND>
__global__ __launch_bounds__(256) void k_resample_h(const unsigned char* __restrict__ rgb_in, const float* __restrict__ dep_in, int rows, int in_w,
                                                    unsigned char* __restrict__ rgb_out, float* __restrict__ dep_out, int ncols, int out_w_uniform,
                                                    const NyuAug* __restrict__ aug, int first_divide) {
  const int b = blockIdx.z, y = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  const int out_w = SECOND ? aug[b].w2 : out_w_uniform;
  const int xx = (SECOND ? aug[b].left : 0) + c;
  // SECOND: the pass reads the crop2 window of the (rows x in_w) image: row y of the window, columns from crop2.left
  const int src_w = SECOND ? aug[b].crop2[3] : in_w;
  if (SECOND && y >= aug[b].crop2[2]) return;
  const long src_row = SECOND ? (long)(aug[b].crop2[0] + y) * in_w + aug[b].crop2[1] : (long)y * in_w;
  Coef k;
  bilinear_coeffs(src_w, out_w, xx, k);
  const unsigned char* rp = rgb_in + ((long)b * rows * in_w + src_row) * 3;
  const float* dp = dep_in + (long)b * rows * in_w + src_row;
  const float div = first_divide ? aug[b].depth_div : 1.0f;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  double sd = 0.0;
#pragma unroll
  for (int x = 0; x < KMAX; ++x) {
    if (x < k.n) {
      const int kk = fixed22(k.k[x]);
      const unsigned char* q = rp + (long)(k.xmin + x) * 3;
      s0 += q[0] * kk; s1 += q[1] * kk; s2 += q[2] * kk;
      float d = dp[k.xmin + x];
      if (first_divide) d = __fdiv_rn(d, div);               // depth / s in float32 (nyu_dataloader.py:241-242)
      sd += (double)d * k.k[x];
    }
  }
  unsigned char* ro = rgb_out + (((long)b * rows + y) * ncols + c) * 3;
  ro[0] = clip8(s0 >> 22); ro[1] = clip8(s1 >> 22); ro[2] = clip8(s2 >> 22);
  dep_out[((long)b * rows + y) * ncols + c] = (float)sd;
}

// vertical, intermediate: out[b][r][c] <- output row r of an (in_h -> out_h) resize (uniform sizes)
__global__ __launch_bounds__(256) void k_resample_v(const unsigned char* __restrict__ rgb_in, const float* __restrict__ dep_in, int in_h, int cols,
                                                    unsigned char* __restrict__ rgb_out, float* __restrict__ dep_out, int out_h) {
  const int b = blockIdx.z, r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  Coef k;
  bilinear_coeffs(in_h, out_h, r, k);
  const unsigned char* rp = rgb_in + (long)b * in_h * cols * 3 + (long)c * 3;
  const float* dp = dep_in + (long)b * in_h * cols + c;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  double sd = 0.0;
#pragma unroll
  for (int x = 0; x < KMAX; ++x) {
    if (x < k.n) {
      const int kk = fixed22(k.k[x]);
      const unsigned char* q = rp + (long)(k.xmin + x) * cols * 3;
      s0 += q[0] * kk; s1 += q[1] * kk; s2 += q[2] * kk;
      sd += (double)dp[(long)(k.xmin + x) * cols] * k.k[x];
    }
  }
  unsigned char* ro = rgb_out + (((long)b * out_h + r) * cols + c) * 3;
  ro[0] = clip8(s0 >> 22); ro[1] = clip8(s1 >> 22); ro[2] = clip8(s2 >> 22);
  dep_out[((long)b * out_h + r) * cols + c] = (float)sd;
}

// vertical, final: crop rows [top, top+oh) of the (in_h -> h2_b) resize, hflip, to_tensor -> planar float32
__global__ __launch_bounds__(256) void k_resample_v_final(const unsigned char* __restrict__ rgb_in, const float* __restrict__ dep_in, int in_h, int cols,
                                                          float* __restrict__ x_out, float* __restrict__ y_out, int oh, const NyuAug* __restrict__ aug) {
  const int b = blockIdx.z, r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  Coef k;
  bilinear_coeffs(aug[b].crop2[2], aug[b].h2, aug[b].top + r, k);     // rows of the crop2 window (t4 holds them from row 0)
  const unsigned char* rp = rgb_in + (long)b * in_h * cols * 3 + (long)c * 3;
  const float* dp = dep_in + (long)b * in_h * cols + c;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  double sd = 0.0;
#pragma unroll
  for (int x = 0; x < KMAX; ++x) {
    if (x < k.n) {
      const int kk = fixed22(k.k[x]);
      const unsigned char* q = rp + (long)(k.xmin + x) * cols * 3;
      s0 += q[0] * kk; s1 += q[1] * kk; s2 += q[2] * kk;
      sd += (double)dp[(long)(k.xmin + x) * cols] * k.k[x];
    }
  }
  const int oc = aug[b].flip ? cols - 1 - c : c;
  const long plane = (long)oh * cols;
  float* xo = x_out + (long)b * 3 * plane + (long)r * cols + oc;
  xo[0] = __fdiv_rn((float)clip8(s0 >> 22), 255.0f);           // TF.to_tensor: uint8 -> float32 / 255
  xo[plane] = __fdiv_rn((float)clip8(s1 >> 22), 255.0f);
  xo[2 * plane] = __fdiv_rn((float)clip8(s2 >> 22), 255.0f);
  y_out[(long)b * plane + (long)r * cols + oc] = (float)sd;
}

// Geometry.c affine_fixed, NEAREST, fill 0
__global__ __launch_bounds__(256) void k_rotate_nearest(const unsigned char* __restrict__ rgb_in, const float* __restrict__ dep_in, int h, int w,
                                                        unsigned char* __restrict__ rgb_out, float* __restrict__ dep_out, const NyuAug* __restrict__ aug) {
  const int b = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
  if (x >= w) return;
  const int* a = aug[b].rot;
  const int xin = (a[2] + a[0] * x + a[1] * y) >> 16, yin = (a[5] + a[3] * x + a[4] * y) >> 16;
  const bool ok = xin >= 0 && xin < w && yin >= 0 && yin < h;
  const long o = ((long)b * h + y) * w + x;
  unsigned char r0 = 0, r1 = 0, r2 = 0;
  float d = 0.f;
  if (ok) {
    const long i = ((long)b * h + yin) * w + xin;
    r0 = rgb_in[3 * i]; r1 = rgb_in[3 * i + 1]; r2 = rgb_in[3 * i + 2];
    d = dep_in[i];
  }
  rgb_out[3 * o] = r0; rgb_out[3 * o + 1] = r1; rgb_out[3 * o + 2] = r2;
  dep_out[o] = d;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
namespace {
struct PrepLayout { size_t img, t1, d1, t2, d2, t3, d3, t4, d4, sums, total; };
inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
PrepLayout prep_layout(long B, long H, long W, long h1, long w1, long ow) {
  PrepLayout L{};
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += al256(bytes); return at; };
  L.img = take((size_t)B * H * W * 3);
  L.t1 = take((size_t)B * H * w1 * 3);  L.d1 = take((size_t)B * H * w1 * 4);
  L.t2 = take((size_t)B * h1 * w1 * 3); L.d2 = take((size_t)B * h1 * w1 * 4);
  L.t3 = take((size_t)B * h1 * w1 * 3); L.d3 = take((size_t)B * h1 * w1 * 4);
  L.t4 = take((size_t)B * h1 * ow * 3); L.d4 = take((size_t)B * h1 * ow * 4);
  L.sums = take((size_t)3 * B * 8);
  L.total = o;
  return L;
}
}  // namespace

size_t nyu_preprocess_workspace_bytes(int B, int H, int W, int h1, int w1, int out_w) { return prep_layout(B, H, W, h1, w1, out_w).total; }

int launch_nyu_preprocess(const unsigned char* rgb, const float* depth, const void* aug_dev, int B, int H, int W, int h1, int w1, int oh, int ow,
                          float* x, float* y, void* ws, size_t ws_bytes, hipStream_t s) {
  RDM_CHECK_ARG(rgb && depth && aug_dev && x && y && ws, "nyu_preprocess: null pointer");
  RDM_CHECK_ARG(B > 0 && H > 0 && W > 0 && h1 > 0 && w1 > 0 && oh > 0 && ow > 0, "nyu_preprocess: sizes must be positive");
  RDM_CHECK_ARG((long)H <= 4L * h1 && (long)W <= 4L * w1, "nyu_preprocess: down-scale beyond 4x (%dx%d -> %dx%d) exceeds the %d-tap filter window", H, W, h1, w1, KMAX);
  RDM_CHECK_ARG(B <= 65535 && H <= 65535, "nyu_preprocess: batch / height exceed the grid limits");
  RDM_CHECK_ARG(h1 < 16384 && w1 < 16384, "nyu_preprocess: the 16.16 fixed-point rotation needs images below 16384 pixels per side");
  const PrepLayout L = prep_layout(B, H, W, h1, w1, ow);
  RDM_CHECK_ARG(ws_bytes >= L.total, "nyu_preprocess: workspace %zu < required %zu bytes", ws_bytes, L.total);
  const NyuAug* aug = static_cast<const NyuAug*>(aug_dev);
  char* base = static_cast<char*>(ws);
  unsigned char* img = reinterpret_cast<unsigned char*>(base + L.img);
  unsigned char *t1 = reinterpret_cast<unsigned char*>(base + L.t1), *t2 = reinterpret_cast<unsigned char*>(base + L.t2);
  unsigned char *t3 = reinterpret_cast<unsigned char*>(base + L.t3), *t4 = reinterpret_cast<unsigned char*>(base + L.t4);
  float *d1 = reinterpret_cast<float*>(base + L.d1), *d2 = reinterpret_cast<float*>(base + L.d2);
  float *d3 = reinterpret_cast<float*>(base + L.d3), *d4 = reinterpret_cast<float*>(base + L.d4);
  unsigned long long* sums = reinterpret_cast<unsigned long long*>(base + L.sums);
  const long npix = (long)H * W;
  // colour jitter on a private copy (the three ops run in the per-sample order)
  RDM_HIP_OK(hipMemcpyAsync(img, rgb, (size_t)B * npix * 3, hipMemcpyDeviceToDevice, s));
  RDM_HIP_OK(hipMemsetAsync(sums, 0, (size_t)3 * B * 8, s));
  const int jb = (int)std::min<long>(cdiv(npix, 256 * 4), 512);
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(k_luma_sum, dim3(jb, B), dim3(256), 0, s, img, npix, aug, r, sums + (size_t)r * B);
    hipLaunchKernelGGL(k_jitter_apply, dim3(jb, B), dim3(256), 0, s, img, npix, aug, r, sums + (size_t)r * B);
  }
  // Resize(resize): H x W -> h1 x w1
  hipLaunchKernelGGL(k_resample_h<false>, dim3(cdiv(w1, 256), H, B), dim3(256), 0, s, img, depth, H, W, t1, d1, w1, w1, aug, 1);
  hipLaunchKernelGGL(k_resample_v, dim3(cdiv(w1, 256), h1, B), dim3(256), 0, s, t1, d1, H, w1, t2, d2, h1);
  // rotate
  hipLaunchKernelGGL(k_rotate_nearest, dim3(cdiv(w1, 256), h1, B), dim3(256), 0, s, t2, d2, h1, w1, t3, d3, aug);
  // Resize(int(resize * s)) restricted to the centre-crop window, + hflip + to_tensor
  hipLaunchKernelGGL(k_resample_h<true>, dim3(cdiv(ow, 256), h1, B), dim3(256), 0, s, t3, d3, h1, w1, t4, d4, ow, 0, aug, 0);
  hipLaunchKernelGGL(k_resample_v_final, dim3(cdiv(ow, 256), oh, B), dim3(256), 0, s, t4, d4, h1, ow, x, y, oh, aug);
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
