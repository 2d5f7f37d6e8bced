// HBM-bound helper kernels of the conv stack (NHWC, float4-vectorised, wave64):
// BatchNorm statistics / finalisation / backward coefficients, transition pooling,
// 3x3/s2 max-pool, im2col of the 7x7/s2 stem, layout packing, fused AdamW.
// Reference call sites: network/RDM_Net.py:524-532 (stem, pool, pad_br, transitions),
// torchvision _DenseLayer BatchNorm2d (third party), network/module.py:41 (AdamW).
#include "rdm_common.h"
#include "elementwise.h"
#include "xsplit_dev.h"

namespace rdm {


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ------------------------------------------------------------------------------------------
// per-channel reductions over rows of a [M][ld] matrix: 256 threads = 64 float4-columns x 4
// row lanes; f32 partials per thread (<= rows_per_block/4 terms), f64 atomics per block.
//   MODE 0: s0 = sum v, s1 = sum v^2
//   MODE 1: v <- v * relu'(x*xs+xt) written back in place; s0 = sum v, s1 = sum v*x
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_colreduce(float* V, int ldv, const float* X, int ldx, const float* xs, const float* xt,
                                                   int M, int C, int rows_per_block, double* s0, double* s1) {
  __shared__ float4 red0[256], red1[256];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 a0 = make_float4(0, 0, 0, 0), a1 = make_float4(0, 0, 0, 0);
  if (c < C) {
    float4 sc = make_float4(0, 0, 0, 0), sh = sc;
    if (MODE == 1) { sc = ld4(xs + c); sh = ld4(xt + c); }
    for (int m = r0 + ry; m < r1; m += 4) {
      float4 v = ld4(V + (long)m * ldv + c);
      if (MODE == 0) {
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
        a1.x += v.x * v.x; a1.y += v.y * v.y; a1.z += v.z * v.z; a1.w += v.w * v.w;
      } else {
        const float4 x = ld4(X + (long)m * ldx + c);
        v.x = fmaf(x.x, sc.x, sh.x) > 0.f ? v.x : 0.f;
        v.y = fmaf(x.y, sc.y, sh.y) > 0.f ? v.y : 0.f;
        v.z = fmaf(x.z, sc.z, sh.z) > 0.f ? v.z : 0.f;
        v.w = fmaf(x.w, sc.w, sh.w) > 0.f ? v.w : 0.f;
        st4(V + (long)m * ldv + c, v);
        a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w;
        a1.x += v.x * x.x; a1.y += v.y * x.y; a1.z += v.z * x.z; a1.w += v.w * x.w;
      }
    }
  }
  red0[threadIdx.x] = a0; red1[threadIdx.x] = a1;
  __syncthreads();
  if (ry == 0 && c < C) {
    for (int k = 1; k < 4; ++k) {
      const float4 b0 = red0[cx + 64 * k], b1 = red1[cx + 64 * k];
      a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
      a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    }
    atomicAdd(s0 + c, (double)a0.x); atomicAdd(s0 + c + 1, (double)a0.y); atomicAdd(s0 + c + 2, (double)a0.z); atomicAdd(s0 + c + 3, (double)a0.w);
    if (s1) { atomicAdd(s1 + c, (double)a1.x); atomicAdd(s1 + c + 1, (double)a1.y); atomicAdd(s1 + c + 2, (double)a1.z); atomicAdd(s1 + c + 3, (double)a1.w); }
  }
}

// rows per workgroup: small matrices (split-K layers) get short row chunks so the reduction still
// spreads over a few hundred workgroups instead of a handful of long serial loops
static int rows_per_block(long M, int C) {
  // every workgroup ends with one f64 atomic per channel, so the number of ROW chunks is the
  // contention per address: aim for ~48 chunks (measured: 143 chunks made a 13 MB reduction take
  // 28 us, atomics-bound), but never fewer than 16 or more than 512 rows per workgroup
  (void)C;
  if (t_deterministic) return (int)std::min<long>(M, 0x7FFFFFF0L);      // one row chunk per column group: fixed order, one add onto the zeroed slot
  long rpb = (M + 47) / 48;
  rpb = (rpb + 15) / 16 * 16;
  if (rpb < 16) rpb = 16;
  if (rpb > 512) rpb = 512;
  return (int)rpb;
}

int launch_colstats(const float* V, int ldv, int M, int C, double* sum, double* sq, hipStream_t s) {
  const int rpb = rows_per_block(M, C);
  dim3 grid(cdiv(C / 4, 64), cdiv(M, rpb));
  hipLaunchKernelGGL(k_colreduce<0>, grid, dim3(256), 0, s, const_cast<float*>(V), ldv, nullptr, 0, nullptr, nullptr, M, C, rpb, sum, sq);
  RDM_LAUNCH_OK();
  return 0;
}

int launch_mask_stats(float* V, int ldv, const float* X, int ldx, const float* xs, const float* xt, int M, int C, double* s0,
                      double* s1, hipStream_t s) {
  const int rpb = rows_per_block(M, C);
  dim3 grid(cdiv(C / 4, 64), cdiv(M, rpb));
  hipLaunchKernelGGL(k_colreduce<1>, grid, dim3(256), 0, s, V, ldv, X, ldx, xs, xt, M, C, rpb, s0, s1);
  RDM_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// BatchNorm finalisation: statistics -> (scale, shift) used by the consumer's conv prologue,
// + running-stat update (momentum 0.1, unbiased variance) exactly as nn.BatchNorm2d.
// ------------------------------------------------------------------------------------------
__global__ void k_bn_finalize(const double* sum, const double* sq, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, long long* num_batches, float* scale, float* shift,
                              float* save_mean, float* save_rstd, int C, int training, float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && num_batches) *num_batches += 1;
  if (c >= C) return;
  float mean, rstd, sc, sh;
  if (training) {
    double var;
    bn_affine_from_sums(sum[c], sq[c], count, gamma[c], beta[c], eps, sc, sh, mean, rstd, var);
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  } else {
    mean = running_mean[c];
    rstd = 1.f / sqrtf(running_var[c] + eps);
    sc = gamma[c] * rstd;
    sh = beta[c] - mean * sc;
  }
  scale[c] = sc;
  shift[c] = sh;
  save_mean[c] = mean;
  save_rstd[c] = rstd;
}

int launch_bn_finalize(const double* sum, const double* sq, double count, const float* gamma, const float* beta, float* rm,
                       float* rv, long long* nbt, float* scale, float* shift, float* save_mean, float* save_rstd, int C,
                       int training, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 256)), dim3(256), 0, s, sum, sq, count, gamma, beta, rm, rv, nbt, scale, shift,
                     save_mean, save_rstd, C, training, 0.1f, 1e-5f);
  RDM_LAUNCH_OK();
  return 0;
}

// The same finalisation for up to BN_BATCH BatchNorms in ONE launch (blockIdx.y = entry): the bookkeeping of the few-pixel blocks, whose
// consumers form the affine themselves (RAW prologue), so nothing waits for these values before backward.
__global__ void k_bn_finalize_batch(BnBatch bb, int training, float momentum, float eps) {
  const BnBatchEntry& e = bb.e[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && training && e.nbt) *e.nbt += 1;
  if (c >= e.C) return;
  float mean, rstd, sc, sh;
  double var;
  bn_affine_from_sums(e.sum[c], e.sq[c], e.count, e.gamma[c], e.beta[c], eps, sc, sh, mean, rstd, var);
  const double unbiased = e.count > 1 ? var * e.count / (e.count - 1) : var;
  e.rm[c] = (1.f - momentum) * e.rm[c] + momentum * mean;
  e.rv[c] = (1.f - momentum) * e.rv[c] + momentum * (float)unbiased;
  e.out[c] = sc; e.out[e.Cout + c] = sh; e.out[2 * e.Cout + c] = mean; e.out[3 * e.Cout + c] = rstd;
}

int launch_bn_finalize_batch(const BnBatch& bb, int count, int max_c, hipStream_t s) {
  if (count <= 0) return 0;
  hipLaunchKernelGGL(k_bn_finalize_batch, dim3(cdiv(max_c, 256), count), dim3(256), 0, s, bb, 1, 0.1f, 1e-5f);
  RDM_LAUNCH_OK();
  return 0;
}

// BatchNorm backward coefficients: dx = A*dz + B*x + Cc (dz already ReLU-masked),
// dgamma = sum dz*xhat, dbeta = sum dz.  s0 = sum dz, s1 = sum dz*x.
__global__ void k_bn_bwd_coeffs(const double* s0, const double* s1, double count, const float* gamma, const float* mean,
                                const float* rstd, float* A, float* Bc, float* Cc, float* dgamma, float* dbeta, int C, int training) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = mean[c], rs = rstd[c], g = gamma[c];
  const double sdz = s0[c], sdzx = s1[c];
  const double sdzxhat = rs * (sdzx - mu * sdz);
  if (dgamma) dgamma[c] = (float)sdzxhat;
  if (dbeta) dbeta[c] = (float)sdz;
  const double a = g * rs;
  if (training) {
    const double m1 = sdz / count, m2 = sdzxhat / count;
    const double b = -a * rs * m2;
    A[c] = (float)a; Bc[c] = (float)b; Cc[c] = (float)(-a * m1 - b * mu);
  } else {
    A[c] = (float)a; Bc[c] = 0.f; Cc[c] = 0.f;
  }
}

int launch_bn_bwd_coeffs(const double* s0, const double* s1, double count, const float* gamma, const float* mean, const float* rstd,
                         float* A, float* Bc, float* Cc, float* dgamma, float* dbeta, int C, int training, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_bwd_coeffs, dim3(cdiv(C, 256)), dim3(256), 0, s, s0, s1, count, gamma, mean, rstd, A, Bc, Cc, dgamma, dbeta, C, training);
  RDM_LAUNCH_OK();
  return 0;
}

// Deferred norm1 backward (RDM_NET_OPT_DEFER_NORM1).  The BatchNorm backward of a dense layer's norm1 adds  a*dz + b*x + c  into the block
// gradient G over ALL the layer's input channels - an O(layers^2) elementwise pass (k_bn_bwd_apply<true>).  a = gamma * rstd is known before the
// layer's dgrad runs, so the 1x1 dgrad's epilogue adds a*dz itself; b and c (they need the dgrad's channel sums) only multiply x and 1, and x - the
// block buffer's channel - is the same for every layer that reads it: the (b, c) of all layers are SUMMED per channel (ping-pong running sums:
// layer i reads `in`, writes `out` for its channels < C) and applied when a channel's gradient is needed next - the 48 channels
// [slice_c0, slice_c0 + slice_n) that the next layer down produced (or the block's input channels after its first layer).  This kernel: the
// layer's coefficients, dgamma / dbeta, the running sums, and the slice's  G += B x + C.
__global__ __launch_bounds__(256) void k_bn_bwd_defer(float* __restrict__ G, int ldg, const float* __restrict__ x, int ldx, const double* __restrict__ s0,
                                                     const double* __restrict__ s1, double inv_count, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, const float* __restrict__ b_in, const float* __restrict__ c_in,
                                                     float* __restrict__ b_out, float* __restrict__ c_out, int M, int C, int slice_c0, int slice_n,
                                                     int training) {
  extern __shared__ float defer_lds[];                                // [slice_n] B | [slice_n] C
  float* const sB = defer_lds;
  float* const sC = defer_lds + slice_n;
  const int tid = threadIdx.x;
  auto coeffs = [&](int c, float& fb, float& fc, float& dg, float& db) {      // the arithmetic of k_bn_bwd_apply
    const double mu = mean[c], rs = rstd[c], g = gamma[c];
    const double sdz = s0[c], sdzx = s1[c];
    const double sdzxhat = rs * (sdzx - mu * sdz);
    dg = (float)sdzxhat; db = (float)sdz;
    fb = 0.f; fc = 0.f;
    if (training) {
      const double aa = g * rs, m1 = sdz * inv_count, m2 = sdzxhat * inv_count;
      const double bb = -aa * rs * m2;
      fb = (float)bb; fc = (float)(-aa * m1 - bb * mu);
    }
  };
  for (int i = tid; i < slice_n; i += 256) {
    const int c = slice_c0 + i;
    float fb, fc, dg, db;
    coeffs(c, fb, fc, dg, db);
    sB[i] = b_in[c] + fb; sC[i] = c_in[c] + fc;
  }
  if (blockIdx.x == 0) {
    for (int c = tid; c < C; c += 256) {
      float fb, fc, dg, db;
      coeffs(c, fb, fc, dg, db);
      b_out[c] = b_in[c] + fb; c_out[c] = c_in[c] + fc;
      if (dgamma) dgamma[c] = dg;
      if (dbeta) dbeta[c] = db;
    }
  }
  __syncthreads();
  const int Q = slice_n >> 2;
  const long total = (long)M * Q;
  for (long i = (long)blockIdx.x * 256 + tid; i < total; i += (long)gridDim.x * 256) {
    const long m = i / Q;
    const int c = (int)(i - m * Q) * 4;
    float* gp = G + m * ldg + slice_c0 + c;
    float4 gv = ld4(gp);
    const float4 xv = ld4(x + m * ldx + slice_c0 + c);
    gv.x += fmaf(sB[c], xv.x, sC[c]); gv.y += fmaf(sB[c + 1], xv.y, sC[c + 1]);
    gv.z += fmaf(sB[c + 2], xv.z, sC[c + 2]); gv.w += fmaf(sB[c + 3], xv.w, sC[c + 3]);
    st4(gp, gv);
  }
}

int launch_bn_bwd_defer(float* G, int ldg, const float* x, int ldx, const double* s0, const double* s1, double count, const float* gamma, const float* mean,
                        const float* rstd, float* dgamma, float* dbeta, const float* b_in, const float* c_in, float* b_out, float* c_out, int M, int C,
                        int slice_c0, int slice_n, int training, hipStream_t s) {
  RDM_CHECK_ARG(slice_n > 0 && slice_n % 4 == 0 && slice_c0 % 4 == 0 && slice_c0 + slice_n <= C && ldg % 4 == 0 && ldx % 4 == 0 && slice_n <= 4096,
                "bn_bwd_defer: slice [%d, +%d) of %d channels", slice_c0, slice_n, C);
  const long total = (long)M * (slice_n / 4);
  const unsigned blocks = (unsigned)std::min<long>(std::max<long>(cdiv(total, 256 * 4), 1), 2048);
  hipLaunchKernelGGL(k_bn_bwd_defer, dim3(blocks), dim3(256), (size_t)slice_n * 8, s, G, ldg, x, ldx, s0, s1, 1.0 / count, gamma, mean, rstd, dgamma, dbeta, b_in,
                     c_in, b_out, c_out, M, C, slice_c0, slice_n, training);
  RDM_LAUNCH_OK();
  return 0;
}

// Zero fill as an ordinary kernel.  hipMemsetAsync / hipMemset2DAsync go through the runtime's fill path, which
// showed up as ~0.8 ms of GPU idle time per step in front of 360 fills (kernel trace); a plain launch queues back-to-back.
// rows x row_floats floats with a row pitch of `ld` floats (ld == row_floats: contiguous).  All multiples of 4, 16-B aligned.
__global__ __launch_bounds__(256) void k_zero_rows(float* __restrict__ p, long ld4, long row4, long total4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long r = i / row4, c = i - r * row4;
    st4(p + (r * ld4 + c) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
  }
}
__global__ __launch_bounds__(256) void k_zero_bytes(unsigned char* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0;
}

int launch_zero_rows(float* p, long rows, long row_floats, long ld, hipStream_t s) {
  if (rows <= 0 || row_floats <= 0) return 0;
  if ((row_floats & 3) || (ld & 3) || ((uintptr_t)p & 15)) {           // odd shapes: byte loop (never on the hot path)
    RDM_CHECK_ARG(ld == row_floats, "zero fill: a strided region needs 16-byte aligned rows");
    const size_t n = (size_t)rows * row_floats * 4;
    hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)std::min<size_t>(cdiv(n, 256), 4096)), dim3(256), 0, s, reinterpret_cast<unsigned char*>(p), n);
  } else {
    long row4 = row_floats / 4, ld4 = ld / 4;
    if (ld == row_floats) { row4 *= rows; ld4 = row4; rows = 1; }     // contiguous: one long row
    const long total4 = rows * row4;
    hipLaunchKernelGGL(k_zero_rows, dim3((unsigned)std::min<long>(cdiv(total4, 256), 256 * 16)), dim3(256), 0, s, p, ld4, row4, total4);
  }
  RDM_LAUNCH_OK();
  return 0;
}

// BN backward in ONE pass: the coefficient computation of k_bn_bwd_coeffs is done per thread for its
// four channels (a thread keeps its channel group and walks rows), then dst (=|+=) A*dz + B*x + Cc.
// Removes a 4-us dependent launch from every BatchNorm of the backward chain (156 per step) - and with
// it the stall of a tiny kernel queued behind the side stream's long-running wgrad blocks.
//
// This pass runs on the dependent chain WHILE the side stream's weight-gradient kernels own the chip (2 waves per SIMD at 219 VGPRs,
// issue-bound): beside conv_wgrad3_row_kernel the first version took 1 260 us for dense_e2's 2.3 GB instead of 470 us alone, while a
// plain float4 copy only went from 305 to 350 us (tools/coresidency_probe.py).  The difference is VALU instructions per byte - one
// co-resident wave (<= 64 VGPRs fit beside the two) gets the issue slots the older waves leave over.  Hence: SRD buffer accesses with
// the row advance in an SGPR offset (no vector address arithmetic at all), packed v_pk_fma_f32 (4 instead of 8 FMAs per float4),
// x (1/count) instead of eight f64 divisions per thread, and half as many, longer workgroups.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
typedef float f32x4e __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2e __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2e __attribute__((ext_vector_type(2)));

// BF (mixed-precision arithmetic mode, RDM_NET_OPT_GEMM_BF16): dz and dst are bf16 rows (ldz / ldd in ELEMENTS of that type) - the consumers of dst
// (the 1x1 dgrad / wgrad on bf16 operands) round it to bf16 anyway, so storing it as bf16 loses nothing and halves two of the three streams
// SP (split rows, xsplit_dev.h): dst is written as [hi x4 | lo x4] bf16 groups at the addresses of the float32 values - the two consumers of dY (the
// split-precision 1x1 dgrad / wgrad) then stage it verbatim instead of each converting and splitting every element again
template <bool ACC, bool BF = false, bool SP = false>
__global__ __launch_bounds__(256, 8) void k_bn_bwd_apply(float* dst, int ldd, const float* dz, int ldz, const float* x, int ldx, const double* s0,
                                                         const double* s1, double inv_count, const float* gamma, const float* mean,
                                                         const float* rstd, float* dgamma, float* dbeta, int M, int C4, int rows_per_block,
                                                         int training, unsigned dst_bytes, unsigned dz_bytes, unsigned x_bytes) {
  const int c4 = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c4 >= C4) return;
  const int c = c4 * 4, rsub = threadIdx.x >> 6;
  f32x2 a[2], b[2], cc[2];
  float dg[4], db[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double mu = mean[c + j], rs = rstd[c + j], g = gamma[c + j];
    const double sdz = s0[c + j], sdzx = s1[c + j];
    const double sdzxhat = rs * (sdzx - mu * sdz);
    dg[j] = (float)sdzxhat; db[j] = (float)sdz;
    const double aa = g * rs;
    float fa = (float)aa, fb = 0.f, fc = 0.f;
    if (training) {
      const double m1 = sdz * inv_count, m2 = sdzxhat * inv_count;
      const double bb = -aa * rs * m2;
      fb = (float)bb; fc = (float)(-aa * m1 - bb * mu);
    }
    a[j >> 1][j & 1] = fa; b[j >> 1][j & 1] = fb; cc[j >> 1][j & 1] = fc;
  }
  if (blockIdx.y == 0 && rsub == 0) {
    // scalar stores: parameter-gradient tensors are only 4-byte aligned in general
    if (dgamma) { dgamma[c] = dg[0]; dgamma[c + 1] = dg[1]; dgamma[c + 2] = dg[2]; dgamma[c + 3] = dg[3]; }
    if (dbeta) { dbeta[c] = db[0]; dbeta[c + 1] = db[1]; dbeta[c + 2] = db[2]; dbeta[c + 3] = db[3]; }
  }
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)dst_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (int)dz_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)x_bytes, 0x00020000);
  const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
  // per-lane byte offsets of row (m0 + rsub); rows advance by 4 through the scalar offset.  All extents < 4 GiB (launcher).
  constexpr unsigned EZ = BF ? 2u : 4u;                                // bytes per element of dz / dst
  const unsigned vz = ((unsigned)rsub * (unsigned)ldz + (unsigned)c) * EZ, vx = ((unsigned)rsub * (unsigned)ldx + (unsigned)c) * 4u,
                 vd = ((unsigned)rsub * (unsigned)ldd + (unsigned)c) * EZ;
  unsigned sz = (unsigned)m0 * (unsigned)ldz * EZ, sx = (unsigned)m0 * (unsigned)ldx * 4u, sd = (unsigned)m0 * (unsigned)ldd * EZ;
  const unsigned dzs = (unsigned)ldz * 4u * EZ, dxs = (unsigned)ldx * 16u, dds = (unsigned)ldd * 4u * EZ;      // 4 rows
  auto ldz4 = [&](int voff, unsigned soff) -> u32x4e {                // 4 values of dz as float bits
    if constexpr (BF) {
      const u32x2e h = __builtin_amdgcn_raw_buffer_load_b64(rz, voff, (int)soff, 0);
      return u32x4e{h.x << 16, h.x & 0xFFFF0000u, h.y << 16, h.y & 0xFFFF0000u};
    } else {
      return __builtin_amdgcn_raw_buffer_load_b128(rz, voff, (int)soff, 0);
    }
  };
  auto st4 = [&](u32x4e r, int voff, unsigned soff) {
    if constexpr (BF) {
      const bf16x2e lo = {(__bf16)__uint_as_float(r.x), (__bf16)__uint_as_float(r.y)}, hi = {(__bf16)__uint_as_float(r.z), (__bf16)__uint_as_float(r.w)};
      __builtin_amdgcn_raw_buffer_store_b64(u32x2e{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)}, rd, voff, (int)soff, 0);
    } else if constexpr (SP) {
      buffer_store_b128_soffset(split_row4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)), rd, voff, (int)soff);
    } else {
      buffer_store_b128_soffset(r, rd, voff, (int)soff);            // (SGPR soffset: protected form, see rdm_common.h)
    }
  };
  auto one = [&](u32x4e z, u32x4e xv, u32x4e o) {
    f32x2 z0 = {__uint_as_float(z.x), __uint_as_float(z.y)}, z1 = {__uint_as_float(z.z), __uint_as_float(z.w)};
    f32x2 x0 = {__uint_as_float(xv.x), __uint_as_float(xv.y)}, x1 = {__uint_as_float(xv.z), __uint_as_float(xv.w)};
    f32x2 r0 = __builtin_elementwise_fma(a[0], z0, __builtin_elementwise_fma(b[0], x0, cc[0]));
    f32x2 r1 = __builtin_elementwise_fma(a[1], z1, __builtin_elementwise_fma(b[1], x1, cc[1]));
    if (ACC) {
      r0 += (f32x2){__uint_as_float(o.x), __uint_as_float(o.y)};
      r1 += (f32x2){__uint_as_float(o.z), __uint_as_float(o.w)};
    }
    u32x4e r = {__float_as_uint(r0.x), __float_as_uint(r0.y), __float_as_uint(r1.x), __float_as_uint(r1.y)};
    return r;
  };
  int m = m0;
  for (; m + 16 <= m1; m += 16) {                                     // 4 row groups in flight; every lane's row is < m1 here
    u32x4e z[4], xv[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      z[k] = ldz4((int)vz, sz + k * dzs);
      xv[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vx, (int)(sx + k * dxs), 0);
      if (ACC) o[k] = __builtin_amdgcn_raw_buffer_load_b128(rd, (int)vd, (int)(sd + k * dds), 0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) st4(one(z[k], xv[k], o[k]), (int)vd, sd + k * dds);
    sz += 4 * dzs; sx += 4 * dxs; sd += 4 * dds;
  }
  for (; m < m1; m += 4) {                                            // tail: rows >= m1 belong to the next workgroup (or do not exist)
    const bool ok = m + rsub < m1;
    const u32x4e z = ldz4(ok ? (int)vz : -1, sz);
    const u32x4e xv = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (int)vx : -1, (int)sx, 0);
    u32x4e o = {0u, 0u, 0u, 0u};
    if (ACC) o = __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? (int)vd : -1, (int)sd, 0);
    st4(one(z, xv, o), ok ? (int)vd : -1, sd);
    sz += dzs; sx += dxs; sd += dds;
  }
}

int launch_bn_bwd_apply(float* dst, int ldd, const float* dz, int ldz, const float* x, int ldx, const double* s0, const double* s1, double count,
                        const float* gamma, const float* mean, const float* rstd, float* dgamma, float* dbeta, int M, int C, bool accumulate,
                        int training, hipStream_t s, bool bf16_rows, bool split_rows) {
  const int C4 = C / 4, gx = cdiv(C4, 64);
  RDM_CHECK_ARG(!((bf16_rows || split_rows) && accumulate) && !(bf16_rows && split_rows), "bn_bwd: bf16 / split rows are built for the plain (norm2) form, one at a time");
  const long ez = bf16_rows ? 2 : 4;
  const long eb[3] = {((long)(M - 1) * ldd + C) * ez, ((long)(M - 1) * ldz + C) * ez, ((long)(M - 1) * ldx + C) * 4};
  if (eb[0] >= 0xFFFFFFFFL || eb[1] >= 0xFFFFFFFFL || eb[2] >= 0xFFFFFFFFL) {
    set_error("bn_bwd: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing");
    return RDM_ERR_UNSUPPORTED;
  }
  int rpb = cdiv(M, std::max(1, 2048 / gx));
  rpb = std::max(16, (rpb + 15) / 16 * 16);
  dim3 grid(gx, cdiv(M, rpb));
  if (accumulate) hipLaunchKernelGGL(k_bn_bwd_apply<true>, grid, dim3(256), 0, s, dst, ldd, dz, ldz, x, ldx, s0, s1, 1.0 / count, gamma, mean, rstd, dgamma, dbeta, M, C4, rpb, training, (unsigned)eb[0], (unsigned)eb[1], (unsigned)eb[2]);
  else if (bf16_rows) hipLaunchKernelGGL((k_bn_bwd_apply<false, true>), grid, dim3(256), 0, s, dst, ldd, dz, ldz, x, ldx, s0, s1, 1.0 / count, gamma, mean, rstd, dgamma, dbeta, M, C4, rpb, training, (unsigned)eb[0], (unsigned)eb[1], (unsigned)eb[2]);
  else if (split_rows) hipLaunchKernelGGL((k_bn_bwd_apply<false, false, true>), grid, dim3(256), 0, s, dst, ldd, dz, ldz, x, ldx, s0, s1, 1.0 / count, gamma, mean, rstd, dgamma, dbeta, M, C4, rpb, training, (unsigned)eb[0], (unsigned)eb[1], (unsigned)eb[2]);
  else hipLaunchKernelGGL(k_bn_bwd_apply<false>, grid, dim3(256), 0, s, dst, ldd, dz, ldz, x, ldx, s0, s1, 1.0 / count, gamma, mean, rstd, dgamma, dbeta, M, C4, rpb, training, (unsigned)eb[0], (unsigned)eb[1], (unsigned)eb[2]);
  RDM_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Activation -> SPLIT ROWS (xsplit_dev.h): dst[m][c] (row stride ldd floats' worth of bytes) = split(f(src[m][c])), f = ReLU(scale * x + shift) or the
// identity.  The operand producer of the split-precision 1x1 weight gradient: relu1(norm1(x)) of a dense layer (RDM_Net.py:526-530 via
// torchvision _DenseLayer) is formed and split ONCE per layer instead of once per 128-channel gradient tile inside the GEMM.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split_rows(const float* __restrict__ src, int lds_, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    float* __restrict__ dst, int ldd, int M, int C4) {
  // a thread keeps its channel quad (and its BatchNorm coefficients) and walks rows: no division per element
  const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  if (c >= C4 * 4) return;
  f32x4e sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (scale) { sc = *reinterpret_cast<const f32x4e*>(scale + c); sh = *reinterpret_cast<const f32x4e*>(shift + c); }
  for (int m = blockIdx.y * 4 + (threadIdx.x >> 6); m < M; m += gridDim.y * 4) {
    f32x4e v = *reinterpret_cast<const f32x4e*>(src + (long)m * lds_ + c);
    if (scale) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
    }
    *reinterpret_cast<xs_u32x4*>(dst + (long)m * ldd + c) = split_row4(v[0], v[1], v[2], v[3]);
  }
}

int launch_split_rows(const float* src, int ld_src, const float* scale, const float* shift, void* dst, int ld_dst, long M, int C, hipStream_t s) {
  RDM_CHECK_ARG(C % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "split_rows: channels / strides multiples of 4, operands 16-byte aligned");
  RDM_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (((uintptr_t)scale | (uintptr_t)shift) & 15) == 0, "split_rows: scale and shift go together, 16-byte aligned");
  RDM_CHECK_ARG(M > 0 && M < (1L << 31), "split_rows: rows");
  const int gx = cdiv(C / 4, 64);
  const int gy = (int)std::min<long>(cdiv(M, 4), std::max(1, 4096 / gx));
  hipLaunchKernelGGL(k_split_rows, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, src, ld_src, scale, shift, static_cast<float*>(dst), ld_dst, (int)M, C / 4);
  RDM_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Transition front end (RDM_Net.py:527,529,531-532): ZeroPad2d((0,1,0,1)) -> BN -> ReLU -> 2x2
// average.  The 1x1 conv is linear, so it is applied AFTER the pooling (4x fewer GEMM rows).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trans_pool(const float* X, int ldx, const float* sc, const float* sh, float* P, int B, int H,
                                                    int W, int Ho, int Wo, int C4) {
  const long total = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long pix = i / C4;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    const float4 s4 = ld4(sc + c), t4 = ld4(sh + c);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        float4 v = make_float4(0, 0, 0, 0);
        if (y < H && x < W) v = ld4(X + ((long)(b * H + y) * W + x) * ldx + c);
        acc.x += fmaxf(fmaf(v.x, s4.x, t4.x), 0.f);
        acc.y += fmaxf(fmaf(v.y, s4.y, t4.y), 0.f);
        acc.z += fmaxf(fmaf(v.z, s4.z, t4.z), 0.f);
        acc.w += fmaxf(fmaf(v.w, s4.w, t4.w), 0.f);
      }
    acc.x *= 0.25f; acc.y *= 0.25f; acc.z *= 0.25f; acc.w *= 0.25f;
    st4(P + i * 4, acc);
  }
}

int launch_trans_pool(const float* X, int ldx, const float* sc, const float* sh, float* P, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(k_trans_pool, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, X, ldx, sc, sh, P, B, H, W, Ho, Wo, C / 4);
  RDM_LAUNCH_OK();
  return 0;
}

// backward, pass 1: reductions over the PADDED extent (H+1)x(W+1) (pad pixels are BN inputs).
// dz(py,px) = 0.25 * dP(py/2,px/2) * [relu'(z)] when the position is covered by a pooling window.
__global__ __launch_bounds__(256) void k_trans_pool_bwd_reduce(const float* dP, const float* X, int ldx, const float* sc, const float* sh,
                                                               int B, int H, int W, int Ho, int Wo, int C, int rows_per_block,
                                                               double* s0, double* s1) {
  __shared__ float4 red0[256], red1[256];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  const int Hp = H + 1, Wp = W + 1;
  const long Mp = (long)B * Hp * Wp;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(Mp, r0 + rows_per_block);
  float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
  if (c < C) {
    const float4 s4 = ld4(sc + c), t4 = ld4(sh + c);
    for (long m = r0 + ry; m < r1; m += 4) {
      const int px = (int)(m % Wp);
      const long t = m / Wp;
      const int py = (int)(t % Hp), b = (int)(t / Hp);
      if (py >= 2 * Ho || px >= 2 * Wo) continue;
      float4 x = make_float4(0, 0, 0, 0);
      if (py < H && px < W) x = ld4(X + ((long)(b * H + py) * W + px) * ldx + c);
      const float4 g = ld4(dP + ((long)(b * Ho + (py >> 1)) * Wo + (px >> 1)) * C + c);
      float4 dz;
      dz.x = fmaf(x.x, s4.x, t4.x) > 0.f ? 0.25f * g.x : 0.f;
      dz.y = fmaf(x.y, s4.y, t4.y) > 0.f ? 0.25f * g.y : 0.f;
      dz.z = fmaf(x.z, s4.z, t4.z) > 0.f ? 0.25f * g.z : 0.f;
      dz.w = fmaf(x.w, s4.w, t4.w) > 0.f ? 0.25f * g.w : 0.f;
      a0.x += dz.x; a0.y += dz.y; a0.z += dz.z; a0.w += dz.w;
      a1.x += dz.x * x.x; a1.y += dz.y * x.y; a1.z += dz.z * x.z; a1.w += dz.w * x.w;
    }
  }
  red0[threadIdx.x] = a0; red1[threadIdx.x] = a1;
  __syncthreads();
  if (ry == 0 && c < C) {
    for (int k = 1; k < 4; ++k) {
      const float4 b0 = red0[cx + 64 * k], b1 = red1[cx + 64 * k];
      a0.x += b0.x; a0.y += b0.y; a0.z += b0.z; a0.w += b0.w;
      a1.x += b1.x; a1.y += b1.y; a1.z += b1.z; a1.w += b1.w;
    }
    atomicAdd(s0 + c, (double)a0.x); atomicAdd(s0 + c + 1, (double)a0.y); atomicAdd(s0 + c + 2, (double)a0.z); atomicAdd(s0 + c + 3, (double)a0.w);
    atomicAdd(s1 + c, (double)a1.x); atomicAdd(s1 + c + 1, (double)a1.y); atomicAdd(s1 + c + 2, (double)a1.z); atomicAdd(s1 + c + 3, (double)a1.w);
  }
}

// backward, pass 2: G[b,y,x,c] = A*dz + B*x + Cc on the real HxW extent (first writer of G)
__global__ __launch_bounds__(256) void k_trans_pool_bwd_apply(const float* dP, const float* X, int ldx, const float* sc, const float* sh,
                                                              const float* A, const float* Bc, const float* Cc, float* G, int ldg, int B,
                                                              int H, int W, int Ho, int Wo, int C) {
  const int C4 = C / 4;
  const long total = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long pix = i / C4;
    const int x_ = (int)(pix % W); long t = pix / W;
    const int y_ = (int)(t % H), b = (int)(t / H);
    const float4 x = ld4(X + pix * ldx + c);
    const float4 s4 = ld4(sc + c), t4 = ld4(sh + c);
    float4 dz = make_float4(0, 0, 0, 0);
    if (y_ < 2 * Ho && x_ < 2 * Wo) {
      const float4 g = ld4(dP + ((long)(b * Ho + (y_ >> 1)) * Wo + (x_ >> 1)) * C + c);
      dz.x = fmaf(x.x, s4.x, t4.x) > 0.f ? 0.25f * g.x : 0.f;
      dz.y = fmaf(x.y, s4.y, t4.y) > 0.f ? 0.25f * g.y : 0.f;
      dz.z = fmaf(x.z, s4.z, t4.z) > 0.f ? 0.25f * g.z : 0.f;
      dz.w = fmaf(x.w, s4.w, t4.w) > 0.f ? 0.25f * g.w : 0.f;
    }
    const float4 a = ld4(A + c), bb = ld4(Bc + c), cc = ld4(Cc + c);
    float4 r;
    r.x = fmaf(a.x, dz.x, fmaf(bb.x, x.x, cc.x));
    r.y = fmaf(a.y, dz.y, fmaf(bb.y, x.y, cc.y));
    r.z = fmaf(a.z, dz.z, fmaf(bb.z, x.z, cc.z));
    r.w = fmaf(a.w, dz.w, fmaf(bb.w, x.w, cc.w));
    st4(G + pix * ldg + c, r);
  }
}

int launch_trans_pool_bwd_reduce(const float* dP, const float* X, int ldx, const float* sc, const float* sh, int B, int H, int W, int C,
                                 double* s0, double* s1, hipStream_t s) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long Mp = (long)B * (H + 1) * (W + 1);
  const int rpb = rows_per_block(Mp, C);
  dim3 grid(cdiv(C / 4, 64), cdiv(Mp, rpb));
  hipLaunchKernelGGL(k_trans_pool_bwd_reduce, grid, dim3(256), 0, s, dP, X, ldx, sc, sh, B, H, W, Ho, Wo, C, rpb, s0, s1);
  RDM_LAUNCH_OK();
  return 0;
}

int launch_trans_pool_bwd_apply(const float* dP, const float* X, int ldx, const float* sc, const float* sh, const float* A,
                                const float* Bc, const float* Cc, float* G, int ldg, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(k_trans_pool_bwd_apply, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, dP, X, ldx, sc, sh, A, Bc, Cc,
                     G, ldg, B, H, W, Ho, Wo, C);
  RDM_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// stem: im2col for the 7x7/s2/p3 conv (RDM_Net.py:524), 3x3/s2/p1 max-pool (:525)
// patches[m][k], k = c*49 + r*7 + s (PyTorch weight order), zero for k >= 147, row length 160
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_im2col_stem(const float* x, float* patches, int B, int H, int W, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * 160;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % 160);
    long m = i / 160;
    float v = 0.f;
    if (k < 147) {
      const int ox = (int)(m % Wo); long t = m / Wo;
      const int oy = (int)(t % Ho), b = (int)(t / Ho);
      const int c = k / 49, rs = k - c * 49, r = rs / 7, q = rs - r * 7;
      const int iy = oy * 2 - 3 + r, ix = ox * 2 - 3 + q;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((long)(b * 3 + c) * H + iy) * W + ix];
    }
    patches[i] = v;
  }
}

int launch_im2col_stem(const float* x, float* patches, int B, int H, int W, hipStream_t s) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const long total = (long)B * Ho * Wo * 160;
  hipLaunchKernelGGL(k_im2col_stem, dim3((int)std::min<long>(cdiv(total, 256), 8192)), dim3(256), 0, s, x, patches, B, H, W, Ho, Wo);
  RDM_LAUNCH_OK();
  return 0;
}

__global__ __launch_bounds__(256) void k_maxpool3s2(const float* X, float* Y, int ldy, unsigned char* arg, int B, int H, int W, int Ho, int Wo, int C4) {
  const long total = (long)B * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long pix = i / C4;
    const int ox = (int)(pix % Wo); long t = pix / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 bi = make_uchar4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int iy = 2 * oy - 1 + r, ix = 2 * ox - 1 + q;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const float4 v = ld4(X + ((long)(b * H + iy) * W + ix) * (C4 * 4) + c);
        const unsigned char t9 = (unsigned char)(r * 3 + q);
        if (v.x > best.x) { best.x = v.x; bi.x = t9; }
        if (v.y > best.y) { best.y = v.y; bi.y = t9; }
        if (v.z > best.z) { best.z = v.z; bi.z = t9; }
        if (v.w > best.w) { best.w = v.w; bi.w = t9; }
      }
    st4(Y + pix * ldy + c, best);
    *reinterpret_cast<uchar4*>(arg + pix * (C4 * 4) + c) = bi;
  }
}

int launch_maxpool3s2(const float* X, float* Y, int ldy, unsigned char* arg, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(k_maxpool3s2, dim3((int)std::min<long>(cdiv(total, 256), 8192)), dim3(256), 0, s, X, Y, ldy, arg, B, H, W, Ho, Wo, C / 4);
  RDM_LAUNCH_OK();
  return 0;
}

// gather form of the max-pool backward (no atomics): each input pixel collects from the <= 4 windows holding it
__global__ __launch_bounds__(256) void k_maxpool3s2_bwd(const float* Gy, int ldg, const unsigned char* arg, float* Gx, int B, int H, int W,
                                                        int Ho, int Wo, int C4) {
  const long total = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long pix = i / C4;
    const int ix = (int)(pix % W); long t = pix / W;
    const int iy = (int)(t % H), b = (int)(t / H);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int oy = max(0, iy / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy)
      for (int ox = max(0, ix / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
        const int r = iy - (2 * oy - 1), q = ix - (2 * ox - 1);
        if (r < 0 || r > 2 || q < 0 || q > 2) continue;
        const unsigned char t9 = (unsigned char)(r * 3 + q);
        const long op = (long)(b * Ho + oy) * Wo + ox;
        const uchar4 a = *reinterpret_cast<const uchar4*>(arg + op * (C4 * 4) + c);
        const float4 g = ld4(Gy + op * ldg + c);
        if (a.x == t9) acc.x += g.x;
        if (a.y == t9) acc.y += g.y;
        if (a.z == t9) acc.z += g.z;
        if (a.w == t9) acc.w += g.w;
      }
    st4(Gx + pix * (C4 * 4) + c, acc);
  }
}

int launch_maxpool3s2_bwd(const float* Gy, int ldg, const unsigned char* arg, float* Gx, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(k_maxpool3s2_bwd, dim3((int)std::min<long>(cdiv(total, 256), 8192)), dim3(256), 0, s, Gy, ldg, arg, Gx, B, H, W, Ho, Wo, C / 4);
  RDM_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// layout helpers
// ------------------------------------------------------------------------------------------
__global__ void k_pack_w(const float* w, float* wp, int O, int I, int T, int Opad) {
  const long total = (long)T * Opad * I;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % I); long t = i / I;
    const int o = (int)(t % Opad), tap = (int)(t / Opad);
    wp[i] = o < O ? w[((long)o * I + c) * T + tap] : 0.f;
  }
}
__global__ void k_unpack_w(const float* wp, float* w, int O, int I, int T, int Opad) {
  const long total = (long)O * I * T;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int tap = (int)(i % T); long t = i / T;
    const int c = (int)(t % I), o = (int)(t / I);
    w[i] = wp[((long)tap * Opad + o) * I + c];
  }
}
int launch_pack_w(const float* w, float* wp, int O, int I, int T, int Opad, hipStream_t s) {
  const long total = (long)T * Opad * I;
  hipLaunchKernelGGL(k_pack_w, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, wp, O, I, T, Opad);
  RDM_LAUNCH_OK();
  return 0;
}
int launch_unpack_w(const float* wp, float* w, int O, int I, int T, int Opad, hipStream_t s) {
  const long total = (long)O * I * T;
  hipLaunchKernelGGL(k_unpack_w, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, wp, w, O, I, T, Opad);
  RDM_LAUNCH_OK();
  return 0;
}

// [M][ld] (first C columns) -> (B,C,HW) and back (zero-filling columns C..ld-1)
__global__ void k_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int C, int HW) {
  const long total = (long)B * C * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int p = (int)(i % HW); long t = i / HW;
    const int c = (int)(t % C), b = (int)(t / C);
    dst[i] = src[((long)b * HW + p) * ld + c];
  }
}
__global__ void k_nchw_to_nhwc(const float* src, float* dst, int ld, int B, int C, int HW) {
  const long total = (long)B * HW * ld;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % ld); long m = i / ld;
    const int p = (int)(m % HW), b = (int)(m / HW);
    dst[i] = c < C ? src[((long)b * C + c) * HW + p] : 0.f;
  }
}
int launch_nhwc_to_nchw(const float* src, int ld, float* dst, int B, int C, int HW, hipStream_t s) {
  const long total = (long)B * C * HW;
  hipLaunchKernelGGL(k_nhwc_to_nchw, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, src, ld, dst, B, C, HW);
  RDM_LAUNCH_OK();
  return 0;
}
int launch_nchw_to_nhwc(const float* src, float* dst, int ld, int B, int C, int HW, hipStream_t s) {
  const long total = (long)B * HW * ld;
  hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((int)std::min<long>(cdiv(total, 256), 4096)), dim3(256), 0, s, src, dst, ld, B, C, HW);
  RDM_LAUNCH_OK();
  return 0;
}

__global__ void k_f64_to_f32(const double* src, float* dst, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
int launch_f64_to_f32(const double* src, float* dst, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_f64_to_f32, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, n);
  RDM_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// fused AdamW over a flat parameter buffer (torch.optim.AdamW semantics, module.py:41)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adamw(float* p, const float* g, float* m, float* v, long n4, long n, float lr, float b1, float b2,
                                               float eps, float wd, float bc1, float rsqrt_bc2, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = ld4(p + i * 4), gg = ld4(g + i * 4), mm = ld4(m + i * 4), vv = ld4(v + i * 4);
    float* P = &pp.x; float* Gv = &gg.x; float* Mv = &mm.x; float* Vv = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = Gv[k] * gscale;
      float w = P[k] * (1.f - lr * wd);
      Mv[k] = b1 * Mv[k] + (1.f - b1) * gr;
      Vv[k] = b2 * Vv[k] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(Vv[k]) * rsqrt_bc2 + eps;
      w -= (lr / bc1) * (Mv[k] / denom);
      P[k] = w;
    }
    st4(p + i * 4, pp); st4(m + i * 4, mm); st4(v + i * 4, vv);
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = n4 * 4 + threadIdx.x;
    const float gr = g[i] * gscale;
    float w = p[i] * (1.f - lr * wd);
    m[i] = b1 * m[i] + (1.f - b1) * gr;
    v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
    w -= (lr / bc1) * (m[i] / (sqrtf(v[i]) * rsqrt_bc2 + eps));
    p[i] = w;
  }
}

int launch_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float wd, int step,
                 float gscale, hipStream_t s) {
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  const long n4 = n / 4;
  hipLaunchKernelGGL(k_adamw, dim3((int)std::min<long>(std::max<long>(cdiv(n4, 256), 1), 256 * 8)), dim3(256), 0, s, p, g, m, v, n4, n, lr, b1, b2,
                     eps, wd, (float)bc1, (float)(1.0 / sqrt(bc2)), gscale);
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
