// Reduced-precision FORWARD of the conv stack (BASELINE config 2: "NYU-v2 batch=8 forward-only bf16").
//
// The reference's default is mixed precision (train.py:11 `--precision 16`, :57-58 AMP O2: fp16 convs, fp32 BatchNorm);
// its inference pass is network/module.py:49-56 -> network/RDM_Net.py:70-103.  Here the same pass runs with
//   * bf16 activations and weights in HBM (half the bytes of every block buffer / bottleneck tensor),
//   * v_mfma_f32_16x16x32_bf16 with f32 accumulation (16x the f32-MFMA rate),
//   * f32 BatchNorm affines applied in the conv staging registers (eval mode: running statistics, folded once per weight
//     update by rdm_net_bf16_prepare), f32 bias, f32 logits; the f64 DORN tail is unchanged.
// Two kernels carry > 99 % of the work:
//   gemm_bf16_kernel     1x1 convs (and the im2col'd stem): C[m][n] = sum_k f(X[m][k]) * W[n][k], both operands k-contiguous
//   conv3x3_bf16_kernel  the 3x3 / pad 1 / 48-output conv of every dense layer on an LDS halo tile (each activation is
//                        staged and normalised ONCE per 9 taps)
// Operand roles are swapped w.r.t. the f32 kernels: the WEIGHT fragment is the MFMA "A" operand and the ACTIVATION fragment
// the "B" operand, so a lane's 4 accumulator registers are 4 consecutive output CHANNELS of one pixel: the bf16 result is
// written as one 8-byte store per lane (32 contiguous bytes per pixel and n-tile) instead of 2-byte scattered stores.
// LDS images are [row][k] with 16-byte chunks XOR-swizzled so that every ds_read_b128 of a fragment is bank-conflict free
// for the 16-lane groups the hardware forms (MI355X_MICROARCH.md, LDS): 128-byte rows: chunk ^= (row >> 1) & 7;
// 64-byte rows read at arbitrary row offsets (halo taps): chunk ^= ((row >> 2) & 1) << 1   (both checked exhaustively).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "rdm_common.h"
#include "elementwise.h"
#include "bf16.h"

namespace rdm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFFFu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 bld(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 bldf(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 r = {(__bf16)a, (__bf16)b};                  // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned bnrelu2(unsigned u, float s0, float s1, float t0, float t1) {
  return pack2(fmaxf(fmaf(bf_lo(u), s0, t0), 0.f), fmaxf(fmaf(bf_hi(u), s1, t1), 0.f));
}
// relu(x * scale + shift) on 8 packed bf16 (channels k .. k+7), f32 arithmetic, one rounding back to bf16
__device__ __forceinline__ uint4 bnrelu8(uint4 v, float4 sa, float4 sb, float4 ta, float4 tb) {
  v.x = bnrelu2(v.x, sa.x, sa.y, ta.x, ta.y);
  v.y = bnrelu2(v.y, sa.z, sa.w, ta.z, ta.w);
  v.z = bnrelu2(v.z, sb.x, sb.y, tb.x, tb.y);
  v.w = bnrelu2(v.w, sb.z, sb.w, tb.z, tb.w);
  return v;
}

__device__ __forceinline__ void xcd_order(int& bx, int& by) {      // see igemm.hip xcd_block_order
  const unsigned gx = gridDim.x, total = gx * gridDim.y;
  const unsigned L = blockIdx.x + gx * blockIdx.y;
  const unsigned x = L & 7u, seq = L >> 3, q = total >> 3, r = total & 7u;
  const unsigned Lp = x * q + (x < r ? x : r) + seq;
  bx = (int)(Lp % gx); by = (int)(Lp / gx);
}

// ---------------------------------------------------------------------------------------------
// GEMM: out[m][n] = bias[n] + sum_k f(X[m][k]) * W[n][k]      f = relu(x*scale[k]+shift[k]) or identity
// Block = 4 wave64s (WM x WN), wave tile (MT*16 pixels) x (NT*16 channels), K walked in 64-deep steps (2 MFMA k-steps).
// global -> registers (next step's loads in flight under this step's MFMAs) -> BN-ReLU -> swizzled LDS, 2 buffers, 1 barrier / step.
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int BK, int NS, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmBf16Args p) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN;
  constexpr int CH = BK / 8, RP = 256 / CH;                          // 16-byte chunks per row; rows covered by one pass of the 256 threads
  constexpr int XL = (BM + RP - 1) / RP, WL = (BN + RP - 1) / RP;    // chunks per thread and step
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * (BM + BN) * BK];
  unsigned short* const Xs0 = lds;
  unsigned short* const Ws0 = lds + 2 * BM * BK;
  // conflict-free chunk swizzles (header): 128-byte rows (BK 64): c ^ ((row >> 1) & 7); 64-byte rows (BK 32): c ^ (((row >> 2) & 1) << 1)
  auto swz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : (((row >> 2) & 1) << 1); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave / WN) * MT * 16, wcol = (wave % WN) * NT * 16;
  int bx, by;
  xcd_order(bx, by);
  const int n0 = bx * BN, m0 = by * BM;
  const __amdgpu_buffer_rsrc_t srdX = srd(p.X, p.x_bytes), srdW = srd(p.W, p.w_bytes);
  const __amdgpu_buffer_rsrc_t srdS = srd(p.scale, p.p_bytes), srdT = srd(p.shift, p.p_bytes);
  const bool bnrelu = p.scale != nullptr;

  const int ch = tid % CH, r0 = tid / CH;                             // this thread's chunk column and first row
  unsigned xoff[XL], woff[WL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int r = r0 + RP * i, m = m0 + r;
    xoff[i] = (r < BM && m < p.M) ? (unsigned)m * (unsigned)(p.ldx * 2) + (unsigned)(ch * 16) : OOB;
  }
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int r = r0 + RP * i, n = n0 + r;
    woff[i] = (r < BN && n < p.N) ? (unsigned)n * (unsigned)(p.ldw * 2) + (unsigned)(ch * 16) : OOB;
  }
  // NS register stages: the loads of step k+NS are issued while step k is multiplied, so every load has NS full steps
  // (NS x [MFMAs + LDS staging + barrier]) to land.  With 16x the f32 MFMA rate a 64-deep step of a small tile is ~300 cycles of
  // work - far shorter than an L2 / HBM round trip - and a shallow prefetch leaves the kernel waiting on memory latency (measured
  // ~1 us per step at M = 2280 whatever the tile).  Small tiles (few registers per stage) take 4 stages, 128 x 96 two.
  struct Stage { uint4 rx[XL]; uint4 rw[WL]; float4 sa, sb, ta, tb; };
  Stage S[NS];
  auto load_step = [&](int kt, Stage& S) {
    const int k0 = kt * BK + ch * 8;
    const bool kok = k0 < p.K;                                        // K is a multiple of 8: a chunk is all in or all out
    const unsigned kb = (unsigned)(kt * BK * 2);
#pragma unroll
    for (int i = 0; i < XL; ++i) S.rx[i] = bld(srdX, (kok && xoff[i] != OOB) ? xoff[i] + kb : OOB);
#pragma unroll
    for (int i = 0; i < WL; ++i) S.rw[i] = bld(srdW, (kok && woff[i] != OOB) ? woff[i] + kb : OOB);
    if (bnrelu) {
      const unsigned po = kok ? (unsigned)(k0 * 4) : OOB;
      S.sa = bldf(srdS, po); S.sb = bldf(srdS, po == OOB ? OOB : po + 16);
      S.ta = bldf(srdT, po); S.tb = bldf(srdT, po == OOB ? OOB : po + 16);
    }
  };
  auto store_step = [&](int buf, const Stage& S) {
    unsigned short* Xs = Xs0 + buf * BM * BK;
    unsigned short* Ws = Ws0 + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int r = r0 + RP * i;
      if (r < BM) {
        uint4 v = S.rx[i];
        if (bnrelu) v = bnrelu8(v, S.sa, S.sb, S.ta, S.tb);
        *reinterpret_cast<uint4*>(Xs + r * BK + ((ch ^ swz(r)) << 3)) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int r = r0 + RP * i;
      if (r < BN) *reinterpret_cast<uint4*>(Ws + r * BK + ((ch ^ swz(r)) << 3)) = S.rw[i];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K-split over grid.z (few-pixel layers): this block owns the 64-deep steps [kbase, kbase + nk)
  const int nk_all = (p.K + BK - 1) / BK, per_z = (nk_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int kbase = blockIdx.z * per_z, nk = min(nk_all, kbase + per_z) - kbase;
  const int sw = swz(l16);                                            // tile row bases are multiples of 16
  auto mma_step = [&](int buf) {
    const unsigned short* Xs = Xs0 + buf * BM * BK;
    const unsigned short* Ws = Ws0 + buf * BN * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int co = ((ks * 4 + g) ^ sw) << 3;
      bf16x8 wf[NT], xf[MT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(Ws + (wcol + j * 16 + l16) * BK + co);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(Xs + (wrow + i * 16 + l16) * BK + co);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
  };
#pragma unroll
  for (int u = 0; u < NS; ++u)
    if (u < nk) load_step(kbase + u, S[u]);                           // step j lives in stage j % NS
  store_step(0, S[0]);
  if (NS < nk) load_step(kbase + NS, S[0]);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int k = kt + u;                                           // the step multiplied now, in LDS buffer k & 1
      if (k < nk) {
        mma_step(k & 1);
        if (k + 1 < nk) store_step((k + 1) & 1, S[(u + 1) % NS]);     // step k+1, loaded NS steps ago
        if (k + 1 + NS < nk) load_step(kbase + k + 1 + NS, S[(u + 1) % NS]);
        __syncthreads();
      }
    }
  }
  if (gridDim.z > 1) {                                                // partial sums -> this split's f32 slab (reduced in a fixed order by k_reduce_partials_bf16)
    float* slab = p.partial + (size_t)blockIdx.z * p.M * p.N;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wcol + j * 16 + g * 4;
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + wrow + i * 16 + l16;
        if (m < p.M) *reinterpret_cast<float4*>(slab + (size_t)m * p.N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }

  // D[row = channel 4g+r of the n-tile][col = pixel l16 of the m-tile]
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + g * 4;
    if (n >= p.N) continue;                                            // N is a multiple of 4
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), os = make_float4(1.f, 1.f, 1.f, 1.f), oh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + n);
    const bool oact = p.oscale != nullptr;
    if (oact) { os = *reinterpret_cast<const float4*>(p.oscale + n); oh = *reinterpret_cast<const float4*>(p.oshift + n); }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wrow + i * 16 + l16;
      if (m >= p.M) continue;
      float v0 = acc[i][j][0] + b4.x, v1 = acc[i][j][1] + b4.y, v2 = acc[i][j][2] + b4.z, v3 = acc[i][j][3] + b4.w;
      if (oact) { v0 = fmaxf(fmaf(v0, os.x, oh.x), 0.f); v1 = fmaxf(fmaf(v1, os.y, oh.y), 0.f); v2 = fmaxf(fmaf(v2, os.z, oh.z), 0.f); v3 = fmaxf(fmaf(v3, os.w, oh.w), 0.f); }
      if (OUT_F32) {
        *reinterpret_cast<float4*>(static_cast<float*>(p.out) + (long)m * p.ldc + n) = make_float4(v0, v1, v2, v3);
      } else {
        *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p.out) + (long)m * p.ldc + n) = make_uint2(pack2(v0, v1), pack2(v2, v3));
      }
    }
  }
}

// one 1-KiB LDS-DMA piece: lane l's 16 bytes at (voff + soff) land at lds_dst + 16*l (lds_dst wave-uniform: it goes to M0)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned char* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, (int)soff, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------
// Short-K 1x1 with many pixels and many outputs (dense_e2: M = 34 656, N = 2752, K = 96 .. 336): persistent PANEL GEMM.
// The tiled kernel above pays a workgroup's fixed costs (launch, address set-up, first-load latency, drain) 7859 times per layer for
// 1.5 - 5 K-steps of work each, and every one of the 29 column tiles of a row panel re-loads AND re-normalises the same activations.
// Here one persistent workgroup of 8 waves per CU walks a contiguous range of (256-row panel, 96-column tile) items, panel-major:
//   * a wave owns 32 rows x 96 columns; its rows' activations live in REGISTERS as MFMA B fragments ([m-tile 2][32-deep k group
//     <= 12] x 16 B per lane), loaded and BN-ReLU'd ONCE per panel (~29 items);
//   * the item's weight tile ([K/64 steps][96 n][64 k] bf16, <= 72 KiB) comes by LDS-DMA, double-buffered by ITEM: the tile of item
//     i+1 is requested right after the barrier that opens item i, so ONE barrier per item orders everything (the loads of item i have
//     landed, every wave is past item i-1);  the consumer's BatchNorm coefficients of the 96 columns ride along as one more piece each;
//   * the epilogue issues a FIXED number of stores (masked lanes store out of range), so the counted vmcnt wait stays exact:
//     behind an item's DMA pieces a wave has issued 12 stores - nothing else;
// LDS image of a K-step: [96 rows][128 B], 16-byte chunks XOR-swizzled (chunk ^ ((row >> 1) & 7)) on the DMA's source side.
// (First version: 128-row panels, 4 consumer + 4 loader waves, an 8-stage ring with a barrier per K-step: 131 us at K = 240 where the
// tiled kernel takes 150 - the weight stream alone, 377 MB through L2 -> LDS per launch, took 70 us.  256-row panels halve it.)
// ---------------------------------------------------------------------------------------------
constexpr int PANEL_STEP = 96 * 128, PANEL_EPI = 2048;
#ifdef RDM_DEV_VARIANTS
#define RDM_STAMP(v) const long v = __builtin_readcyclecounter()
#else
#define RDM_STAMP(v)
#endif

template <int NKK>            // 32-deep K half-steps per item: 32 (NKK - 1) < K <= 32 NKK (compile-time: the multiply loop is one straight-line block)
__global__ __launch_bounds__(512, 2) void gemm_panel_bf16_kernel(GemmBf16Args p) {
  constexpr int NKS = (NKK + 1) / 2;      // 64-deep K-steps (DMA granularity)
  constexpr int BM = 256, BN = 96, NPIECE = NKS * 12, PW = (NPIECE + 7) / 8, BUF = NKS * PANEL_STEP;     // LDS: 2 weight buffers, then 4 coefficient slots
  constexpr int NST = 12;                                             // store instructions per item and wave
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, g = lane >> 4;
  const int ntn = (p.N + BN - 1) / BN, npan = (p.M + BM - 1) / BM;
  const long items = (long)npan * ntn;
  const int it0 = (int)((long)blockIdx.x * items / gridDim.x), it1 = (int)((long)(blockIdx.x + 1) * items / gridDim.x);
  if (it0 == it1) return;
  const __amdgpu_buffer_rsrc_t srdO = srd(p.out, p.o_bytes);
  const __amdgpu_buffer_rsrc_t srdX = srd(p.X, p.x_bytes);
  const __amdgpu_buffer_rsrc_t srdE = srd(wave == 0 ? p.oscale : p.oshift, (unsigned)(p.N * 4));
  const bool bnrelu = p.scale != nullptr, oact = p.oscale != nullptr;

  // DMA pieces of this wave: piece q = wave + 8*j of the item's NKS*12 (step = q / 12, 8 rows x 128 B each); the last column tile is ragged
  const __amdgpu_buffer_rsrc_t srdW = srd(p.W, p.w_bytes);
  unsigned voff[PW], lastmask = 0;                                    // bit j: piece j's row exists in the (ragged) last column tile
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int q = wave + 8 * j, ks = q / 12, r = (q - ks * 12) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    voff[j] = q < NPIECE ? (unsigned)r * (unsigned)(p.ldw * 2) + (unsigned)(ks * 128 + c * 16) : OOB;
    if ((ntn - 1) * BN + r < p.N) lastmask |= 1u << j;
  }
  // (a pre-tiled weight image - every piece 1 KiB contiguous instead of 8 row segments - was tried: 86.5 vs 89.3 us, not worth a second copy)
  const unsigned evoff = lane < 24 ? (unsigned)(lane * 16) : OOB;     // 96 floats of the column tile's scale (wave 0) / shift (wave 1)
  auto request = [&](int item) {                                      // the weight tile (+ epilogue coefficients) of `item` -> buffer item & 1
    const int t = item % ntn;
    unsigned char* const dst = smem + (item & 1) * BUF;
    const unsigned so = (unsigned)(t * BN) * (unsigned)(p.ldw * 2);
    const bool last = t == ntn - 1;
#pragma unroll
    for (int j = 0; j < PW; ++j)
      if (wave + 8 * j < NPIECE) dma16(srdW, (!last || ((lastmask >> j) & 1)) ? voff[j] : OOB, so, dst + (wave + 8 * j) * 1024);
    if (wave < 2) dma16(srdE, (oact && !(last && (ntn - 1) * BN + (int)(lane * 4) >= p.N)) ? evoff : OOB, (unsigned)(t * BN * 4), smem + 2 * BUF + (item & 3) * PANEL_EPI + wave * 1024);
  };

  bf16x8 xr[2][NKK];
  f32x4 acc[2][6];
  const unsigned sw = (unsigned)((l16 >> 1) & 7);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto mma = [&](int item) {                                            // acc = the item's 32 x 96 products
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* const wrd = smem + (item & 1) * BUF + l16 * 128;   // weight fragment of column tile j, k-group kc: + j*2048 + ((kc ^ sw) << 4)
    bf16x8 wf[2][6];                                                    // fragments of half-step kk+1 are read while kk multiplies (scheduler pinned)
    auto frags = [&](int set, int kk) {
#pragma unroll
      for (int j = 0; j < 6; ++j) wf[set][j] = *reinterpret_cast<const bf16x8*>(wrd + (kk >> 1) * PANEL_STEP + j * 2048 + ((((unsigned)((kk & 1) * 4 + g)) ^ sw) << 4));
    };
#ifdef RDM_DEV_VARIANTS
    if (p.abl & 4) return;
#endif
    constexpr bool DB = true;                                            // (long K: the activations take the registers of the second fragment set)
    frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      if (DB) { if (kk + 1 < NKK) frags((kk + 1) & 1, kk + 1); __builtin_amdgcn_sched_barrier(0); }
      else if (kk > 0) frags(0, kk);
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[DB ? (kk & 1) : 0][j], xr[i][kk], acc[i][j], 0, 0, 0);
      if (DB) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // epilogue: D[row = channel 4g+r of the n-tile][col = pixel l16 of the m-tile]; ALWAYS 12 store instructions per wave .
  // The kernel is bound by vector-instruction ISSUE (in-kernel stamps: matrix-pipe time + VALU + DMA issue of a SIMD's two waves add
  // up), so the epilogue is kept short: packed f32 FMAs, ReLU as a packed signed-16-bit max on the bf16 pairs (a negative bf16 is a
  // negative int16), the column offset in the store's scalar offset, one out-of-range select per row instead of one per store.
  // (Tried without effect on the 88 us of the K = 240 layer: the second wave of each SIMD running its epilogue one item late; a per-wave
  // LDS transpose to 16-byte stores of 192 contiguous bytes per row; a pre-tiled weight image with 1-KiB contiguous DMA pieces.)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short i16x2 __attribute__((ext_vector_type(2)));
  auto epilogue = [&](int m0, int n0, int slot) {
    const float* const ep = reinterpret_cast<const float*>(smem + 2 * BUF + (slot & 3) * PANEL_EPI);
    const bool ragged = n0 + BN > p.N;                                  // (uniform) the last column tile: per-lane column checks
    unsigned rowoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + i * 16 + l16;
      bool ok = slot >= 0 && m < p.M;
#ifdef RDM_DEV_VARIANTS
      if (p.abl & 1) ok = false;
#endif
      rowoff[i] = ok ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(n0 + g * 4)) * 2u : OOB;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x4 os = f32x4{1.f, 1.f, 1.f, 1.f}, oh = f32x4{0.f, 0.f, 0.f, 0.f};
      if (oact && slot >= 0) { os = *reinterpret_cast<const f32x4*>(ep + j * 16 + g * 4); oh = *reinterpret_cast<const f32x4*>(ep + 256 + j * 16 + g * 4); }
      const f32x2 os01 = {os[0], os[1]}, os23 = {os[2], os[3]}, oh01 = {oh[0], oh[1]}, oh23 = {oh[2], oh[3]};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x2 a01 = {acc[i][j][0], acc[i][j][1]}, a23 = {acc[i][j][2], acc[i][j][3]};
        if (oact) { a01 = __builtin_elementwise_fma(a01, os01, oh01); a23 = __builtin_elementwise_fma(a23, os23, oh23); }
        unsigned p01 = pack2(a01[0], a01[1]), p23 = pack2(a23[0], a23[1]);
        if (oact) {
          p01 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, p01), i16x2{0, 0}));
          p23 = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, p23), i16x2{0, 0}));
        }
        const u32x2 pk = {p01, p23};
        unsigned off = rowoff[i];
        if (ragged && n0 + j * 16 + g * 4 >= p.N) off = OOB;
        __builtin_amdgcn_raw_buffer_store_b64(pk, srdO, (int)off, j * 32, 0);   // (the column tile offset rides in soffset: it is outside the range check)
      }
    }
  };
  // the producer-side BatchNorm affine of the K input channels, once per workgroup: [scale NKK*32][shift NKK*32] floats behind the coefficient slots
  float* const coef = reinterpret_cast<float*>(smem + 2 * BUF + 4 * PANEL_EPI);
  if (bnrelu)
    for (int c = tid; c < NKK * 32; c += 512) { coef[c] = c < p.K ? p.scale[c] : 0.f; coef[NKK * 32 + c] = c < p.K ? p.shift[c] : 0.f; }
  __syncthreads();
  int cur_pan = -1;
  request(it0);
  for (int item = it0; item < it1; ++item) {
    const int pan = item / ntn, t = item - pan * ntn;
    const int m0 = pan * BM + wave * 32, n0 = t * BN;
    if (pan != cur_pan) {                                               // a new row panel: its activations, normalised once, as B fragments
      cur_pan = pan;
      // ALL loads first, one wait, then the affine (coefficients staged in LDS once per workgroup): written as the obvious loop - load,
      // convert, next - the compiler waits for every load in turn and a reload is a chain of 2 NKK memory round trips
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int m = m0 + i * 16 + l16, k0 = kk * 32 + g * 8;
          const uint4 v = bld(srdX, (k0 < p.K && m < p.M) ? (unsigned)m * (unsigned)(p.ldx * 2) + (unsigned)(k0 * 2) : OOB);   // K is a multiple of 8; past K: zeros
          xr[i][kk] = __builtin_bit_cast(bf16x8, v);
        }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (also: the counted wait below assumes nothing but pieces and stores in flight)
      if (bnrelu) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          const int k0 = kk * 32 + g * 8;
          const float4 sa = *reinterpret_cast<const float4*>(coef + k0), sb = *reinterpret_cast<const float4*>(coef + k0 + 4);
          const float4 ta = *reinterpret_cast<const float4*>(coef + NKK * 32 + k0), tb = *reinterpret_cast<const float4*>(coef + NKK * 32 + k0 + 4);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            uint4 v = bnrelu8(__builtin_bit_cast(uint4, xr[i][kk]), sa, sb, ta, tb);
            if (k0 >= p.K) v = make_uint4(0, 0, 0, 0);                  // columns past K multiply (finite) weights of the next row: force exact zeros
            xr[i][kk] = __builtin_bit_cast(bf16x8, v);
          }
        }
      }
    }
    // this item's pieces have landed: behind them this wave issued exactly the previous item's 12 stores (nothing before the first item)
    RDM_STAMP(t0);
    if (item == it0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    RDM_STAMP(t1);
    __builtin_amdgcn_s_barrier();                                       // ... everybody's have; every wave is past the other buffer (item - 1)
    asm volatile("" ::: "memory");
    RDM_STAMP(t2);
    if (item + 1 < it1) request(item + 1);
    RDM_STAMP(t3);
    mma(item);
    RDM_STAMP(t4);
    epilogue(m0, n0, item);
#ifdef RDM_DEV_VARIANTS
    if ((p.abl & 8) && p.partial && blockIdx.x == 3 && lane == 0) {     // tools/panel_gemm_stamps.py: cycles per phase, per wave of one workgroup
      RDM_STAMP(t5);
      float* d = p.partial + wave * 8;
      d[0] += (float)(t1 - t0); d[1] += (float)(t2 - t1); d[2] += (float)(t3 - t2); d[3] += (float)(t4 - t3); d[4] += (float)(t5 - t4); d[5] += 1.f;
    }
#endif
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 / 48 outputs on a zero-PADDED LDS image.  A block owns BM = MT*64 consecutive output pixels (linear
// NHWC index).  Its input neighbourhood is staged as a 2-D image of padded rows: every image row gets a zero column on both
// sides, every image a zero row above and below (padded row index = b*(H+2) + y + 1), so ALL nine taps of EVERY output pixel -
// image borders and batch boundaries included - are plain reads at (row + r - 1, col + q - 1): no validity masks in the MFMA loop
// (they were 6 VALU ops per fragment, 216 per slab, in a loop that is bound by instruction issue: 5.75 VALU per MFMA measured).
// The pad slots are zeroed once per block and never written again; per 32-channel slab only the real pixels are loaded
// (BN-ReLU applied once per element, in the staging registers) together with the 9 x 48 x 32 weight slab.  The LDS byte address of
// every (m-tile, tap) fragment is computed once per block (9*MT VGPRs); the next slab's global loads are in flight under the
// 9 x 3 x MT MFMAs of this one (register prefetch), 2 barriers per slab.
// Dynamic LDS: [padded rows x (W+2)][32] bf16 + [9*48][32] bf16.
// ---------------------------------------------------------------------------------------------
template <int MT, int HL>      // HL >= ceil(image slots * 4 / 256): 16-byte chunks per thread and slab
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(Conv3Bf16Args p) {
  constexpr int BM = MT * 64, NT = 3, CS = 32, WLN = 7;              // 9*48*4 = 1728 weight chunks / 256 threads = 6.75
  extern __shared__ __attribute__((aligned(16))) unsigned short dyn[];
  const int W = p.W, H = p.H, Wp = W + 2, Hp = H + 2;
  unsigned short* const Ah = dyn;
  unsigned short* const Wl = dyn + (size_t)p.slots * CS;              // p.slots: image slots the launcher sized the LDS for (multiple of 8)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = wave * MT * 16;
  const int m0 = blockIdx.x * BM;
  const __amdgpu_buffer_rsrc_t srdY = srd(p.Y, p.y_bytes), srdW = srd(p.Wt, p.w_bytes);
  const __amdgpu_buffer_rsrc_t srdS = srd(p.scale, p.p_bytes), srdT = srd(p.shift, p.p_bytes);

  auto prow_of = [&](int m, int& x) {                                 // padded row (global over the batch) and column of pixel m
    const int hw = H * W, b = m / hw, rem = m - b * hw, y = rem / W;
    x = rem - y * W;
    return b * Hp + y + 1;
  };
  int xdummy;
  const int m_last = min(p.M, m0 + BM) - 1;
  const int pr_lo = prow_of(m0, xdummy) - 1, pr_hi = prow_of(m_last, xdummy) + 1;
  const int nslots = (pr_hi - pr_lo + 1) * Wp;                        // <= p.slots (launcher bound)

  // zero the image once: pad columns / rows (and out-of-batch rows) are never written afterwards
  for (int i = tid; i < p.slots * (CS / 8); i += 256) *reinterpret_cast<uint4*>(Ah + i * 8) = make_uint4(0, 0, 0, 0);

  // LDS byte address of the A fragment of (m-tile i, tap): slot of the lane's pixel shifted by (r-1, q-1), chunk g swizzled
  unsigned faddr[MT][9];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    int x;
    const int m = min(m0 + wrow + i * 16 + l16, p.M - 1);             // rows past M are never stored: any in-range address will do
    const int sc = (prow_of(m, x) - pr_lo) * Wp + x + 1;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int sl = sc + (t / 3 - 1) * Wp + (t % 3 - 1);
      faddr[i][t] = (unsigned)(sl * (CS * 2) + ((g ^ (((sl >> 2) & 1) << 1)) << 4));
    }
  }
  const int ch = tid & 3;                                             // chunk (8 channels) of the 32-channel slab, same for all of a thread's loads
  unsigned hoff[HL], hlds[HL];
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int sl = (tid + 256 * i) >> 2;
    hoff[i] = OOB; hlds[i] = 0;
    if (sl < nslots) {
      const int prow = sl / Wp, pcol = sl - prow * Wp, pr = pr_lo + prow;
      const int b = pr / Hp, yy = pr - b * Hp - 1, x = pcol - 1;
      if (pr >= 0 && b < p.B && (unsigned)yy < (unsigned)H && (unsigned)x < (unsigned)W) {
        hoff[i] = (unsigned)((b * H + yy) * W + x) * (unsigned)(p.ldy * 2) + (unsigned)(ch * 16);
        hlds[i] = (unsigned)(sl * CS + ((ch ^ (((sl >> 2) & 1) << 1)) << 3));
      }
    }
  }
  unsigned woff[WLN];
#pragma unroll
  for (int i = 0; i < WLN; ++i) {
    const int row = (tid + 256 * i) >> 2;                             // tap*48 + n
    woff[i] = row < 9 * 48 ? (unsigned)(row / 48) * (unsigned)(p.wtap * 2) + (unsigned)(row % 48) * (unsigned)(p.ldw * 2) + (unsigned)(ch * 16) : OOB;
  }
  uint4 rh[HL], rw[WLN];
  float4 sa, sb, ta, tb;
  const bool bn = p.scale != nullptr;                                 // NULL: the input is already activated (the producer's epilogue did it)
  auto load_slab = [&](int cs) {
    const unsigned cb = (unsigned)(cs * CS * 2);
    // bottleneck widths are 48 * odd: the last slab holds 16 channels.  Chunks past C are forced to zero on BOTH operands (the
    // activation row continues with the next pixel there) and their affine to 0, so relu(0 * 0 + 0) * 0 contributes nothing.
    const bool kok = cs * CS + ch * 8 < p.C;
#pragma unroll
    for (int i = 0; i < HL; ++i) rh[i] = bld(srdY, (kok && hoff[i] != OOB) ? hoff[i] + cb : OOB);
#pragma unroll
    for (int i = 0; i < WLN; ++i) rw[i] = bld(srdW, (kok && woff[i] != OOB) ? woff[i] + cb : OOB);
    if (bn) {
      const unsigned po = kok ? (unsigned)((cs * CS + ch * 8) * 4) : OOB;
      sa = bldf(srdS, po); sb = bldf(srdS, po == OOB ? OOB : po + 16);
      ta = bldf(srdT, po); tb = bldf(srdT, po == OOB ? OOB : po + 16);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int i = 0; i < HL; ++i)
      if (hoff[i] != OOB) *reinterpret_cast<uint4*>(Ah + hlds[i]) = bn ? bnrelu8(rh[i], sa, sb, ta, tb) : rh[i];      // real pixels only: pads stay zero
#pragma unroll
    for (int i = 0; i < WLN; ++i) {
      const int row = (tid + 256 * i) >> 2;
      if (row < 9 * 48) *reinterpret_cast<uint4*>(Wl + row * CS + ((ch ^ (((row >> 2) & 1) << 1)) << 3)) = rw[i];
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K-split over whole channel slabs: grid.y splits, each owns a contiguous slab range (the launcher guarantees none is empty)
  const int ncs_all = (p.C + CS - 1) / CS;
  const int per = (ncs_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int cs0 = blockIdx.y * per, ncs = min(ncs_all, cs0 + per);
  load_slab(cs0);
  __syncthreads();                                                    // the zero fill is complete before the first real pixels land
  store_slab();
  __syncthreads();
  const int wsw = ((l16 >> 2) & 1) << 1;                              // weight rows of an n-tile start at a multiple of 16
  const char* const AhB = reinterpret_cast<const char*>(Ah);
  for (int cs = cs0; cs < ncs; ++cs) {
    const bool more = cs + 1 < ncs;
    if (more) load_slab(cs + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      bf16x8 wf[NT], xf[MT];
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(Wl + (tap * 48 + j * 16 + l16) * CS + ((g ^ wsw) << 3));
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(AhB + faddr[i][tap]);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                                  // every wave is past its last read of this slab
    if (more) { store_slab(); __syncthreads(); }
  }
  if (gridDim.y > 1) {                                                // partial sums: plain f32 stores into this split's slab (no atomics: deterministic)
    float* slab = p.partial + (size_t)blockIdx.y * p.M * 48;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = j * 16 + g * 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = m0 + wrow + i * 16 + l16;
        if (m < p.M) *reinterpret_cast<float4*>(slab + (size_t)m * 48 + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = j * 16 + g * 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wrow + i * 16 + l16;
      if (m < p.M)
        *reinterpret_cast<uint2*>(p.out + (long)m * p.ldc + n) = make_uint2(pack2(acc[i][j][0], acc[i][j][1]), pack2(acc[i][j][2], acc[i][j][3]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 / 48 outputs on an ALREADY ACTIVATED input (the producing 1x1 applied BatchNorm + ReLU in its epilogue:
// eval mode knows the affine ahead of time), both operands by LDS-DMA, loader / consumer wave specialisation.
//   * no prologue  => no staging registers, no BN-ReLU VALU work, no ds_write: `buffer_load_dwordx4 ... lds` moves 1-KiB pieces
//     (64 lanes x 16 B) straight into the LDS image; zero padding = out-of-range source offsets (the DMA writes zeros);
//   * a workgroup = NC consumer waves + NC loader waves (one of each per SIMD at NC = 4) and owns BM = NC*128 consecutive pixels of
//     ONE image (tiles never straddle images).  A consumer holds a 128 x 48 output tile (96 accumulator registers) and does nothing
//     but ds_read_b128 + MFMA; a loader does nothing but issue DMAs (an LDS-DMA costs its wave 60-180 issue cycles - measured: with
//     every wave loading AND multiplying the two costs added up, 3.0 us per slab where the MFMAs alone take 2.1);
//   * per 32-channel slab: the zero-padded image rows of the tile ([slot][32 ch] bf16, 16-byte chunks XOR-swizzled on the SOURCE
//     side as in conv3x3_bf16_kernel) and the slab's weights in MFMA-FRAGMENT order ([tap][n-tile][lane][8 bf16]: one 1-KiB piece
//     per fragment, read back conflict-free at lane*16);
//   * 2 stages at FIXED LDS offsets (image 0 | image 1 | weights 0 | weights 1 = 159 KB: the stage is an instruction immediate),
//     ONE barrier per slab: loaders wait for their DMAs -> barrier -> loaders issue slab s+1, consumers multiply slab s;
//   * K-split over grid.y: partial sums to f32 slabs; the LAST workgroup of a tile to arrive (device-scope ticket) adds them in
//     ascending split order - no second launch, no atomics on data, same bits every run.
// C must be a multiple of 32 (the network pads its bottleneck widths; pad channels are zero on both operands).
// ---------------------------------------------------------------------------------------------
constexpr int ACT_IMG_BYTES = 52 * 1024, ACT_W_BYTES = 27 * 1024, ACT_MAX_SLOTS = ACT_IMG_BYTES / 64;     // 832 padded pixels
constexpr int ACT_LDS_BYTES = 2 * (ACT_IMG_BYTES + ACT_W_BYTES);

template <int NC>
__global__ __launch_bounds__(NC * 128, 2) void conv3x3_act_bf16_kernel(Conv3ActArgs p) {
  constexpr int MT = 8, NT = 3, BM = NC * 128, WPIECES = 27, NTHR = NC * 128;
  constexpr int HP = (ACT_MAX_SLOTS / 16 + NC - 1) / NC;              // image pieces per loader wave (13 at NC = 4)
  constexpr int WP = (WPIECES + NC - 1) / NC;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l16 = lane & 15, g = lane >> 4;   // (uniform: LDS-DMA targets go to M0)
  const int W = p.W, H = p.H, HW = H * W;
  const int tile = blockIdx.x, b = tile / p.tiles_per_img, t = tile - b * p.tiles_per_img;
  // Tile geometry.  Narrow rows (p.rect == 0): BM consecutive pixels of image b in row-major order - whole rows, ragged at both ends.
  // Wide rows (the padded rows of such a tile would not fit the LDS image, e.g. 304 pixels at 352x1216): a RECTANGLE of BM / 64 rows x 64
  // columns.  Either way: local pixel ml -> image (y, x); LDS image = rows y0-1 .. , columns x0-1 .. x0+Wt, Wp = Wt + 2 slots per row.
  constexpr int RTW = 64;
  const bool rect = p.rect != 0;
  const int tiles_x = rect ? (W + RTW - 1) / RTW : 1;
  const int ml0 = rect ? 0 : t * BM, mlN = rect ? BM : min(HW, ml0 + BM);
  const int y0 = rect ? (t / tiles_x) * (BM / RTW) : ml0 / W, x0 = rect ? (t % tiles_x) * RTW : 0;
  const int Wt = rect ? RTW : W, Wp = Wt + 2;
  const int nrows = rect ? BM / RTW + 2 : (mlN - 1) / W - y0 + 3;
  const int nslots = nrows * Wp, npieces = (nslots + 15) >> 4;        // <= ACT_MAX_SLOTS / 16 (launcher)
  auto pixel = [&](int ml, int& y, int& x) -> bool {                  // image coordinates of local pixel ml; false: not an output of this tile
    if (rect) { y = y0 + (ml >> 6); x = x0 + (ml & (RTW - 1)); return y < H && x < W; }
    const int gl = ml0 + ml; y = gl / W; x = gl - y * W; return gl < mlN;
  };
  const int ncs_all = p.C >> 5, split = (int)gridDim.y;
  const int per = (ncs_all + split - 1) / split;
  const int cs0 = blockIdx.y * per, cs1 = min(ncs_all, cs0 + per);   // the launcher leaves no split empty
  const bool loader = wave >= NC;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (loader) {
    const int lw = wave - NC;
    const __amdgpu_buffer_rsrc_t srdY = srd(p.Y, p.y_bytes), srdW = srd(p.Wimg, p.w_bytes);
    // image pieces of this wave: piece q = lw + i*NC covers slots 16q .. 16q+15; lane -> (slot, LDS chunk); the SOURCE chunk is swizzled
    unsigned voff[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
      const int sl = (lw + i * NC) * 16 + (lane >> 2), c = (lane & 3) ^ (((sl >> 2) & 1) << 1);
      const int prow = sl / Wp, x = x0 + sl - prow * Wp - 1, y = y0 - 1 + prow;
      const bool ok = sl < nslots && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      voff[i] = ok ? (unsigned)((b * H + y) * W + x) * (unsigned)(p.ldy * 2) + (unsigned)(c * 16) : OOB;
    }
    const unsigned wvoff = (unsigned)(lane * 16);
    auto issue = [&](int st, int cs) {
      unsigned char* const ib = smem + st * ACT_IMG_BYTES;
      unsigned char* const wb = smem + 2 * ACT_IMG_BYTES + st * ACT_W_BYTES;
      const unsigned so = (unsigned)cs * 64u, wso = (unsigned)cs * (unsigned)ACT_W_BYTES;
#ifdef RDM_DEV_VARIANTS
      if (!(p.abl & 1))
#endif
#pragma unroll
      for (int i = 0; i < HP; ++i)
        if (lw + i * NC < npieces) dma16(srdY, voff[i], so, ib + (lw + i * NC) * 1024);
#ifdef RDM_DEV_VARIANTS
      if (!(p.abl & 2))
#endif
#pragma unroll
      for (int q = 0; q < WP; ++q)
        if (lw + q * NC < WPIECES) dma16(srdW, wvoff + (unsigned)((lw + q * NC) * 1024), wso, wb + (lw + q * NC) * 1024);
    };
    issue(0, cs0);
    for (int cs = cs0; cs < cs1; ++cs) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's pieces of slab cs have landed (nothing younger is in flight)
      __builtin_amdgcn_s_barrier();                                   // ... and everybody's; the consumers are past their reads of the other stage
      asm volatile("" ::: "memory");
      if (cs + 1 < cs1) issue(((cs - cs0) & 1) ^ 1, cs + 1);
    }
  } else {
    // LDS address (stage 0, in 16-byte units: 13 bits) of the activation fragment of (m-tile i, tap): slot of the lane's pixel shifted
    // by the tap; two taps per register (96 accumulators + 72 full addresses + the fragment sets would spill), unpacked by one
    // shift-of-a-halfword per read
    unsigned fpk[MT][5];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      int y, x;
      pixel(min(wave * 128 + i * 16 + l16, mlN - ml0 - 1), y, x);     // rows past the tile are never stored: any in-range address will do
      const int sc = (y - y0 + 1) * Wp + (x - x0) + 1;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int sl = sc + (tp / 3 - 1) * Wp + (tp % 3 - 1);
        const unsigned a16 = (unsigned)(sl * 4 + (g ^ (((sl >> 2) & 1) << 1)));
        if (tp & 1) fpk[i][tp >> 1] |= a16 << 16; else fpk[i][tp >> 1] = a16;
      }
    }
    int ya, xa;
    const bool active = pixel(wave * 128, ya, xa);                    // ragged last tile of an image: idle consumers only synchronise
    // one slab from the stage at compile-time LDS offsets: 18 steps (tap, half of the 8 m-tiles) of 12 MFMAs; the fragments of step
    // k+1 are read while step k multiplies (two register sets; the scheduler is pinned, it would otherwise fold the sets and wait on every read)
    auto slab = [&](auto stc) {
      constexpr int ST = decltype(stc)::value;
      const unsigned char* const ib = smem + ST * ACT_IMG_BYTES;
      const unsigned char* const wb = smem + 2 * ACT_IMG_BYTES + ST * ACT_W_BYTES + lane * 16;
      bf16x8 wf[2][NT], xf[2][4];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 5; ++q) asm volatile("" : "+v"(fpk[i][q]));   // (keeps the 72 unpacked addresses from being hoisted out of the slab loop into registers)
      auto wfrags = [&](int set, int tp) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[set][j] = *reinterpret_cast<const bf16x8*>(wb + (tp * 3 + j) * 1024);
      };
      auto xfrags = [&](int set, int tp, int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[set][i] = *reinterpret_cast<const bf16x8*>(ib + (((fpk[h * 4 + i][tp >> 1] >> ((tp & 1) * 16)) & 0xFFFFu) << 4));
      };
      wfrags(0, 0); xfrags(0, 0, 0);
#pragma unroll
      for (int k = 0; k < 18; ++k) {
        const int tp = k >> 1, h = k & 1;
        if (k < 17) {
          if (h == 1) wfrags((tp + 1) & 1, tp + 1);
          xfrags((k + 1) & 1, (k + 1) >> 1, (k + 1) & 1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tp & 1][j], xf[k & 1][i], acc[h * 4 + i][j], 0, 0, 0);
        // one wave per SIMD: the next step's address unpacks and fragment reads go BETWEEN this step's MFMAs (an MFMA holds the
        // vector issue for half of its 16 cycles; issued as a block in front they cost 0.8 us per slab of MFMA idle time)
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#ifdef RDM_DEV_VARIANTS
    const bool work = active && !(p.abl & 4);
#else
    const bool work = active;
#endif
    for (int cs = cs0; cs < cs1; cs += 2) {                            // stages alternate in straight-line code (accumulation stays in place)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (work) slab(std::integral_constant<int, 0>{});
      if (cs + 1 < cs1) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (work) slab(std::integral_constant<int, 1>{});
      }
    }
  }

  const long pixb = (long)b * HW;
  int ys, xs;
  const bool store = !loader && pixel(wave * 128, ys, xs);
  if (split == 1) {
    if (!store) return;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = j * 16 + g * 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        int y, x;
        if (pixel(wave * 128 + i * 16 + l16, y, x))
          *reinterpret_cast<uint2*>(p.out + (pixb + (long)y * W + x) * p.ldc + n) = make_uint2(pack2(acc[i][j][0], acc[i][j][1]), pack2(acc[i][j][2], acc[i][j][3]));
      }
    }
    return;
  }
  // K-split: this workgroup's partial sums, then the last arrival of the tile adds the `split` slabs in ascending order.
  // Coherence across the 8 XCD L2s WITHOUT cache-wide fences (a __threadfence pair per workgroup writes back and invalidates a whole
  // L2 each time: measured 65 us of a 170 us launch): the partial sums are written and read with agent-scope (sc1) accesses, which
  // go through to the memory side; ordering = this wave's stores acknowledged (vmcnt(0)) -> workgroup barrier -> ticket (device-scope
  // atomic at the memory side) -> the last workgroup's sc1 loads.
  const __amdgpu_buffer_rsrc_t srdP = srd(p.partial + (size_t)tile * split * (size_t)(BM * 48), (unsigned)(split * BM * 48 * 4));
  if (store) {
    const unsigned mine = (unsigned)blockIdx.y * (unsigned)(BM * 48 * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = j * 16 + g * 4;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int r = wave * 128 + i * 16 + l16;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), srdP, (int)(mine + (unsigned)(r * 48 + n) * 4u), 0, 16);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const flag = reinterpret_cast<int*>(smem);                     // every wave is past its last fragment read
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(p.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = ticket == (unsigned)(split - 1);
    if (ticket == (unsigned)(split - 1)) __hip_atomic_store(p.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
  }
  __syncthreads();
  if (!*flag) return;
  // every thread owns 12 float4 of the tile (BM*12 / NTHR); a slab element comes from the memory side (~1.5 us): all loads of 6
  // elements x 6 slabs are issued before the first add - 2 round trips for a split <= 6 instead of 12 x split
  constexpr int G = 6, ZB = 5;
#pragma unroll
  for (int k0 = 0; k0 < 12; k0 += G) {
    f32x4 a[G];
    unsigned off[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int idx = tid + (k0 + k) * NTHR, r = idx / 12;
      int y, x;
      off[k] = pixel(r, y, x) ? (unsigned)idx * 16u : OOB;            // idx*16 = (r*48 + q4*4)*4
      a[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int z0 = 0; z0 < split; z0 += ZB) {
      f32x4 v[G][ZB];
#pragma unroll
      for (int z = 0; z < ZB; ++z)
#pragma unroll
        for (int k = 0; k < G; ++k)
          v[k][z] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdP, (int)((z0 + z < split && off[k] != OOB) ? off[k] + (unsigned)(z0 + z) * (unsigned)(BM * 48 * 4) : OOB), 0, 16));
#pragma unroll
      for (int z = 0; z < ZB; ++z)                                    // ascending split order; slabs past `split` read as +0
#pragma unroll
        for (int k = 0; k < G; ++k) { a[k][0] += v[k][z][0]; a[k][1] += v[k][z][1]; a[k][2] += v[k][z][2]; a[k][3] += v[k][z][3]; }
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const int idx = tid + (k0 + k) * NTHR, r = idx / 12, q4 = idx - r * 12;
      int y, x;
      if (pixel(r, y, x)) *reinterpret_cast<uint2*>(p.out + (pixb + (long)y * W + x) * p.ldc + q4 * 4) = make_uint2(pack2(a[k][0], a[k][1]), pack2(a[k][2], a[k][3]));
    }
  }
}

// 3x3 weights -> the fragment-order image conv3x3_act_bf16_kernel streams: img[slab][tap][n-tile j][lane][e] = w[n = 16j + (lane & 15)]
// [c = 32 slab + 8 (lane >> 4) + e][tap]; c >= C -> 0.  Source: PyTorch OIHW [48][C][9] (packed = 0) or [9][48][C] (packed = 1).
__global__ __launch_bounds__(256) void k_pack_w3_frag_bf16(const float* __restrict__ w, unsigned short* __restrict__ img, int C, int Cpad, int packed) {
  const long total = (long)(Cpad / 32) * 27 * 64 * 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long f = i >> 9;                                                   // (slab*9 + tap)*3 + j
    const int j = (int)(f % 3); f /= 3;
    const int tap = (int)(f % 9), slab = (int)(f / 9);
    const int n = j * 16 + (lane & 15), c = slab * 32 + (lane >> 4) * 8 + e;
    float v = 0.f;
    if (c < C) v = packed ? w[((long)tap * 48 + n) * C + c] : w[((long)n * C + c) * 9 + tap];
    img[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

// out[m][n .. n+3] = bf16(sum_s partial[s][m][n .. n+3]) in a fixed order (s ascending): the reduction of the K-split
__global__ __launch_bounds__(256) void k_reduce_partials_bf16(const float* __restrict__ partial, int split, long MN4, int n4, unsigned short* __restrict__ out, int ldc,
                                                              const float* __restrict__ oscale, const float* __restrict__ oshift) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (long)gridDim.x * 256) {
    float4 a = *reinterpret_cast<const float4*>(partial + i * 4);
    for (int s = 1; s < split; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(partial + ((size_t)s * MN4 + i) * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const long m = i / n4; const int n = (int)(i - m * n4) * 4;
    if (oscale) {
      const float4 os = *reinterpret_cast<const float4*>(oscale + n), oh = *reinterpret_cast<const float4*>(oshift + n);
      a.x = fmaxf(fmaf(a.x, os.x, oh.x), 0.f); a.y = fmaxf(fmaf(a.y, os.y, oh.y), 0.f); a.z = fmaxf(fmaf(a.z, os.z, oh.z), 0.f); a.w = fmaxf(fmaf(a.w, os.w, oh.w), 0.f);
    }
    *reinterpret_cast<uint2*>(out + m * ldc + n) = make_uint2(pack2(a.x, a.y), pack2(a.z, a.w));
  }
}

// ---------------------------------------------------------------------------------------------
// helper kernels (HBM bound)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_f32_to_bf16_rows(const float* __restrict__ src, int lds_, unsigned short* __restrict__ dst, int ldd, long rows, int cols, int cols_pad) {
  const long total = rows * cols_pad;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cols_pad; const int c = (int)(i - r * cols_pad);
    const float v = c < cols ? src[r * lds_ + c] : 0.f;
    dst[r * ldd + c] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}
// [O][I][T] (PyTorch OIHW, T = kh*kw) -> [T][O][ld] bf16 (columns I .. ld-1 are left alone: the caller zeroed them)
__global__ __launch_bounds__(256) void k_pack_w_bf16(const float* __restrict__ w, unsigned short* __restrict__ wp, int O, int I, int ld, int T) {
  const long total = (long)T * O * I;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % I); const long t = i / I;
    const int o = (int)(t % O), tap = (int)(t / O);
    wp[t * ld + c] = __builtin_bit_cast(unsigned short, (__bf16)w[((long)o * I + c) * T + tap]);
  }
}
// im2col of the 7x7/s2/p3 stem (RDM_Net.py:524): patches[m][k], k = c*49 + r*7 + s, zero for k >= 147, row length 160, bf16
__global__ __launch_bounds__(256) void k_im2col_stem_bf16(const float* __restrict__ x, unsigned short* __restrict__ patches, int B, int H, int W, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * 80;                          // two k per thread: one 4-byte store
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k2 = (int)(i % 80) * 2;
    long m = i / 80;
    const int ox = (int)(m % Wo); long t = m / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k2 + e;
      v[e] = 0.f;
      if (k < 147) {
        const int c = k / 49, rs = k - c * 49, r = rs / 7, q = rs - r * 7;
        const int iy = oy * 2 - 3 + r, ix = ox * 2 - 3 + q;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = x[((long)(b * 3 + c) * H + iy) * W + ix];
      }
    }
    *reinterpret_cast<unsigned*>(patches + m * 160 + k2) = pack2(v[0], v[1]);
  }
}
// nn.MaxPool2d(3, 2, 1) on bf16 NHWC (RDM_Net.py:525): 8 channels per thread
__global__ __launch_bounds__(256) void k_maxpool3s2_bf16(const unsigned short* __restrict__ X, unsigned short* __restrict__ Y, int ldy, int B, int H, int W, int Ho, int Wo, int C8) {
  const long total = (long)B * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    long pix = i / C8;
    const int ox = (int)(pix % Wo); long t = pix / Wo;
    const int oy = (int)(t % Ho), b = (int)(t / Ho);
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int iy = 2 * oy - 1 + r, ix = 2 * ox - 1 + q;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(X + ((long)(b * H + iy) * W + ix) * (C8 * 8) + c);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { best[2 * e] = fmaxf(best[2 * e], bf_lo(u[e])); best[2 * e + 1] = fmaxf(best[2 * e + 1], bf_hi(u[e])); }
      }
    *reinterpret_cast<uint4*>(Y + pix * ldy + c) = make_uint4(pack2(best[0], best[1]), pack2(best[2], best[3]), pack2(best[4], best[5]), pack2(best[6], best[7]));
  }
}
// transition front end (RDM_Net.py:527,529,531-532): pad_br -> BN -> ReLU -> 2x2 average, bf16 in / bf16 out, 8 channels per thread
__global__ __launch_bounds__(256) void k_trans_pool_bf16(const unsigned short* __restrict__ X, int ldx, const float* __restrict__ sc, const float* __restrict__ sh,
                                                         unsigned short* __restrict__ P, int B, int H, int W, int Ho, int Wo, int C8) {
  const long total = (long)B * Ho * Wo * C8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    long pix = i / C8;
    const int ox = (int)(pix % Wo); pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int b = (int)(pix / Ho);
    float s[8], t[8], acc[8];
    *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(sc + c); *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(sc + c + 4);
    *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(sh + c); *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(sh + c + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int y = 2 * oy + dy, x = 2 * ox + dx;
        uint4 v = make_uint4(0, 0, 0, 0);                              // the zero pad row / column is a BatchNorm INPUT
        if (y < H && x < W) v = *reinterpret_cast<const uint4*>(X + ((long)(b * H + y) * W + x) * ldx + c);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += fmaxf(fmaf(bf_lo(u[e]), s[2 * e], t[2 * e]), 0.f);
          acc[2 * e + 1] += fmaxf(fmaf(bf_hi(u[e]), s[2 * e + 1], t[2 * e + 1]), 0.f);
        }
      }
    *reinterpret_cast<uint4*>(P + i * 8) = make_uint4(pack2(0.25f * acc[0], 0.25f * acc[1]), pack2(0.25f * acc[2], 0.25f * acc[3]),
                                                       pack2(0.25f * acc[4], 0.25f * acc[5]), pack2(0.25f * acc[6], 0.25f * acc[7]));
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int launch_f32_to_bf16_rows(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, int cols_pad, hipStream_t s) {
  const long total = rows * cols_pad;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(k_f32_to_bf16_rows, dim3((unsigned)std::min<long>(cdiv(total, 256), 8192)), dim3(256), 0, s, src, ld_src, static_cast<unsigned short*>(dst), ld_dst, rows, cols, cols_pad);
  RDM_LAUNCH_OK();
  return 0;
}
int launch_pack_w_bf16(const float* w, void* wp, int O, int I, int ld, int T, hipStream_t s) {
  const long total = (long)T * O * I;
  hipLaunchKernelGGL(k_pack_w_bf16, dim3((unsigned)std::min<long>(cdiv(total, 256), 8192)), dim3(256), 0, s, w, static_cast<unsigned short*>(wp), O, I, ld, T);
  RDM_LAUNCH_OK();
  return 0;
}
int launch_im2col_stem_bf16(const float* x, void* patches, int B, int H, int W, hipStream_t s) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const long total = (long)B * Ho * Wo * 80;
  hipLaunchKernelGGL(k_im2col_stem_bf16, dim3((unsigned)std::min<long>(cdiv(total, 256), 16384)), dim3(256), 0, s, x, static_cast<unsigned short*>(patches), B, H, W, Ho, Wo);
  RDM_LAUNCH_OK();
  return 0;
}
int launch_maxpool3s2_bf16(const void* X, void* Y, int ldy, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(k_maxpool3s2_bf16, dim3((unsigned)std::min<long>(cdiv(total, 256), 16384)), dim3(256), 0, s, static_cast<const unsigned short*>(X),
                     static_cast<unsigned short*>(Y), ldy, B, H, W, Ho, Wo, C / 8);
  RDM_LAUNCH_OK();
  return 0;
}
int launch_trans_pool_bf16(const void* X, int ldx, const float* sc, const float* sh, void* P, int B, int H, int W, int C, hipStream_t s) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(k_trans_pool_bf16, dim3((unsigned)std::min<long>(cdiv(total, 256), 16384)), dim3(256), 0, s, static_cast<const unsigned short*>(X), ldx, sc, sh,
                     static_cast<unsigned short*>(P), B, H, W, Ho, Wo, C / 8);
  RDM_LAUNCH_OK();
  return 0;
}

int launch_gemm_bf16(const GemmBf16Args& a_in, bool out_f32, hipStream_t s) {
  GemmBf16Args a = a_in;
  RDM_CHECK_ARG(a.K > 0 && a.K % 8 == 0 && a.ldx % 8 == 0 && a.ldw % 8 == 0, "gemm_bf16: K (%d) and the row strides must be multiples of 8", a.K);
  RDM_CHECK_ARG(a.N % 4 == 0 && a.ldc % 4 == 0, "gemm_bf16: N (%d) and ldc must be multiples of 4", a.N);
  RDM_CHECK_ARG((((uintptr_t)a.X | (uintptr_t)a.W | (uintptr_t)a.out) & 15) == 0, "gemm_bf16: operands must be 16-byte aligned");
  RDM_CHECK_ARG((a.scale == nullptr) == (a.shift == nullptr) && (((uintptr_t)a.scale | (uintptr_t)a.shift | (uintptr_t)a.bias) & 15) == 0, "gemm_bf16: bad prologue / bias pointers");
  RDM_CHECK_ARG((a.oscale == nullptr) == (a.oshift == nullptr) && (((uintptr_t)a.oscale | (uintptr_t)a.oshift) & 15) == 0 && !(a.oscale && out_f32), "gemm_bf16: bad output-activation pointers");
  const long xb = ((long)(a.M - 1) * a.ldx + a.K) * 2, wb = ((long)(a.N - 1) * a.ldw + a.K) * 2;
  if (xb >= 0xFFFFFFFFL || wb >= 0xFFFFFFFFL) { set_error("gemm_bf16: operand extent >= 4 GiB"); return RDM_ERR_UNSUPPORTED; }
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb; a.p_bytes = (unsigned)(a.K * 4);
  // tiles: 128 x 96 on big grids whose N is (nearly) a multiple of 96, 128 x 48 else; few pixels: 64 x 48 / 32 x 96 (more workgroups)
  const long t96 = (long)cdiv(a.M, 128) * cdiv(a.N, 96);
  const bool n96 = (double)cdiv(a.N, 96) * 96 <= 1.04 * a.N;
  // Wave quantisation: 512 workgroups are resident (2 per CU).  dense_e3 (M = 8816, N = 1408) has 1035 tiles of 128 x 96 - three rounds,
  // the third with 11 workgroups - but 897 of 128 x 112 (4 waves x 32 rows x 112 columns: 56 accumulators): two.  Cost = rounds x tile area.
  const long t112 = (long)cdiv(a.M, 128) * cdiv(a.N, 112);
  bool wide112 = cdiv(t112, 512L) * 112 < cdiv(t96, 512L) * 96 && t96 <= 4096;
#ifdef RDM_DEV_VARIANTS
  if (g_variant == 304) wide112 = false;
#endif
  // few pixels (decoder: 640 rows) and a long K: split K over grid.z so the chip is not left to ~80 workgroups
  int split = 1;
  const long small_blocks = (long)cdiv(a.M, 32) * cdiv(a.N, 96);
  const int nk64 = cdiv(a.K, 64);
  if (a.partial && !out_f32 && !a.bias && a.M <= 1024 && small_blocks < 256 && nk64 >= 8) {
    split = (int)std::min<long>(std::min<long>(nk64 / 4, cdiv(512, small_blocks)), (long)(a.partial_floats / ((size_t)a.M * a.N)));
    if (split < 1) split = 1;
    split = cdiv(nk64, cdiv(nk64, split));                                // no empty split
  }
#define RDM_G(MT_, NT_, WM_, WN_, BK_, NS_)                                                                                               \
  do {                                                                                                                                      \
    dim3 grid(cdiv(a.N, NT_ * 16 * WN_), cdiv(a.M, MT_ * 16 * WM_), split);                                                                 \
    if (out_f32) hipLaunchKernelGGL((gemm_bf16_kernel<MT_, NT_, WM_, WN_, BK_, NS_, true>), grid, dim3(256), 0, s, a);                      \
    else hipLaunchKernelGGL((gemm_bf16_kernel<MT_, NT_, WM_, WN_, BK_, NS_, false>), grid, dim3(256), 0, s, a);                             \
    RDM_CENSUS("gemm_bf16_kernel/%dx%d%s", MT_ * 16 * WM_, NT_ * 16 * WN_, split > 1 ? "/splitK" : "");                                     \
  } while (0)
  // (measured at M = 2280: 64x48 / 64x96 / 128x48 / 128x96 tiles and 2 vs 4 register stages all land at 21-26 us - the kernel is
  // bound by instruction issue, ~140 non-MFMA instructions per 64-deep step of which the BN-ReLU staging transform is the largest part)
  // many pixels x many outputs x short K (dense_e2): the persistent panel kernel
#ifdef RDM_DEV_VARIANTS
  a.abl = g_variant >= 200 && g_variant < 216 ? g_variant - 200 : 0;
#endif
  // (at dense_e3's 525 items - two per workgroup - it only ties the tiled kernel: 37.3 vs 37.0 us at K = 336)
  const bool panel = !out_f32 && !a.bias && split == 1 && a.K <= 352 && a.M >= 8192 && a.N >= 1024 && (long)cdiv(a.M, 256) * cdiv(a.N, 96) >= 1024;
  void* tk = profile_begin(s, 2.0 * a.M * a.N * (double)a.K, panel ? 11 : 7, 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (double)a.M * a.N * (out_f32 ? 4 : 2));
  if (panel) {
    // per call, for the CURRENT device (a process may drive several GPUs, from several threads): the attribute query is a cached host-side lookup
    int n_cu = 0;
    { int dev = 0; RDM_HIP_OK(hipGetDevice(&dev)); RDM_HIP_OK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev)); }
    const int nkk = cdiv(a.K, 32), nks = (nkk + 1) / 2, lds = 2 * nks * PANEL_STEP + 4 * PANEL_EPI + nkk * 256;
    const long ob = ((long)(a.M - 1) * a.ldc + a.N) * 2;
    if (ob >= 0xFFFFFFFFL) { set_error("gemm_bf16: output extent >= 4 GiB"); return RDM_ERR_UNSUPPORTED; }
    a.o_bytes = (unsigned)ob;
    dim3 grid((unsigned)n_cu);
#define RDM_GP(NKK_)                                                                                                                      \
  case NKK_: {                                                                                                                             \
    /* set on every launch: the attribute is per device and per function, the call is a cheap host-side store */                          \
    RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_bf16_kernel<NKK_>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ((NKK_ + 1) / 2) * PANEL_STEP + 4 * PANEL_EPI + NKK_ * 256)); \
    hipLaunchKernelGGL((gemm_panel_bf16_kernel<NKK_>), grid, dim3(512), lds, s, a);                                                        \
  } break
    switch (nkk) { RDM_GP(1); RDM_GP(2); RDM_GP(3); RDM_GP(4); RDM_GP(5); RDM_GP(6); RDM_GP(7); RDM_GP(8); RDM_GP(9); RDM_GP(10); RDM_GP(11); default: break; }
#undef RDM_GP
    RDM_CENSUS("gemm_panel_bf16_kernel/nkk%d", cdiv(a.K, 32));
  } else if (t96 >= 512 && wide112) RDM_G(2, 7, 4, 1, 64, 2);
  else if (t96 >= 512 && n96) RDM_G(4, 3, 2, 2, 64, 2);
  else if ((long)cdiv(a.M, 128) * cdiv(a.N, 48) >= 512) RDM_G(2, 3, 4, 1, 64, 2);
  else if (a.M > 1024) RDM_G(2, 3, 2, 2, 64, 2);                         // 64 x 96
  else RDM_G(1, 3, 2, 2, 64, 2);                                         // 32 x 96
  if (split > 1) {
    const long mn4 = (long)a.M * a.N / 4;
    hipLaunchKernelGGL(k_reduce_partials_bf16, dim3((unsigned)std::min<long>(cdiv(mn4, 256), 4096)), dim3(256), 0, s, a.partial, split, mn4, a.N / 4,
                       static_cast<unsigned short*>(a.out), a.ldc, a.oscale, a.oshift);
  }
#undef RDM_G
  profile_end(tk, s);
  RDM_LAUNCH_OK();
  return 0;
}

int launch_conv3x3_bf16(const Conv3Bf16Args& a_in, hipStream_t s) {
  Conv3Bf16Args a = a_in;
  RDM_CHECK_ARG(a.C > 0 && a.C % 8 == 0 && a.ldy % 8 == 0 && a.ldw % 8 == 0 && a.wtap % 8 == 0 && a.ldc % 4 == 0, "conv3x3_bf16: C (%d) and the strides must be multiples of 8", a.C);
  RDM_CHECK_ARG((((uintptr_t)a.Y | (uintptr_t)a.Wt | (uintptr_t)a.scale | (uintptr_t)a.shift) & 15) == 0 && ((uintptr_t)a.out & 7) == 0, "conv3x3_bf16: operands must be 16-byte aligned");
  RDM_CHECK_ARG((a.scale == nullptr) == (a.shift == nullptr) && a.M == a.B * a.H * a.W, "conv3x3_bf16: scale and shift come together; M must be B*H*W");
  const long yb = ((long)(a.M - 1) * a.ldy + a.C) * 2, wb = (8L * a.wtap + 47L * a.ldw + a.C) * 2;
  if (yb >= 0xFFFFFFFFL || wb >= 0xFFFFFFFFL) { set_error("conv3x3_bf16: operand extent >= 4 GiB"); return RDM_ERR_UNSUPPORTED; }
  a.y_bytes = (unsigned)yb; a.w_bytes = (unsigned)wb; a.p_bytes = (unsigned)(a.C * 4);
  // Tile and K-split.  Big pixel tiles stage the 27.6 KB weight slab once per 256 pixels (1.7x less L2 traffic per MFMA than
  // 128-pixel tiles); short grids are filled by splitting the channel slabs over grid.y (>= 2 slabs per split), partial sums go to
  // f32 slabs that one tiny launch reduces in a fixed order.  Target: 2-3 workgroups per CU.
  auto image_slots = [&](int bm_) {
    // zero-padded LDS image: the tile's real rows (<= bm / W + 2 partial rows), one row above and below, and two pad rows for every
    // image boundary the tile can straddle; each row W + 2 wide
    const int rows = bm_ / a.W + 2 + 2 + 2 * (bm_ / (a.H * a.W) + 1);
    return (rows * (a.W + 2) + 7) & ~7;
  };
  int bm = a.M >= 8192 ? 256 : a.M >= 4096 ? 128 : 64;
  if (bm == 256 && cdiv(image_slots(256) * 4, 256) > 12) bm = 128;        // the 256-pixel kernel would spill with > 12 image chunks per thread (very wide rows)
  const int tiles = cdiv(a.M, bm), slabs = cdiv(a.C, 32);
  int split = 1;
  if (a.partial && tiles < 1024) {
    // Cost model in units of one slab of one block: time ~ rounds x (slabs per block + 2 for prologue / epilogue) + the reduction
    // pass (split x M x 48 x 8 bytes at ~3 TB/s).  `slots` workgroups are resident at once: wave quantisation matters - a grid of
    // 1.06 x slots runs as long as one of 2 x slots.  (The first version of this kernel declared no minimum occupancy, the
    // compiler parked 72 values in AGPRs, and ONE workgroup ran per CU: a time-vs-grid sweep, tools/bf16_occupancy_probe.py, shows it.)
    const int hl_ = cdiv(image_slots(bm) * 4, 256);
    const int slots = 256 * ((bm == 256 || hl_ > 6) ? 2 : 3);             // workgroups per CU: register-limited (see -Rpass-analysis: 2 or 3 waves per SIMD)
    const double t_slab = bm == 256 ? 2.5e-6 : bm == 128 ? 1.5e-6 : 1.0e-6;
    const double red = (double)a.M * 384.0 / 3e12 / t_slab;
    const long cap = std::min<long>(std::max(slabs / 2, 1), (long)(a.partial_floats / ((size_t)a.M * 48)));
    double best = -1;
    for (long sp = 1; sp <= cap; ++sp) {
      const long per = cdiv(slabs, sp), spe = cdiv(slabs, per);           // effective split (no empty range)
      const long blocks = (long)tiles * spe, rounds = (blocks + slots - 1) / slots;
      const double cost = (double)rounds * (double)(per + 2) + (spe > 1 ? spe * red : 0.0);
      if (best < 0 || cost < best * 0.97) { best = cost; split = (int)spe; }
    }
  }
  a.split = split;
  a.slots = image_slots(bm);
  const int hl = cdiv(a.slots * 4, 256);
  RDM_CHECK_ARG(hl <= 20, "conv3x3_bf16: rows of %d pixels need an LDS image of %d pixels (> 1280)", a.W, a.slots);
  const size_t ldsb = ((size_t)a.slots * 32 + 9 * 48 * 32) * 2;
  RDM_CHECK_ARG(ldsb <= 160 * 1024, "conv3x3_bf16: LDS image of %zu bytes exceeds the 160 KB LDS", ldsb);
  void* tk = profile_begin(s, 2.0 * a.M * 48.0 * a.C * 9.0, 8, 2.0 * ((double)a.M * a.C + 9.0 * 48 * a.C + (double)a.M * 48));
  dim3 grid(tiles, split);
#define RDM_C3(MT_, HL_)                                                                                                                  \
  do {                                                                                                                                     \
    if (ldsb > 65536) RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<MT_, HL_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
    hipLaunchKernelGGL((conv3x3_bf16_kernel<MT_, HL_>), grid, dim3(256), ldsb, s, a);                                                       \
  } while (0)
#define RDM_C3M(MT_)                                                          \
  do {                                                                        \
    if (hl <= 4) RDM_C3(MT_, 4); else if (hl <= 6) RDM_C3(MT_, 6); else if (hl <= 9) RDM_C3(MT_, 9); else if (hl <= 12) RDM_C3(MT_, 12); \
    else RDM_C3(MT_, 20);                                                     \
  } while (0)
  if (bm == 256) RDM_C3M(4); else if (bm == 128) RDM_C3M(2); else RDM_C3M(1);
#undef RDM_C3M
#undef RDM_C3
  if (split > 1) {
    const long mn4 = (long)a.M * 12;
    hipLaunchKernelGGL(k_reduce_partials_bf16, dim3((unsigned)std::min<long>(cdiv(mn4, 256), 4096)), dim3(256), 0, s, a.partial, split, mn4, 12, a.out, a.ldc, nullptr, nullptr);
  }
  profile_end(tk, s);
  RDM_LAUNCH_OK();
  return 0;
}


// ---- conv3x3_act_bf16_kernel: tile / split choice ----
namespace {
struct Conv3ActPlan { int nc = 0, split = 1, tpi = 0, rect = 0; double cost = 0; };
// max zero-padded LDS slots over the tiles of one image
int act_slots(int H, int W, int bm) {
  const int HW = H * W, tpi = cdiv(HW, bm);
  int rows = 0;
  for (int t = 0; t < tpi; ++t) rows = std::max(rows, (std::min(HW, (t + 1) * bm) - 1) / W - (t * bm) / W + 3);
  return rows * (W + 2);
}
bool plan_conv3_act(int C, int B, int H, int W, size_t partial_floats, int n_counters, Conv3ActPlan& best) {
  const int HW = H * W, slabs = C / 32;
  best = Conv3ActPlan{};
  for (int pass = 0; pass < 2 && best.nc == 0; ++pass)                 // pass 1 (nothing fits as whole rows): 64-column rectangles
  for (int nc = 1; nc <= 4; nc += pass ? nc : 1) {                     // (rectangles of 2, 4, 8 rows: nc 1, 2, 4)
    const int bm = nc * 128;
    const int tpi = pass ? cdiv(H, bm / 64) * cdiv(W, 64) : cdiv(HW, bm);
    if (!pass && act_slots(H, W, bm) > ACT_MAX_SLOTS) continue;
    const long tiles = (long)B * tpi;
    for (int sp = 1; sp <= std::min(slabs, 32); ++sp) {
      const int per = cdiv(slabs, sp), spe = cdiv(slabs, per);
      if (spe != sp) continue;
      if (spe > 1 && ((size_t)tiles * spe * bm * 48 > partial_floats || tiles > n_counters)) break;
      const long blocks = tiles * spe, rounds = cdiv(blocks, 256L);          // one workgroup per CU (159 KB of LDS)
      // microseconds.  Per slab: 216 MFMAs per consumer (1.7) or, when few workgroups run, the DMA round trip (1.2); prologue +
      // epilogue 3; ordered combine: 2 memory round trips per 6 slabs
      const double cost = (double)rounds * (per * 1.8 + 3.0) + (spe > 1 ? 1.0 + 2 * cdiv(spe, 6) * 1.7 : 0.0);
      if (best.nc == 0 || cost < best.cost * 0.98) { best.nc = nc; best.split = spe; best.tpi = tpi; best.rect = pass; best.cost = cost; }
    }
  }
  return best.nc != 0;
}
}  // namespace

size_t conv3x3_act_partial_floats(int C, int B, int H, int W) {
  // enough for a split of 8 on 512-pixel tiles (the heuristic takes what it is given)
  const long tiles = (long)B * cdiv(H * W, 128);
  return (size_t)tiles * 128 * 48 * (size_t)std::min(std::max(C / 64, 1), 8) + (size_t)B * 8 * 512 * 48;
}
int conv3x3_act_tiles(int B, int H, int W) { return B * cdiv(H * W, 128); }
bool conv3x3_act_fits(int B, int H, int W) {
  Conv3ActPlan pl;
  return plan_conv3_act(32, B, H, W, 0, 0, pl);
}

int launch_pack_w3_frag_bf16(const float* w, void* img, int C, int Cpad, int packed, hipStream_t s) {
  RDM_CHECK_ARG(C > 0 && Cpad >= C && Cpad % 32 == 0, "pack_w3_frag_bf16: padded channel count (%d) must be a multiple of 32 >= %d", Cpad, C);
  const long total = (long)(Cpad / 32) * 27 * 512;
  hipLaunchKernelGGL(k_pack_w3_frag_bf16, dim3((unsigned)std::min<long>(cdiv(total, 256), 8192)), dim3(256), 0, s, w, static_cast<unsigned short*>(img), C, Cpad, packed);
  RDM_LAUNCH_OK();
  return 0;
}

int launch_conv3x3_act_bf16(const Conv3ActArgs& a_in, hipStream_t s) {
  Conv3ActArgs a = a_in;
  RDM_CHECK_ARG(a.C > 0 && a.C % 32 == 0 && a.ldy % 8 == 0 && a.ldc % 4 == 0, "conv3x3_act_bf16: C (%d) must be a multiple of 32, the strides of 8 / 4", a.C);
  RDM_CHECK_ARG((((uintptr_t)a.Y | (uintptr_t)a.Wimg) & 15) == 0 && ((uintptr_t)a.out & 7) == 0, "conv3x3_act_bf16: operands must be 16-byte aligned");
  const long M = (long)a.B * a.H * a.W;
  const long yb = ((M - 1) * a.ldy + a.C) * 2, wb = (long)(a.C / 32) * 27 * 1024;
  if (yb >= 0xFFFFFFFFL || wb >= 0xFFFFFFFFL) { set_error("conv3x3_act_bf16: operand extent >= 4 GiB"); return RDM_ERR_UNSUPPORTED; }
  a.y_bytes = (unsigned)yb; a.w_bytes = (unsigned)wb;
  Conv3ActPlan pl;
  const bool can_split = a.partial && a.counters;
  if (!plan_conv3_act(a.C, a.B, a.H, a.W, can_split ? a.partial_floats : 0, can_split ? a.n_counters : 0, pl)) {
    set_error("conv3x3_act_bf16: rows of %d pixels do not fit the LDS image", a.W);
    return RDM_ERR_UNSUPPORTED;
  }
  a.split = pl.split; a.slots = ACT_MAX_SLOTS; a.tiles_per_img = pl.tpi; a.rect = pl.rect;
#ifdef RDM_DEV_VARIANTS
  a.abl = g_variant >= 100 && g_variant < 116 ? g_variant - 100 : 0;
#endif
  void* tk = profile_begin(s, 2.0 * M * 48.0 * a.C * 9.0, 12, 2.0 * ((double)M * a.C + 9.0 * 48 * a.C + (double)M * 48));
  dim3 grid((unsigned)(a.B * pl.tpi), (unsigned)pl.split);
#define RDM_C3A(NC_)                                                                                                                      \
  case NC_: {                                                                                                                              \
    RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_act_bf16_kernel<NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, ACT_LDS_BYTES)); /* per device: set on every launch */ \
    hipLaunchKernelGGL((conv3x3_act_bf16_kernel<NC_>), grid, dim3(NC_ * 128), ACT_LDS_BYTES, s, a);                                        \
  } break
  switch (pl.nc) {
    RDM_C3A(1); RDM_C3A(2); RDM_C3A(3); RDM_C3A(4);
    default: set_error("conv3x3_act_bf16: bad plan"); return RDM_ERR_UNSUPPORTED;
  }
#undef RDM_C3A
  RDM_CENSUS("conv3x3_act_bf16_kernel/nc%d/%s%s", pl.nc, pl.split > 1 ? "splitK" : "direct", pl.rect ? "/rect" : "");
  profile_end(tk, s);
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
