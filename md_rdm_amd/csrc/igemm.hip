// fp32 MFMA implicit-GEMM convolution kernels for gfx950 (CDNA4), NHWC activations.
//
// Replaces the ATen/cuDNN conv2d calls the reference reaches from
//   network/RDM_Net.py:524 (conv_e1, via im2col), :526-531 (_DenseBlock/_Transition 1x1 and 3x3),
//   :144 (decoder dense block), :146-147 (conv1/conv2), :163-236 (WSM convs)
// and their autograd-generated dgrad / wgrad.
//
// Design (MI355X-first, not a cuDNN translation):
//  * v_mfma_f32_16x16x4_f32: exact-f32 matrix core path (1e-4 parity needs f32; 157 TF peak).
//    Lane l holds A[row l&15][k l>>4] and B[k l>>4][col l&15]; D: col = l&15, row = 4*(l>>4)+reg.
//  * A 256-thread workgroup = 4 wave64s arranged WM x WN, each wave owning an (MT*16)x(NT*16)
//    accumulator tile.  Channel counts of this network are all multiples of 48, so the wave tile
//    is 64x48 (MT=4, NT=3): no N-quantisation waste on 48/96/.../2736-wide layers.
//  * K is walked in 16-deep slabs (every contracted extent here is a multiple of 16), channel
//    slab outer / filter tap inner so the 9 taps of a 3x3 re-hit the same lines in L1/L2.
//  * Operands are staged global -> registers -> LDS ([k][row] image, row contiguous, +4 pad)
//    with the next slab's global loads in flight under the current slab's MFMAs; two LDS
//    buffers, one barrier per slab.  The BN scale/shift + ReLU of the *consumer* layer is
//    applied in the staging registers (concat-free DenseNet: each layer reads a channel prefix
//    of its block buffer in place), so normalised activations never exist in HBM.
//  * Epilogues: store (+bias), store + per-channel sum/sum^2 (f64 atomics) for the next
//    BatchNorm, ReLU-mask + BN-backward reductions for dgrad, f32 atomics for split-K.
#include <algorithm>
#include <utility>
#include <vector>

#include "rdm_common.h"
#include "elementwise.h"
#include "xsplit.h"

namespace rdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 16;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float4 bnrelu4(float4 v, float4 sc, float4 sh) {
  v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
  v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
  v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
  v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
  return v;
}

template <int MT, int NT>
__device__ __forceinline__ void mma_slab(const float* __restrict__ As, const float* __restrict__ Bs, int lda, int ldb,
                                         int wrow, int wcol, int l16, int g, f32x4 (&acc)[MT][NT]) {
  // all fragments of the slab are fetched first (4*(MT+NT) VGPRs), so the MT*NT*4 MFMAs issue
  // back-to-back behind ONE LDS latency instead of one per k-step
  float a[BK / 4][MT], b[BK / 4][NT];
#pragma unroll
  for (int ks = 0; ks < BK / 4; ++ks) {
    const int k = ks * 4 + g;
#pragma unroll
    for (int i = 0; i < MT; ++i) a[ks][i] = As[k * lda + wrow + i * 16 + l16];
#pragma unroll
    for (int j = 0; j < NT; ++j) b[ks][j] = Bs[k * ldb + wcol + j * 16 + l16];
  }
#pragma unroll
  for (int ks = 0; ks < BK / 4; ++ks)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Global operands are read through buffer descriptors (SRD): 32-bit byte offsets instead of 64-bit
// pointer arithmetic, and the hardware range check returns 0 for an out-of-range offset - so zero
// padding, ragged tile edges and the K tail cost one v_cndmask (offset := ~0u) instead of a
// divergent branch per load.  Every operand extent is < 4 GiB (checked by the launchers).
// ---------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFFFu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ---------------------------------------------------------------------------------------------
// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs in dispatch order
// (x fastest), and each XCD has its own 4 MB L2: with the plain order the gx column tiles that
// share one A row-tile land on 8 different L2s and the tile is pulled over the fabric up to 8 times
// (the 1x1 dgrad of dense_e2 read 3.5x its operand bytes).  Remapped, XCD x works through the
// contiguous range [x*total/8, (x+1)*total/8) of the logical (x fastest) order, so blocks that are
// adjacent in time on one XCD are adjacent column tiles of the same row-tile.  A bijection for any
// grid size; placement only affects speed, never results.  A/B on dense_e2: 1x1 dgrad 116.6 vs 109.7 TFLOP/s.
// ---------------------------------------------------------------------------------------------
// flat: 0 = XCD-aware, 1 = hardware order, G >= 2 = XCD-aware AND row-grouped: inside an XCD's range the tiles are walked in groups
// of G row-tiles, column by column (G blocks in a row share one WEIGHT tile, the group's G activation tiles stay in L2 across all
// columns).  For the 3x3 dgrad of dense_e2 (57 column tiles x 83 KB of weights = 4.7 MB, more than one L2 holds) the plain order
// streams every weight tile once per ROW tile: 1.22 GB of L2 fills per launch against 18 MB of operands (tools/fetch_calibration.py).
__device__ __forceinline__ void xcd_block_order(int flat, int& bx, int& by, int& bz) {
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if (flat == 1) return;
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
  const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned x = L & 7u, seq = L >> 3, q = total >> 3, r = total & 7u;
  const unsigned Lp = x * q + (x < r ? x : r) + seq;
  if (flat >= 2 && gridDim.z == 1) {
    const unsigned G = (unsigned)flat, gsz = G * gx, grp = Lp / gsz, rem = Lp - grp * gsz;
    const unsigned rows = min(G, gy - grp * G);                       // the last group may be short
    const unsigned c = rem / rows;
    bx = (int)c; by = (int)(grp * G + (rem - c * rows)); bz = 0;
    return;
  }
  bx = (int)(Lp % gx);
  const unsigned t = Lp / gx;
  by = (int)(t % gy); bz = (int)(t / gy);
}

template <int MT, int NT, int EPI>
__device__ __forceinline__ void conv_epilogue(const FwdArgs& p, f32x4 (&acc)[MT][NT], int m0, int n0, int wrow, int wcol, int l16, int g) {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + l16;
    const bool nok = n < p.N;
    float bias = 0.f, xs = 0.f, xt = 0.f;
    if (EPI == EPI_STORE && p.bias != nullptr && nok) bias = p.bias[n];
    if ((EPI == EPI_MASK_STATS || EPI == EPI_MASK_STATS_ATOMIC) && nok) { xs = p.x_scale[n]; xt = p.x_shift[n]; }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow + i * 16 + g * 4 + r;
        if (nok && m < p.M) {
          float v = acc[i][j][r];
          float* dst = p.out + (long)m * p.ldc + n;
          if (EPI == EPI_STORE) {
            *dst = v + bias;
          } else if (EPI == EPI_STORE_STATS) {
            if (p.add_out) v += *dst;            // this launch holds the LAST K-partial of the element: finish the sum, then reduce
            *dst = v;
            s0 += v; s1 += v * v;
          } else if (EPI == EPI_MASK_STATS) {
            const float x = p.X[(long)m * p.ldx + n];
            v = (fmaf(x, xs, xt) > 0.f) ? v : 0.f;
            *dst = v;
            s0 += v; s1 += v * x;
          } else if (EPI == EPI_MASK_STATS_ATOMIC) {
            // split-K: the ReLU gate and both BatchNorm-backward reductions are LINEAR in the partial
            // sum, so every K-split gates and reduces its own partial (sum of gated partials = gated sum)
            const float x = p.X[(long)m * p.ldx + n];
            v = (fmaf(x, xs, xt) > 0.f) ? v : 0.f;
            if (v != 0.f) atomicAdd(dst, v);
            s0 += v; s1 += v * x;
          } else {
            atomicAdd(dst, v);
          }
        }
      }
    }
    if (EPI == EPI_STORE_STATS || EPI == EPI_MASK_STATS || EPI == EPI_MASK_STATS_ATOMIC) {
      s0 += __shfl_xor(s0, 16); s1 += __shfl_xor(s1, 16);
      s0 += __shfl_xor(s0, 32); s1 += __shfl_xor(s1, 32);
      if (g == 0 && nok) {
        atomicAdd(p.stat0 + n, (double)s0);
        atomicAdd(p.stat1 + n, (double)s1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel: A is an NHWC tensor gathered per filter tap (K-contiguous),
// B is the packed weight [tap][n][c] read either along c (forward) or along n (dgrad).
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, bool TAPS, bool B_KSTRIDED, int EPI, bool RAWBN = false>
__global__ __launch_bounds__(256, 4) void conv_fwd_kernel(FwdArgs p) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN;
  __shared__ __attribute__((aligned(16))) float Sbn[RAWBN ? 2 * RAWBN_MAX_C : 4];      // RAWBN: (scale | shift) of the C contracted channels, formed here
  constexpr int LDA = BM + 4, LDB = BN + 4;                        // [k][row] images
  constexpr int AL = (BM * 4 + 255) / 256;                         // float4 loads of A per thread per slab
  constexpr int BL = B_KSTRIDED ? (BK * (BN / 4) + 255) / 256 : (BN * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave / WN) * MT * 16, wcol = (wave % WN) * NT * 16;
  int bx, by, bz;
  xcd_block_order(p.xcd_flat, bx, by, bz);
  const int n0 = bx * BN, m0 = by * BM;
  const ConvGeom& G = p.g;
  const int ntaps = TAPS ? G.KH * G.KW : 1;
  const int nslab_total = (p.C / BK) * ntaps;
  int s_begin = 0, s_end = nslab_total;
  if (EPI == EPI_ATOMIC || EPI == EPI_MASK_STATS_ATOMIC) {
    const int per = (nslab_total + (int)gridDim.z - 1) / (int)gridDim.z;
    s_begin = bz * per;
    s_end = min(nslab_total, s_begin + per);
    if (s_begin >= s_end) return;
  }
  const __amdgpu_buffer_rsrc_t srdA = make_srd(p.A, p.a_bytes), srdW = make_srd(p.Wt, p.w_bytes);

  // ---- per-thread operand bookkeeping, fixed for the whole K loop ----
  const int kq_a = tid & 3;                    // every A load of this thread covers the same k-quad
  unsigned a_voff[AL];                         // byte offset of (row's origin pixel, k-quad) - modulo 2^32
  int a_iy[AL], a_ix[AL];                      // TAPS: input coordinate of tap (0,0)
  bool a_ok[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int idx = tid + i * 256;
    const int m = m0 + (idx >> 2);
    a_ok[i] = (idx < BM * 4) && (m < p.M);
    a_iy[i] = 0; a_ix[i] = 0;
    int pix = m;
    if (TAPS && a_ok[i]) {
      const int hw = G.Ho * G.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / G.Wo, ox = rem - oy * G.Wo;
      a_iy[i] = G.dir > 0 ? oy * G.SH - G.PH : oy + G.PH;
      a_ix[i] = G.dir > 0 ? ox * G.SW - G.PW : ox + G.PW;
      pix = (b * G.H + a_iy[i]) * G.W + a_ix[i];          // may be "negative": offsets are modular
    }
    a_voff[i] = (unsigned)pix * (unsigned)(p.lda * 4) + (unsigned)(kq_a * 16);
  }
  unsigned b_voff[BL];
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int idx = tid + i * 256;
    if (!B_KSTRIDED) {
      const int row = idx >> 2, kq = idx & 3;
      b_voff[i] = (idx < BN * 4 && n0 + row < p.N) ? (unsigned)(n0 + row) * (unsigned)(p.ldw * 4) + (unsigned)(kq * 16) : OOB;
    } else {
      const int k = idx / (BN / 4), r4 = idx - k * (BN / 4);
      b_voff[i] = (idx < BK * (BN / 4) && n0 + r4 * 4 < p.N) ? (unsigned)k * (unsigned)(p.ldw * 4) + (unsigned)((n0 + r4 * 4) * 4) : OOB;
    }
  }
  const bool bnrelu = RAWBN || p.a_scale != nullptr;
  if (RAWBN) {
    for (int c = tid; c < p.C; c += 256) {
      float sc, sh, mean, rstd; double var;
      bn_affine_from_sums(p.a_sum[c], p.a_sq[c], p.a_count, p.a_gamma[c], p.a_beta[c], 1e-5f, sc, sh, mean, rstd, var);
      Sbn[c] = sc; Sbn[RAWBN_MAX_C + c] = sh;
    }
    __syncthreads();
  }

  // slab cursor: channel slab `cs` outer, filter tap (r,q) inner - advanced incrementally (scalar)
  int cur_cs = s_begin, cur_tap = 0, cur_r = 0, cur_q = 0;
  if (TAPS) { cur_cs = s_begin / ntaps; cur_tap = s_begin - cur_cs * ntaps; cur_r = cur_tap / G.KW; cur_q = cur_tap - cur_r * G.KW; }
  auto advance = [&]() {
    if (TAPS) {
      ++cur_tap; ++cur_q;
      if (cur_q == G.KW) { cur_q = 0; ++cur_r; }
      if (cur_tap == ntaps) { cur_tap = 0; cur_r = 0; cur_q = 0; ++cur_cs; }
    } else {
      ++cur_cs;
    }
  };

  // Raw loads only: nothing below consumes a loaded value, so the loads stay in flight across the
  // MFMA phase of the current slab; BN-ReLU / zero padding are applied when staging into LDS.
  float4 ra[AL], rb[BL], rsc = make_float4(1.f, 1.f, 1.f, 1.f), rsh = make_float4(0.f, 0.f, 0.f, 0.f);
  bool rok[AL];
  auto load_slab = [&]() {
    const int c0 = cur_cs * BK;
    // wave-uniform part of the offsets
    const unsigned a_uni = (unsigned)(c0 * 4) + (TAPS ? (unsigned)(G.dir * (cur_r * G.W + cur_q)) * (unsigned)(p.lda * 4) : 0u);
    const unsigned b_uni = B_KSTRIDED ? (unsigned)cur_tap * (unsigned)(p.wtap * 4) + (unsigned)c0 * (unsigned)(p.ldw * 4)
                                      : (unsigned)cur_tap * (unsigned)(p.wtap * 4) + (unsigned)(c0 * 4);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      bool ok = a_ok[i];
      if (TAPS) {
        const int iy = a_iy[i] + G.dir * cur_r, ix = a_ix[i] + G.dir * cur_q;
        ok = ok & ((unsigned)iy < (unsigned)G.H) & ((unsigned)ix < (unsigned)G.W);
      }
      rok[i] = ok;
      ra[i] = bld4(srdA, ok ? a_voff[i] + a_uni : OOB);
    }
    if (RAWBN) { rsc = ld4(Sbn + c0 + kq_a * 4); rsh = ld4(Sbn + RAWBN_MAX_C + c0 + kq_a * 4); }
    else if (bnrelu) { rsc = ld4(p.a_scale + c0 + kq_a * 4); rsh = ld4(p.a_shift + c0 + kq_a * 4); }
#pragma unroll
    for (int i = 0; i < BL; ++i) rb[i] = bld4(srdW, b_voff[i] == OOB ? OOB : b_voff[i] + b_uni);
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = tid + i * 256;
      if (idx < BM * 4) {
        const int row = idx >> 2;
        float4 v = ra[i];
        if (bnrelu) { v = bnrelu4(v, rsc, rsh); if (!rok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f); }
        float* d = &As[buf][(kq_a * 4) * LDA + row];
        d[0] = v.x; d[LDA] = v.y; d[2 * LDA] = v.z; d[3 * LDA] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int idx = tid + i * 256;
      if (!B_KSTRIDED) {
        if (idx < BN * 4) {
          const int row = idx >> 2, kq = idx & 3;
          float* d = &Bs[buf][(kq * 4) * LDB + row];
          d[0] = rb[i].x; d[LDB] = rb[i].y; d[2 * LDB] = rb[i].z; d[3 * LDB] = rb[i].w;
        }
      } else {
        if (idx < BK * (BN / 4)) {
          const int k = idx / (BN / 4), r4 = idx - k * (BN / 4);
          *reinterpret_cast<float4*>(&Bs[buf][k * LDB + r4 * 4]) = rb[i];
        }
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_slab();
  store_slab(0);
  __syncthreads();
  int buf = 0;
  // Ablation on MI355X (e2 3x3 forward): MFMA + fragment reads alone run at 90 % of the f32 matrix-core peak;
  // issuing the next slab's loads costs 17 %, staging them into LDS 9-12 %, the barrier 3 %.
  for (int s = s_begin; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    if (more) { advance(); load_slab(); }
    mma_slab<MT, NT>(As[buf], Bs[buf], LDA, LDB, wrow, wcol, l16, g, acc);
    if (more) store_slab(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  conv_epilogue<MT, NT, EPI>(p, acc, m0, n0, wrow, wcol, l16, g);
}

// ---------------------------------------------------------------------------------------------
// 1x1 forward with a 256 x 128 macro tile and LDS-DMA staging (big grids: dense_e2 / dense_e3 bottlenecks).
// Round-2 measurements (csrc/microbench.hip): on these shapes the vendor GEMM reaches 130 TFLOP/s where the
// 128 x 48 / 128 x 96 register-staged tiles reach 98 - 114 - the small tiles pay the tile boundary (prologue latency, epilogue, drain)
// thousands of times.  Here a workgroup owns 256 pixels x 128 channels (wave tile 128 x 64: 128 accumulator registers), both operands
// reach LDS by `buffer_load_dwordx4 ... lds` (SRD + one 32-bit lane offset per 1-KiB piece, the slab advance in a scalar offset: no
// staging registers, no ds_write, no vector address arithmetic), 3 buffers of 24 KB keep one slab in flight across each barrier,
// 2 workgroups per CU.  LDS image: lane-linear [row][16 floats]; the 16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3)
// (applied to the SOURCE address), so a fragment ds_read_b128 (16 rows x one chunk) touches 16 distinct slots of the 256-byte bank
// row.  Lane (row, kq) takes channels 4kq .. 4kq+3 of the slab and feeds element e to MFMA step e - a permutation of k shared by
// both operands.  The consumer BN-ReLU is applied to the A fragments (8 VALU per 16-row tile and slab next to 16 MFMAs).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass cannot type-check the gfx950 builtin and would silently drop the kernel's stub)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, (int)soff, 0, 0);
#endif
}

template <int EPI, bool BNRELU, int MT = 8>      // MT = 8: 256 x 128 macro tile; MT = 6: 192 x 128 (chosen where it fills the workgroup rounds better)
__global__ __launch_bounds__(256, 2) void conv1x1_dma256_kernel(FwdArgs p) {
  constexpr int NT = 4, BM = MT * 32, BN = 128, NBUF = 3, NA = BM / 16 / 4, NB = BN / 16 / 4;      // 1-KiB pieces (16 rows x 64 B) per wave and slab: 4 (3) + 2
  constexpr int SLAB = (BM + BN) * BK + (BNRELU ? 512 : 0);                         // + one piece each for the slab's 16 scale / 16 shift values
  __shared__ __attribute__((aligned(1024))) float smem[NBUF * SLAB];               // ONE object (a second one makes hipcc drain vmcnt before every ds_read)
  float* const sm = smem;
  auto As = [&](int buf) { return sm + buf * SLAB; };
  auto Bs = [&](int buf) { return sm + buf * SLAB + BM * BK; };
  auto Cf = [&](int buf) { return sm + buf * SLAB + (BM + BN) * BK; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave >> 1) * MT * 16, wcol = (wave & 1) * NT * 16;
  int bx, by, bz;
  xcd_block_order(p.xcd_flat, bx, by, bz);
  const int n0 = bx * BN, m0 = by * BM;
  const __amdgpu_buffer_rsrc_t ra = make_srd(p.A, p.a_bytes), rw = make_srd(p.Wt, p.w_bytes);
  // the consumer BN-ReLU's coefficients travel with the slab: wave 0 copies the 16 scale values, wave 1 the 16 shift values (lanes 0..3,
  // the other lanes are out of range and write zeros) - every load of the loop is an LDS-DMA, so the counted waits stay exact
  const __amdgpu_buffer_rsrc_t rc = make_srd(wave == 0 ? p.a_scale : p.a_shift, (unsigned)(p.C * 4));
  const unsigned coff = lane < 4 ? (unsigned)(lane * 16) : OOB;
  const int prow = lane >> 2, pch = (lane & 3) ^ ((prow >> 2) & 3);
  unsigned aoff[NA], boff[NB];
#pragma unroll
  for (int t = 0; t < NA; ++t) aoff[t] = (unsigned)(min(m0 + (wave + 4 * t) * 16 + prow, p.M - 1) * p.lda + pch * 4) * 4u;
#pragma unroll
  for (int t = 0; t < NB; ++t) boff[t] = (unsigned)(min(n0 + (wave + 4 * t) * 16 + prow, p.N - 1) * p.ldw + pch * 4) * 4u;
  auto issue = [&](int buf, int s) {
    const unsigned so = (unsigned)s * (BK * 4u);
#pragma unroll
    for (int t = 0; t < NA; ++t) lds_dma16(ra, aoff[t], so, As(buf) + (wave + 4 * t) * 256);
#pragma unroll
    for (int t = 0; t < NB; ++t) lds_dma16(rw, boff[t], so, Bs(buf) + (wave + 4 * t) * 256);
    if (BNRELU && wave < 2) lds_dma16(rc, coff, so, Cf(buf) + wave * 256);
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = p.C / BK;
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
    if (b < nk) issue(b, b);
  const int sl = ((g ^ (l16 >> 2)) & 3) * 4;
  for (int s = 0; s < nk; ++s) {
    // slab s has landed: all but the youngest (NBUF - 2) groups of this wave's DMAs (the tail has nothing younger in flight)
    if (s + NBUF - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (BNRELU && wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB + 1) * (NBUF - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB) * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NBUF - 1 < nk) issue((s + NBUF - 1) % NBUF, s + NBUF - 1);      // into the buffer slab s-1 was read from: every wave is past it
    const int buf = s % NBUF;
    f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (BNRELU) { sc = *reinterpret_cast<const f32x4*>(Cf(buf) + g * 4); sh = *reinterpret_cast<const f32x4*>(Cf(buf) + 256 + g * 4); }
    f32x4 b4[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Bs(buf) + (wcol + j * 16 + l16) * BK + sl);
    constexpr int MH = MT / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                        // the A fragments in two halves: 16 + 16 fragment registers beside 128 accumulators
      f32x4 a4[MH];
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        a4[i] = *reinterpret_cast<const f32x4*>(As(buf) + (wrow + (h * MH + i) * 16 + l16) * BK + sl);
        if (BNRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[i][e] = fmaxf(fmaf(a4[i][e], sc[e], sh[e]), 0.f);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[h * MH + i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[h * MH + i][j], 0, 0, 0);
    }
  }
  conv_epilogue<MT, NT, EPI>(p, acc, m0, n0, wrow, wcol, l16, g);
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 forward with an LDS HALO tile (the live DenseNet 3x3: 48 outputs, K = 9*Cb).
// A workgroup owns 256 consecutive output pixels (linear NHWC index) x 48 channels.  Per 16-channel
// slab it stages ONE halo run of 256 + 2*(W+1) pixels - BN scale/shift + ReLU applied once per
// element - and the 9 filter taps read their A fragments from that run at offset r*W + q; border /
// batch wrap-around is undone per (row, tap) with a 9-bit validity mask held in registers.  Versus the
// generic kernel that re-gathers, re-normalises and re-stages the tile for every tap this is 5.6x
// fewer global loads, BN-ReLU VALU ops and LDS stores per MFMA.  Weights are staged per tap
// (double-buffered, one barrier per tap).  Needs 2*(W+1) <= 256 (else the launcher uses the generic path).
// ---------------------------------------------------------------------------------------------
template <int EPI, int HL, bool DGRAD, int MT, bool RAWBN = false>   // HL = float4 halo loads per thread per channel slab = ceil(halo / 64); MT = 16-row tiles per wave
__global__ __launch_bounds__(256, 4) void conv3x3_halo_kernel(FwdArgs p) {
  __shared__ __attribute__((aligned(16))) float Sbn[RAWBN ? 2 * RAWBN_MAX_C : 4];
  // DGRAD: the same machinery run on the output gradient: dx[m][c] = sum_{tap,n} dy[m + (1-r)*W + (1-q)][n] * w[tap][n][c]
  // (taps mirrored, weights read along their input-channel rows), epilogue = ReLU gate + BN-backward sums.
  constexpr int NT = 3, BM = MT * 64, BN = 48;     // MT = 4: 256 pixels per block; MT = 2: 128 (layers with few pixels)
  // LDS image [k][pixel]: row stride = 16 mod 32 floats, so the two k-groups of a 32-lane half read disjoint
  // bank halves; k-quad kq is skewed by 8*kq floats, so the transposing ds_write_b32 of a half-wave
  // (8 pixels x 4 k-quads) lands on 32 distinct banks too.  (Was stride = 4 mod 32: 2-way conflicts both ways.)
  constexpr int LDH = HL * 64 + 16;             // halo run of 256 + 2*(W+1) <= HL*64 pixels (+pad)
  constexpr int LDB = BN;
  __shared__ __attribute__((aligned(16))) float Ah[BK * LDH + 24];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = wave * MT * 16;
  int bx, by, bz;
  xcd_block_order(p.xcd_flat, bx, by, bz);
  const int n0 = bx * BN, m0 = by * BM;
  const ConvGeom& G = p.g;
  const int W = G.W, H = G.H;
  const int halo = BM + 2 * (W + 1);
  const int ncs = p.C / BK;
  int cs_begin = 0, cs_end = ncs;
  if (EPI == EPI_ATOMIC || EPI == EPI_MASK_STATS_ATOMIC) {
    const int per = (ncs + (int)gridDim.z - 1) / (int)gridDim.z;
    cs_begin = bz * per;
    cs_end = min(ncs, cs_begin + per);
    if (cs_begin >= cs_end) return;
  }
  const __amdgpu_buffer_rsrc_t srdA = make_srd(p.A, p.a_bytes), srdW = make_srd(p.Wt, p.w_bytes);

  // ---- validity of every (row, tap) pair for this lane's row of each of its MT tiles ----
  unsigned vmask[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wrow + i * 16 + l16;
    unsigned v = 0;
    if (m < p.M) {
      const int hw = H * W;
      const int rem = m - (m / hw) * hw;
      const int oy = rem / W, ox = rem - oy * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = DGRAD ? 1 - t / 3 : t / 3 - 1, dx = DGRAD ? 1 - t % 3 : t % 3 - 1;
        if ((unsigned)(oy + dy) < (unsigned)H && (unsigned)(ox + dx) < (unsigned)W) v |= 1u << t;
      }
    }
    vmask[i] = v;
  }
  // ---- halo staging: thread -> (halo pixel h, k-quad) ----
  const int kq_a = tid & 3;
  unsigned h_voff[HL];
  bool h_ok[HL];
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const int hidx = (tid + i * 256) >> 2;                       // halo pixel index 0..halo-1
    const long pix = (long)m0 - (W + 1) + hidx;
    h_ok[i] = hidx < halo && pix >= 0 && pix < (long)G.B * H * W;
    h_voff[i] = (unsigned)pix * (unsigned)(p.lda * 4) + (unsigned)(kq_a * 16);
  }
  const bool bnrelu = RAWBN || p.a_scale != nullptr;
  if (RAWBN) {
    for (int c = tid; c < p.C; c += 256) {
      float sc, sh, mean, rstd; double var;
      bn_affine_from_sums(p.a_sum[c], p.a_sq[c], p.a_count, p.a_gamma[c], p.a_beta[c], 1e-5f, sc, sh, mean, rstd, var);
      Sbn[c] = sc; Sbn[RAWBN_MAX_C + c] = sh;
    }
    __syncthreads();
  }
  float4 rh[HL], rsc = make_float4(1.f, 1.f, 1.f, 1.f), rsh = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_halo = [&](int cs) {
    const unsigned uni = (unsigned)(cs * BK * 4);
#pragma unroll
    for (int i = 0; i < HL; ++i) rh[i] = bld4(srdA, h_ok[i] ? h_voff[i] + uni : OOB);
    if (RAWBN) { rsc = ld4(Sbn + cs * BK + kq_a * 4); rsh = ld4(Sbn + RAWBN_MAX_C + cs * BK + kq_a * 4); }
    else if (bnrelu) { rsc = ld4(p.a_scale + cs * BK + kq_a * 4); rsh = ld4(p.a_shift + cs * BK + kq_a * 4); }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < HL; ++i) {
      const int hidx = (tid + i * 256) >> 2;
      if (hidx < halo) {
        float4 v = rh[i];
        if (bnrelu) { v = bnrelu4(v, rsc, rsh); if (!h_ok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f); }
        float* d = &Ah[(kq_a * 4) * LDH + kq_a * 8 + hidx];
        d[0] = v.x; d[LDH] = v.y; d[2 * LDH] = v.z; d[3 * LDH] = v.w;
      }
    }
  };
  // ---- weights per tap: forward 48 rows (n) x 16 k read along k; dgrad 16 k (n) x 48 cols (c) read along c ----
  const int b_row = DGRAD ? tid / 12 : tid >> 2, b_q = DGRAD ? tid % 12 : tid & 3;     // dgrad: (k, float4 column group)
  const bool b_act = tid < BN * 4 && (DGRAD ? n0 + b_q * 4 < p.N : n0 + b_row < p.N);
  const unsigned b_voff = DGRAD ? (unsigned)b_row * (unsigned)(p.ldw * 4) + (unsigned)((n0 + b_q * 4) * 4)
                                : (unsigned)(n0 + b_row) * (unsigned)(p.ldw * 4) + (unsigned)(b_q * 16);
  float4 rb;
  auto load_b = [&](int cs, int tap) {
    const unsigned uni = (unsigned)tap * (unsigned)(p.wtap * 4) + (DGRAD ? (unsigned)(cs * BK) * (unsigned)(p.ldw * 4) : (unsigned)(cs * BK * 4));
    rb = bld4(srdW, b_act ? b_voff + uni : OOB);
  };
  auto store_b = [&](int buf) {
    if (tid < BN * 4) {
      if (DGRAD) {
        *reinterpret_cast<float4*>(&Bs[buf][b_row * LDB + b_q * 4]) = rb;
      } else {
        float* d = &Bs[buf][(b_q * 4) * LDB + b_row];
        d[0] = rb.x; d[LDB] = rb.y; d[2 * LDB] = rb.z; d[3 * LDB] = rb.w;
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_halo(cs_begin);
  load_b(cs_begin, 0);
  store_halo();
  store_b(0);
  __syncthreads();
  int buf = 0;
  for (int cs = cs_begin; cs < cs_end; ++cs) {
    const bool more_cs = cs + 1 < cs_end;
    if (more_cs) load_halo(cs + 1);                       // in flight during the 9 taps of this slab
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const bool more = tap < 8 || more_cs;
      if (more) load_b(tap < 8 ? cs : cs + 1, tap < 8 ? tap + 1 : 0);
      const int r = tap / 3, q = tap - r * 3;
      // halo index of row (wrow + i*16 + l16) for this tap: halo[0] is pixel m0 - (W+1)
      const int off = wrow + l16 + (DGRAD ? (2 - r) * W + (2 - q) : r * W + q);
      float a[BK / 4][MT], b[BK / 4][NT];
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        const int k = ks * 4 + g;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float v = Ah[k * LDH + ks * 8 + off + i * 16];
          a[ks][i] = ((vmask[i] >> tap) & 1u) ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) b[ks][j] = Bs[buf][k * LDB + j * 16 + l16];
      }
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      if (more) store_b(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
    if (more_cs) {                                        // every wave is past its last read of Ah (barrier above)
      store_halo();
      __syncthreads();
    }
  }
  conv_epilogue<MT, NT, EPI>(p, acc, m0, n0, wrow, 0, l16, g);
  if (RAWBN && EPI == EPI_ATOMIC && p.tickets != nullptr) {
    // Channel statistics of the finished tile by the LAST K split to arrive (round 5: the separate column reduction was a 7-14 us launch on the
    // dependent chain of every few-pixel layer).  Ordering as in conv3x3_act_bf16_kernel (bf16.hip): this wave's atomics acknowledged
    // (vmcnt(0)) -> workgroup barrier -> device-scope ticket -> the last workgroup reads the tile with agent-scope (sc1) loads, which go
    // through to the memory side where the atomics were performed - no cache-wide fence.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                   // (also: every wave is past its last read of Ah)
    int* const flag = reinterpret_cast<int*>(&Bs[0][0]);
    if (tid == 0) {
      const int per = (ncs + (int)gridDim.z - 1) / (int)gridDim.z, active = (ncs + per - 1) / per;       // splits with an empty slab range left at the top
      const unsigned ticket = __hip_atomic_fetch_add(p.tickets + by, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = ticket == (unsigned)(active - 1);
      if (ticket == (unsigned)(active - 1)) __hip_atomic_store(p.tickets + by, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
    }
    __syncthreads();
    if (!*flag) return;
    // 252 threads = 12 float4 columns x 21 row groups; partial sums through Ah ([21][48][2] floats), then one thread per channel
    const int q4 = tid % 12, rg = tid / 12;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (tid < 252 && n0 + q4 * 4 < p.N) {
      const __amdgpu_buffer_rsrc_t srdO = make_srd(p.out, (unsigned)(((long)(p.M - 1) * p.ldc + p.N) * 4));
      constexpr int RPT = (BM + 20) / 21;
      float4 v[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = rg + 21 * k, m = m0 + r;
        v[k] = (r < BM && m < p.M) ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(srdO, (int)((unsigned)m * (unsigned)(p.ldc * 4) + (unsigned)((n0 + q4 * 4) * 4)), 0, 16))
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        s0.x += v[k].x; s0.y += v[k].y; s0.z += v[k].z; s0.w += v[k].w;
        s1.x = fmaf(v[k].x, v[k].x, s1.x); s1.y = fmaf(v[k].y, v[k].y, s1.y); s1.z = fmaf(v[k].z, v[k].z, s1.z); s1.w = fmaf(v[k].w, v[k].w, s1.w);
      }
    }
    __syncthreads();                                                   // the flag has been read by every thread
    if (tid < 252) {
      *reinterpret_cast<float4*>(&Ah[(rg * 48 + q4 * 4) * 2]) = make_float4(s0.x, s1.x, s0.y, s1.y);
      *reinterpret_cast<float4*>(&Ah[(rg * 48 + q4 * 4) * 2 + 4]) = make_float4(s0.z, s1.z, s0.w, s1.w);
    }
    __syncthreads();
    if (tid < 48 && n0 + tid < p.N) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int r = 0; r < 21; ++r) { a += Ah[(r * 48 + tid) * 2]; b += Ah[(r * 48 + tid) * 2 + 1]; }
      atomicAdd(p.stat0 + n0 + tid, (double)a);
      atomicAdd(p.stat1 + n0 + tid, (double)b);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad kernel: both operands are read along their row (channel) dimension at a fixed pixel
// (K = output pixels).  dW[tap][n][c] += sum_m G[m][n] * f(X[pix(m,tap)][c]).
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, bool TAPS>
__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int AL = (BK * (BM / 4) + 255) / 256, BL = (BK * (BN / 4) + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave / WN) * MT * 16, wcol = (wave % WN) * NT * 16;
  const ConvGeom& G = p.g;
  // 1-D grid.  Work item = (column tile, row tile, K split); the KH*KW taps of one work item read the
  // SAME activation tile, so they are placed 8 block-ids apart: same XCD (blocks are dealt
  // round-robin over the 8 XCDs), adjacent in time -> the tile is fetched from HBM once and the
  // other taps hit that XCD's L2.  Placement only affects speed, never results.
  const int ntaps_g = TAPS ? G.KH * G.KW : 1;
  int tap = 0;
  long item = blockIdx.x;
  if (!TAPS && !p.xcd_flat) {                       // XCD x works through a contiguous range of items (see xcd_block_order)
    const unsigned total = gridDim.x, L = blockIdx.x, x = L & 7u, q = total >> 3, r = total & 7u;
    item = x * q + (x < r ? x : r) + (L >> 3);
  }
  if (TAPS) {
    const long L = blockIdx.x, lane8 = L & 7, seq = L >> 3;
    tap = (int)(seq % ntaps_g);
    item = lane8 + 8 * (seq / ntaps_g);
    if (item >= p.n_items) return;
  }
  const int ctiles = (p.C + BN - 1) / BN, ntiles = (p.N + BM - 1) / BM;
  const int ct = (int)(item % ctiles);
  const long t2 = item / ctiles;
  const int nt = (int)(t2 % ntiles), split = (int)(t2 / ntiles);
  const int c0 = ct * BN, n0 = nt * BM;
  const int tr = TAPS ? tap / G.KW : 0, tq = TAPS ? tap - tr * G.KW : 0;
  const int Mpix = G.B * G.Ho * G.Wo;
  const int nslab_total = (Mpix + BK - 1) / BK;
  const int per = (nslab_total + p.split_k - 1) / p.split_k;
  const int s_begin = split * per, s_end = min(nslab_total, s_begin + per);
  if (s_begin >= s_end) return;
  const __amdgpu_buffer_rsrc_t srdG = make_srd(p.G, p.g_bytes), srdX = make_srd(p.Xs, p.x_bytes);

  // fixed per thread: k (pixel within the slab) and the float4 column group of each load
  unsigned a_voff[AL];  int a_m[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int idx = tid + i * 256;
    const int k = idx / (BM / 4), r4 = idx - k * (BM / 4);
    const bool ok = idx < BK * (BM / 4) && n0 + r4 * 4 < p.N;
    a_m[i] = ok ? s_begin * BK + k : 0x40000000;             // "never < Mpix" for dead lanes
    a_voff[i] = (unsigned)(s_begin * BK + k) * (unsigned)(p.ldg * 4) + (unsigned)((n0 + r4 * 4) * 4);
  }
  float4 bsc[BL], bsh[BL];
  unsigned b_col[BL]; int b_m[BL];
  int pb_[BL], poy[BL], pox[BL];
  const bool bnrelu = p.x_scale != nullptr;
#pragma unroll
  for (int i = 0; i < BL; ++i) {
    const int idx = tid + i * 256;
    const int k = idx / (BN / 4), r4 = idx - k * (BN / 4);
    const bool ok = idx < BK * (BN / 4) && c0 + r4 * 4 < p.C;
    const int m = s_begin * BK + k;
    b_m[i] = ok ? m : 0x40000000;
    b_col[i] = (unsigned)((c0 + r4 * 4) * 4);
    bsc[i] = make_float4(1.f, 1.f, 1.f, 1.f); bsh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bnrelu && ok) { bsc[i] = ld4(p.x_scale + c0 + r4 * 4); bsh[i] = ld4(p.x_shift + c0 + r4 * 4); }
    pb_[i] = 0; poy[i] = 0; pox[i] = m;
    if (TAPS) {
      const int hw = G.Ho * G.Wo;
      pb_[i] = m / hw;
      const int rem = m - pb_[i] * hw;
      poy[i] = rem / G.Wo; pox[i] = rem - poy[i] * G.Wo;
    }
  }

  float4 ra[AL], rb[BL];
  bool rok[BL];
  auto load_slab = [&]() {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const bool okm = a_m[i] < Mpix;
      ra[i] = bld4(srdG, okm ? a_voff[i] : OOB);
      a_m[i] += BK; a_voff[i] += (unsigned)(BK * p.ldg * 4);
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      bool ok = b_m[i] < Mpix;
      int pix = b_m[i];
      if (TAPS) {
        const int iy = poy[i] * G.SH - G.PH + tr, ix = pox[i] * G.SW - G.PW + tq;
        ok = ok & ((unsigned)iy < (unsigned)G.H) & ((unsigned)ix < (unsigned)G.W);
        pix = (pb_[i] * G.H + iy) * G.W + ix;
        pox[i] += BK;                                        // advance the pixel cursor by one slab
        while (pox[i] >= G.Wo) { pox[i] -= G.Wo; ++poy[i]; }
        while (poy[i] >= G.Ho) { poy[i] -= G.Ho; ++pb_[i]; }
      }
      rok[i] = ok;
      rb[i] = bld4(srdX, ok ? (unsigned)pix * (unsigned)(p.ldx * 4) + b_col[i] : OOB);
      b_m[i] += BK;
    }
  };
  auto store_slab = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int idx = tid + i * 256;
      if (idx < BK * (BM / 4)) {
        const int k = idx / (BM / 4), r4 = idx - k * (BM / 4);
        *reinterpret_cast<float4*>(&As[buf][k * LDA + r4 * 4]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int idx = tid + i * 256;
      if (idx < BK * (BN / 4)) {
        const int k = idx / (BN / 4), r4 = idx - k * (BN / 4);
        float4 v = rb[i];
        if (bnrelu) { v = bnrelu4(v, bsc[i], bsh[i]); if (!rok[i]) v = make_float4(0.f, 0.f, 0.f, 0.f); }
        *reinterpret_cast<float4*>(&Bs[buf][k * LDB + r4 * 4]) = v;
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_slab();
  store_slab(0);
  __syncthreads();
  int buf = 0;
  for (int s = s_begin; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    if (more) load_slab();
    mma_slab<MT, NT>(As[buf], Bs[buf], LDA, LDB, wrow, wcol, l16, g, acc);
    if (more) store_slab(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  float* base = p.dW + (long)tap * p.wtap;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c = c0 + wcol + j * 16 + l16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wrow + i * 16 + g * 4 + r;
        if (c < p.C && n < p.N) atomicAdd(base + (long)n * p.ldw + c, acc[i][j][r]);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// optional per-launch timing of the MFMA kernels (bench.py roofline): HIP events recorded on the
// launch stream around every conv kernel; read back (and summed) by rdm_profile_read().
// ---------------------------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int kind; double bytes; };
struct Prof {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
} g_prof;
struct ProfScope {
  hipStream_t s; double flops; hipEvent_t a{}, b{}; bool on; int kind = 0; double bytes = 0;
  ProfScope(hipStream_t s_, double f) : s(s_), flops(f), on(g_prof.on) { if (on) { a = g_prof.get(); b = g_prof.get(); hipEventRecord(a, s); } }
  ~ProfScope() { if (on) { hipEventRecord(b, s); g_prof.recs.push_back({a, b, flops, kind, bytes}); } }
};
// per-kernel breakdown of the last profile_read()
constexpr int PROF_KINDS = 19;
const char* const kProfKindName[PROF_KINDS] = {
    "conv_fwd_kernel (forward / weights k-contiguous)", "conv_fwd_kernel (dgrad / weights k-strided)", "conv3x3_halo_kernel (forward)",
    "conv3x3_halo_kernel (dgrad)", "conv_wgrad_kernel (1x1)", "conv_wgrad_kernel (taps)", "(unused since round 4: the 3x3 row weight-gradient kernel was deleted)",
    "gemm_bf16_kernel (1x1 forward, bf16 MFMA)", "conv3x3_bf16_kernel (3x3 forward, bf16 MFMA)",
    "conv3x3_wino_fwd_kernel (Winograd F(2x2,3x3) forward; FLOPs = the direct convolution's)",
    "conv3x3_wino_wgrad_kernel (Winograd F(3x3,2x2) weight gradient; FLOPs = the direct convolution's)",
    "gemm_panel_bf16_kernel (short-K 1x1 forward, persistent panels, bf16 MFMA)", "conv3x3_act_bf16_kernel (3x3 forward on an activated input, LDS-DMA, bf16 MFMA)",
    "xs_wgrad1x1_kernel (1x1 weight gradient, bf16x3 split MFMA; FLOPs = the f32 product's)", "xs_dgrad3x3_kernel (3x3 input gradient, bf16x3 split MFMA; FLOPs = the f32 product's)",
    "xs_wgrad3x3_kernel (3x3 weight gradient, bf16x3 split MFMA; FLOPs = the f32 product's)", "xs_dgrad1x1_kernel (1x1 input gradient, bf16x3 split MFMA; FLOPs = the f32 product's)",
    "xs_fwd1x1_kernel (1x1 forward, bf16x6 three-way split MFMA; FLOPs = the f32 product's)",
    "conv3x3_wino_x6_kernel (Winograd F(2x2,3x3) forward, bf16x6 three-way split MFMA; FLOPs = the direct convolution's)"};
double g_kind_ms[PROF_KINDS], g_kind_flops[PROF_KINDS], g_kind_bytes[PROF_KINDS];
int g_kind_n[PROF_KINDS];
}  // namespace

// the same bracket for launchers outside this file (bf16.hip)
void* profile_begin(hipStream_t s, double flops, int kind, double bytes) {
  if (!g_prof.on) return nullptr;
  ProfScope* t = new ProfScope(s, flops);
  t->kind = kind;
  t->bytes = bytes;
  return t;
}
double profile_kind_bytes(int kind) { return kind >= 0 && kind < PROF_KINDS ? g_kind_bytes[kind] : -1.0; }
void profile_end(void* ticket, hipStream_t) { delete static_cast<ProfScope*>(ticket); }

static hipEvent_t g_prof_base = nullptr;
void profile_enable(bool on) {
  if (on && !g_prof.on) {              // common time origin so intervals from different streams can be merged
    if (!g_prof_base) hipEventCreate(&g_prof_base);
    hipEventRecord(g_prof_base, nullptr);
    hipEventSynchronize(g_prof_base);
  }
  g_prof.on = on;
}
// ms_sum: sum of kernel durations; ms_union: length of the union of their [start,end] intervals
// (weight-gradient kernels run concurrently on the library's side stream, so the sum over-counts
// the time the chip spends in these kernels)
int profile_read(double* ms_sum, double* ms_union, double* flops, int* launches) {
  double t = 0, f = 0;
  std::vector<std::pair<float, float>> iv;
  iv.reserve(g_prof.recs.size());
  for (int k = 0; k < PROF_KINDS; ++k) { g_kind_ms[k] = 0; g_kind_flops[k] = 0; g_kind_n[k] = 0; g_kind_bytes[k] = 0; }
  for (auto& r : g_prof.recs) {
    RDM_HIP_OK(hipEventSynchronize(r.b));
    float a0 = 0, b0 = 0;
    RDM_HIP_OK(hipEventElapsedTime(&a0, g_prof_base, r.a));
    RDM_HIP_OK(hipEventElapsedTime(&b0, g_prof_base, r.b));
    iv.emplace_back(a0, b0);
    t += b0 - a0; f += r.flops;
    g_kind_ms[r.kind] += b0 - a0; g_kind_flops[r.kind] += r.flops; g_kind_n[r.kind] += 1; g_kind_bytes[r.kind] += r.bytes;
    g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b);
  }
  std::sort(iv.begin(), iv.end());
  double u = 0; float cs = 0, ce = -1;
  for (auto& x : iv) {
    if (ce < 0) { cs = x.first; ce = x.second; }
    else if (x.first <= ce) { if (x.second > ce) ce = x.second; }
    else { u += ce - cs; cs = x.first; ce = x.second; }
  }
  if (ce >= 0) u += ce - cs;
  if (ms_sum) *ms_sum = t;
  if (ms_union) *ms_union = u;
  if (flops) *flops = f;
  if (launches) *launches = (int)g_prof.recs.size();
  g_prof.recs.clear();
  return 0;
}
int profile_kind(int kind, const char** name, double* ms_sum, double* flops, int* launches) {
  RDM_CHECK_ARG(kind >= 0 && kind < PROF_KINDS, "profile_kind: kind %d out of range [0, %d)", kind, PROF_KINDS);
  if (name) *name = kProfKindName[kind];
  if (ms_sum) *ms_sum = g_kind_ms[kind];
  if (flops) *flops = g_kind_flops[kind];
  if (launches) *launches = g_kind_n[kind];
  return 0;
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
#ifdef RDM_DEV_VARIANTS
int g_variant = 0;      // A/B switch for in-process kernel comparisons (rdm_debug_variant)
#endif

static const char* epi_name(int e) {
  static const char* const n[5] = {"STORE", "STORE_STATS", "MASK_STATS", "ATOMIC", "MASK_STATS_ATOMIC"};
  return e >= 0 && e < 5 ? n[e] : "?";
}

int pick_split_k(long tiles, long kslabs, int slots) {
  // `slots` = workgroups resident on the chip at once (256 CUs x blocks/CU).  The grid runs in
  // ceil(blocks / slots) rounds; a last round that is mostly empty wastes up to one round, so among
  // the splits that give between LO and HI rounds pick the one whose last round is fullest (ties: fewer
  // splits).  Every split pays a prologue (row decode) and an atomic epilogue, so keep >= MINS slabs each.
  // (MINS, LO, HI) = (16, 1, 3): swept on one MI355X over the B=16 228x304 step - (8, 2, 5) was 7 % slower
  // (too many short splits on the M=4560 / M=1280 layers), 32 slabs 4 % slower, 64 slabs 16 % slower
  constexpr int MINS = 16;
  constexpr double LO = 1.0, HI = 3.0;      // re-swept in round 3 with the Winograd kernels in place (LO 0.6-0.85, HI 2-3): 71.9-72.6 ms, no change
  if (tiles >= 6L * slots) return 1;
  long cap = kslabs / MINS;
  if (cap < 1) cap = 1;
  if (cap > 128) cap = 128;
  long lo = (long)((LO * slots + tiles - 1) / tiles), hi = (long)((HI * slots + tiles - 1) / tiles);
  if (lo < 1) lo = 1;
  if (lo > cap) lo = cap;
  if (hi > cap) hi = cap;
  if (hi < lo) hi = lo;
  int best = (int)lo;
  double best_eff = -1;
  for (long sp = lo; sp <= hi; ++sp) {
    const double rounds = (double)(tiles * sp) / slots;
    const double eff = rounds / (double)((long)(rounds + 0.999999));
    if (eff > best_eff + 0.02) { best_eff = eff; best = (int)sp; }
  }
  return best;
}

template <int MT, int NT, int WM, int WN, bool TAPS, bool BK_, int EPI>
static void launch_fwd_cfg(const FwdArgs& a, int split, hipStream_t s) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN;
  dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), split);
  hipLaunchKernelGGL((conv_fwd_kernel<MT, NT, WM, WN, TAPS, BK_, EPI>), grid, dim3(256), 0, s, a);
}

template <bool TAPS, bool BK_, int EPI>
static void launch_fwd_epi(const FwdArgs& a, int cfg, int split, hipStream_t s) {
  if (cfg == 0) launch_fwd_cfg<4, 3, 4, 1, TAPS, BK_, EPI>(a, split, s);        // 256 x 48
  else if (cfg == 1) launch_fwd_cfg<4, 3, 2, 2, TAPS, BK_, EPI>(a, split, s);   // 128 x 96
  else if (cfg == 2) launch_fwd_cfg<2, 3, 2, 2, TAPS, BK_, EPI>(a, split, s);   //  64 x 96
  else launch_fwd_cfg<2, 3, 4, 1, TAPS, BK_, EPI>(a, split, s);                 // 128 x 48
}

static int launch_conv_fwd_impl(const FwdArgs& a_in, bool b_kstrided, Epilogue epi, hipStream_t s);

int launch_conv_fwd(const FwdArgs& a_in, bool b_kstrided, Epilogue epi, hipStream_t s) {
  if (!t_deterministic) return launch_conv_fwd_impl(a_in, b_kstrided, epi, s);
  // deterministic mode: no K split, statistics / gate as separate ordered passes over the stored result
  FwdArgs a = a_in;
  RDM_CHECK_ARG(!a.accumulate && !a.add_out, "deterministic mode: accumulate / add_out launches are not available (the plan disables layer pipelining)");
  a.split_k = 1;
  const Epilogue e2 = (epi == EPI_STORE_STATS || epi == EPI_MASK_STATS) ? EPI_STORE : epi;
  const int rc = launch_conv_fwd_impl(a, b_kstrided, e2, s);
  if (rc < 0) return rc;
  if (epi == EPI_STORE_STATS) return launch_colstats(a.out, a.ldc, a.M, a.N, a.stat0, a.stat1, s);
  if (epi == EPI_MASK_STATS) return launch_mask_stats(a.out, a.ldc, a.X, a.ldx, a.x_scale, a.x_shift, a.M, a.N, a.stat0, a.stat1, s);
  return rc;
}

static int launch_conv_fwd_impl(const FwdArgs& a_in, bool b_kstrided, Epilogue epi, hipStream_t s) {
  FwdArgs a = a_in;
  RDM_CHECK_ARG(!a.out_bf16 && !a.a_bf16 && !a.acc_scaled, "conv: bf16 rows / the scaled accumulating epilogue belong to the kernels of xsplit.hip only");
  RDM_CHECK_ARG(a.C % 16 == 0 && a.C > 0, "conv: contracted channels (%d) must be a positive multiple of 16", a.C);
  RDM_CHECK_ARG(a.lda % 4 == 0 && a.ldw % 4 == 0 && a.wtap % 4 == 0, "conv: strides must be multiples of 4 floats");
  RDM_CHECK_ARG(!b_kstrided || a.N % 4 == 0, "dgrad: N (%d) must be a multiple of 4", a.N);
  RDM_CHECK_ARG(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.Wt & 15) == 0, "conv: operands must be 16-byte aligned");
  RDM_CHECK_ARG(a.g.dir == 1 || (a.g.SH == 1 && a.g.SW == 1), "dgrad gather supports stride 1 only");
  RDM_CHECK_ARG(a.M == a.g.B * a.g.Ho * a.g.Wo, "conv: M (%d) != B*Ho*Wo", a.M);
  const bool taps = !(a.g.KH == 1 && a.g.KW == 1 && a.g.SH == 1 && a.g.SW == 1 && a.g.PH == 0 && a.g.PW == 0 &&
                      a.g.H == a.g.Ho && a.g.W == a.g.Wo);
  const long kslabs = (long)(a.C / 16) * (taps ? a.g.KH * a.g.KW : 1);
  // tile choice: 128x96 when N is a multiple of 96 and the grid is large, else 256x48 (every channel
  // count of this network is a multiple of 48); short grids are filled by split-K
  const long t0 = (long)cdiv(a.M, 256) * cdiv(a.N, 48), t1 = (long)cdiv(a.M, 128) * cdiv(a.N, 96);
  int cfg = 0;
  if (a.N % 96 == 0 && (t1 >= 512 || a.M <= 128)) cfg = 1;
  if (t1 >= 512 && (double)cdiv(a.N, 96) * 96 <= 1.04 * a.N) cfg = 1;      // a ragged last N tile is fine up to 4 % waste (A/B: 110.7 vs 107.6 TF)
  if (a.M <= 64) cfg = 2;
  // few rows (dense_e4: 4560, decoder: 1280): 128 x 48 tiles give 4x the workgroups of 256 x 48 before any split-K and no ragged
  // column tile; measured 20-45 % faster than 256 x 48 + deeper split-K on the 1x1 convs of those blocks (64 x 96 ties)
  if (!taps && a.M <= 32768 && a.M > 64 && g_variant != 16) cfg = 3;
  if (!taps && cfg == 0 && g_variant != 20) cfg = 3;       // also at dense_e2 size: 1x1 dgrad 121.1 vs 116.3 TFLOP/s
  // (a 64x96 wave tile - 256x96 block, 2 waves/SIMD - was measured: +1 % on the 1x1 forward, -21 % on the 3x3 dgrad)
  const long tiles = cfg == 0 ? t0 : cfg == 1 ? t1 : cfg == 2 ? (long)cdiv(a.M, 64) * cdiv(a.N, 96) : (long)cdiv(a.M, 128) * cdiv(a.N, 48);
  int split = a.split_k > 0 ? a.split_k : pick_split_k(tiles, kslabs, 256 * 4);
  if (epi == EPI_STORE_STATS) split = 1;                 // sum of squares is not linear in the K-partials
  if (epi == EPI_STORE && a.bias != nullptr) split = 1;
  if (a.accumulate) {
    RDM_CHECK_ARG(epi == EPI_STORE && a.bias == nullptr, "conv: accumulate mode supports the plain epilogue only");
    epi = EPI_ATOMIC;
  } else if (split > 1) {   // split-K: f32 atomics into a zeroed (possibly strided) output slice
    epi = epi == EPI_MASK_STATS ? EPI_MASK_STATS_ATOMIC : EPI_ATOMIC;
    if (int zrc = launch_zero_rows(a.out, a.M, a.N, a.ldc, s)) return zrc;
  }
  a.split_k = split;
  a.xcd_flat = g_variant == 11 || a.N <= 48;      // a single column tile has nothing to share (and the 3x3 forward measured 4 % slower remapped)
  {
    const long npix = (long)a.g.B * a.g.H * a.g.W, nt = taps ? a.g.KH * a.g.KW : 1;
    const long ab = ((npix - 1) * a.lda + a.C) * 4;
    const long wb = b_kstrided ? ((nt - 1) * a.wtap + (long)(a.C - 1) * a.ldw + a.N) * 4 : ((nt - 1) * a.wtap + (long)(a.N - 1) * a.ldw + a.C) * 4;
    if (ab >= 0xFFFFFFFFL || wb >= 0xFFFFFFFFL) { set_error("conv: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
    a.a_bytes = (unsigned)ab; a.w_bytes = (unsigned)wb;
  }
  ProfScope prof(s, 2.0 * a.M * a.N * (double)kslabs * 16);
  prof.kind = b_kstrided ? 1 : 0;

  const bool halo_fwd = !b_kstrided && a.g.dir == 1, halo_dgrad = b_kstrided && a.g.dir == -1;
  if ((halo_fwd || halo_dgrad) && g_variant != 7 && taps && a.g.KH == 3 && a.g.KW == 3 && a.g.SH == 1 && a.g.SW == 1 && a.g.PH == 1 && a.g.PW == 1 &&
      a.g.H == a.g.Ho && a.g.W == a.g.Wo && 2 * (a.g.W + 1) <= 256 &&   // halo <= 512 pixels = 8 float4 per thread
      (long)a.g.B * a.g.H * a.g.W < (1L << 30)) {
    prof.kind = halo_dgrad ? 3 : 2;
    // LDS halo kernel; K is split over whole channel slabs (a split keeps its 9-tap groups together).
    // `epi` / `split` were already resolved above (split > 1 => atomic epilogue, output zeroed).
    // few pixels (dense_e4, decoder): 128-pixel tiles double the workgroups before any split
    const bool small = a.M <= 8192 && 128 + 2 * (a.g.W + 1) <= 256 && g_variant != 21;
    const int bm = small ? 128 : 256;
    // the split chosen above (for the generic tiling) is kept for 128-pixel tiles too: re-picking it for the doubled tile count
    // was measured at whole-step level and lost (81.5 vs 79.0 ms/step at NYU B=16)
    int sp = split;
    if (sp > a.C / 16) sp = a.C / 16;
    {
      // row-grouped block order (xcd_block_order) when the weights of all column tiles do not stay in one 4 MB L2 but a group of
      // 8 activation (halo) tiles does: dense_e2's dgrad, 57 column tiles x 83 KB.  Measured there (rocprofv3 FETCH_SIZE, calibrated:
      // tools/fetch_calibration.py): L2 fills of the main loop 1.22 GB -> 0.18 GB per launch, whole kernel 2.55x -> 1.21x its
      // compulsory bytes; the time does not move (1.468 ms either way - the kernel is bound by instruction issue, not by the fabric).
      const double w_total = (double)a.N * 9.0 * a.C * 4.0, a_tile = (double)(bm + 2 * (a.g.W + 1)) * a.C * 4.0;
      if (!a.xcd_flat && sp == 1 && g_variant != 29 && w_total > 2.5e6 && 8.0 * a_tile <= 2.0e6) a.xcd_flat = 8;
    }
    dim3 grid(cdiv(a.N, 48), cdiv(a.M, bm), sp);
    const int hl = cdiv(bm + 2 * (a.g.W + 1), 64);           // 256-pixel tiles: 5 (W <= 31) .. 8 (<= 127); 128-pixel tiles: 3, 4
    RDM_CENSUS("conv3x3_halo_kernel/%s/px%d/hl%d/%s", halo_dgrad ? "dgrad" : "fwd", bm, small ? std::max(hl, 3) : std::max(hl, 5), epi_name(epi));
    if (a.a_sum != nullptr) {                     // RAW BatchNorm prologue: the 128-pixel forward tiles of the few-pixel blocks only
      RDM_CHECK_ARG(small && halo_fwd && a.C <= RAWBN_MAX_C && (epi == EPI_STORE || epi == EPI_ATOMIC), "conv3x3: the raw BatchNorm prologue is built for the 128-pixel forward tiles, C <= %d", RAWBN_MAX_C);
      RDM_CHECK_ARG(a.tickets == nullptr || (epi == EPI_ATOMIC && a.stat0 && a.stat1 && a.N <= 48 && a.ldc % 4 == 0 && ((uintptr_t)a.out & 15) == 0),
                    "conv3x3: in-launch statistics belong to the accumulating raw-BatchNorm form with <= 48 outputs, rows 16-byte aligned");
      RDM_CENSUS("conv3x3_halo_kernel/fwd/px128/hl%d/%s/rawbn%s", std::max(hl, 3), epi_name(epi), a.tickets ? "/stats" : "");
      if (epi == EPI_STORE) { if (hl <= 3) hipLaunchKernelGGL((conv3x3_halo_kernel<EPI_STORE, 3, false, 2, true>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((conv3x3_halo_kernel<EPI_STORE, 4, false, 2, true>), grid, dim3(256), 0, s, a); }
      else { if (hl <= 3) hipLaunchKernelGGL((conv3x3_halo_kernel<EPI_ATOMIC, 3, false, 2, true>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((conv3x3_halo_kernel<EPI_ATOMIC, 4, false, 2, true>), grid, dim3(256), 0, s, a); }
      RDM_LAUNCH_OK();
      return 0;
    }
#define RDM_HALO2(E_, D_)                                                                                       \
    if (small) {                                                                                                 \
      if (hl <= 3) hipLaunchKernelGGL((conv3x3_halo_kernel<E_, 3, D_, 2>), grid, dim3(256), 0, s, a);           \
      else hipLaunchKernelGGL((conv3x3_halo_kernel<E_, 4, D_, 2>), grid, dim3(256), 0, s, a);                   \
    } else if (hl <= 5) hipLaunchKernelGGL((conv3x3_halo_kernel<E_, 5, D_, 4>), grid, dim3(256), 0, s, a);      \
    else if (hl == 6) hipLaunchKernelGGL((conv3x3_halo_kernel<E_, 6, D_, 4>), grid, dim3(256), 0, s, a);        \
    else if (hl == 7) hipLaunchKernelGGL((conv3x3_halo_kernel<E_, 7, D_, 4>), grid, dim3(256), 0, s, a);        \
    else hipLaunchKernelGGL((conv3x3_halo_kernel<E_, 8, D_, 4>), grid, dim3(256), 0, s, a);
#define RDM_HALO(E_) if (halo_dgrad) { RDM_HALO2(E_, true) } else { RDM_HALO2(E_, false) }
    switch (epi) {
      case EPI_STORE: RDM_HALO(EPI_STORE) break;
      case EPI_STORE_STATS: RDM_HALO(EPI_STORE_STATS) break;
      case EPI_MASK_STATS: RDM_HALO(EPI_MASK_STATS) break;
      case EPI_MASK_STATS_ATOMIC: RDM_HALO(EPI_MASK_STATS_ATOMIC) break;
      default: RDM_HALO(EPI_ATOMIC) break;
    }
#undef RDM_HALO
#undef RDM_HALO2
    RDM_LAUNCH_OK();
    return 0;
  }
  // mid-size 1x1 forward (dense_e3: 17 632 pixels x 1392 channels): 256 x 128 macro tile on the LDS-DMA pipeline (conv1x1_dma256_kernel).
  // In-process A/B with the BN-ReLU prologue and the statistics epilogue (tools/conv_microbench.py): K = 384 / 576 / 720 / 912
  // 100.4 / 108.0 / 112.5 / 115.0 vs 95.2 / 101.4 / 104.3 / 105.8 TFLOP/s (+5.5 .. 8.7 %); at dense_e2 size (69 312 pixels, K <= 336)
  // the two are level (109.5 vs 110.6 at K = 336), so the 128 x 96 register-staged kernel stays there.
  if (!taps && !b_kstrided && split == 1 && (epi == EPI_STORE || epi == EPI_STORE_STATS) && !a.add_out && a.bias == nullptr && a.M >= 16384 && a.M < 32768 &&
      a.N >= 512 && g_variant != 38) {
    // 512 workgroups are resident (2 per CU): dense_e3 at B = 16 has 759 tiles of 256 x 128 - a full round and a half-empty one - but 1012
    // of 192 x 128: two nearly full rounds of workgroups that are a quarter shorter.  Cost = rounds x tile rows.
    const long t256 = (long)cdiv(a.N, 128) * cdiv(a.M, 256), t192 = (long)cdiv(a.N, 128) * cdiv(a.M, 192);
    const bool m192 = cdiv(t192, 512L) * 192 < cdiv(t256, 512L) * 256 && g_variant != 39;
    dim3 grid(cdiv(a.N, 128), cdiv(a.M, m192 ? 192 : 256), 1);
    const bool bn = a.a_scale != nullptr;
    RDM_CENSUS("conv1x1_dma256_kernel/%s/bn%d%s", epi_name(epi), bn ? 1 : 0, m192 ? "/m192" : "");
#define RDM_D256(EPI_, BN_) do { if (m192) hipLaunchKernelGGL((conv1x1_dma256_kernel<EPI_, BN_, 6>), grid, dim3(256), 0, s, a); else hipLaunchKernelGGL((conv1x1_dma256_kernel<EPI_, BN_, 8>), grid, dim3(256), 0, s, a); } while (0)
    if (epi == EPI_STORE) { if (bn) RDM_D256(EPI_STORE, true); else RDM_D256(EPI_STORE, false); }
    else { if (bn) RDM_D256(EPI_STORE_STATS, true); else RDM_D256(EPI_STORE_STATS, false); }
#undef RDM_D256
    RDM_LAUNCH_OK();
    return 0;
  }
  if (a.a_sum != nullptr) {                       // RAW BatchNorm prologue: the 128 x 48 1x1 forward of the few-pixel blocks (conv1 "part B")
    RDM_CHECK_ARG(!taps && !b_kstrided && cfg == 3 && a.C <= RAWBN_MAX_C && (epi == EPI_STORE_STATS || epi == EPI_STORE || epi == EPI_ATOMIC),
                  "conv: the raw BatchNorm prologue is built for the 128 x 48 1x1 forward, C <= %d", RAWBN_MAX_C);
    RDM_CENSUS("conv_fwd_kernel/fwd/1x1/tile128x48/%s/rawbn", epi_name(epi));
    dim3 grid(cdiv(a.N, 48), cdiv(a.M, 128), split);
    if (epi == EPI_STORE_STATS) hipLaunchKernelGGL((conv_fwd_kernel<2, 3, 4, 1, false, false, EPI_STORE_STATS, true>), grid, dim3(256), 0, s, a);
    else if (epi == EPI_STORE) hipLaunchKernelGGL((conv_fwd_kernel<2, 3, 4, 1, false, false, EPI_STORE, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_fwd_kernel<2, 3, 4, 1, false, false, EPI_ATOMIC, true>), grid, dim3(256), 0, s, a);
    RDM_LAUNCH_OK();
    return 0;
  }
  RDM_CENSUS("conv_fwd_kernel/%s/%s/tile%s/%s", b_kstrided ? "dgrad" : "fwd", taps ? "taps" : "1x1",
             cfg == 0 ? "256x48" : cfg == 1 ? "128x96" : cfg == 2 ? "64x96" : "128x48", epi_name(epi));
#define RDM_FWD_DISPATCH(TAPS_, BKS_)                                                           \
  switch (epi) {                                                                               \
    case EPI_STORE: launch_fwd_epi<TAPS_, BKS_, EPI_STORE>(a, cfg, split, s); break;            \
    case EPI_STORE_STATS: launch_fwd_epi<TAPS_, BKS_, EPI_STORE_STATS>(a, cfg, split, s); break; \
    case EPI_MASK_STATS: launch_fwd_epi<TAPS_, BKS_, EPI_MASK_STATS>(a, cfg, split, s); break;  \
    case EPI_MASK_STATS_ATOMIC: launch_fwd_epi<TAPS_, BKS_, EPI_MASK_STATS_ATOMIC>(a, cfg, split, s); break;  \
    default: launch_fwd_epi<TAPS_, BKS_, EPI_ATOMIC>(a, cfg, split, s); break;                  \
  }
  if (taps) { if (b_kstrided) { RDM_FWD_DISPATCH(true, true) } else { RDM_FWD_DISPATCH(true, false) } }
  else      { if (b_kstrided) { RDM_FWD_DISPATCH(false, true) } else { RDM_FWD_DISPATCH(false, false) } }
#undef RDM_FWD_DISPATCH
  RDM_LAUNCH_OK();
  return 0;
}

template <int MT, int NT, int WM, int WN>
static void launch_wgrad_cfg(const WgradArgs& a, bool taps, hipStream_t s) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN;
  const int ntaps = taps ? a.g.KH * a.g.KW : 1;
  WgradArgs b = a;
  b.n_items = (long)cdiv(a.C, BN) * cdiv(a.N, BM) * a.split_k;
  const long padded = (b.n_items + 7) / 8 * 8;               // whole groups of 8 so every (item, tap) pair exists
  dim3 grid((unsigned)(taps ? padded * ntaps : b.n_items));
  if (taps) hipLaunchKernelGGL((conv_wgrad_kernel<MT, NT, WM, WN, true>), grid, dim3(256), 0, s, b);
  else hipLaunchKernelGGL((conv_wgrad_kernel<MT, NT, WM, WN, false>), grid, dim3(256), 0, s, b);
}

int launch_conv_wgrad(const WgradArgs& a_in, hipStream_t s) {
  WgradArgs a = a_in;
  if (t_deterministic) a.split_k = 1;                     // one workgroup per output element: its atomic add onto the zeroed gradient is exact
  RDM_CHECK_ARG(a.N % 4 == 0 && a.C % 4 == 0, "wgrad: N (%d) and C (%d) must be multiples of 4", a.N, a.C);
  RDM_CHECK_ARG(a.ldg % 4 == 0 && a.ldx % 4 == 0, "wgrad: strides must be multiples of 4 floats");
  RDM_CHECK_ARG(((uintptr_t)a.G & 15) == 0 && ((uintptr_t)a.Xs & 15) == 0, "wgrad: operands must be 16-byte aligned");
  const bool taps = !(a.g.KH == 1 && a.g.KW == 1 && a.g.SH == 1 && a.g.SW == 1 && a.g.PH == 0 && a.g.PW == 0 &&
                      a.g.H == a.g.Ho && a.g.W == a.g.Wo);
  const int ntaps = taps ? a.g.KH * a.g.KW : 1;
  const long Mpix = (long)a.g.B * a.g.Ho * a.g.Wo;
  // split-precision (bf16x3) kernel where the caller allows it: long contractions only (the plan's many-pixel blocks)
  if (a.xsplit && !t_deterministic && !taps && Mpix >= 1024 && xs_wgrad1x1_supported(a)) return launch_xs_wgrad1x1(a, s);
  if (a.xsplit && !t_deterministic && taps && Mpix >= 1024 && xs_wgrad3x3_supported(a)) return launch_xs_wgrad3x3(a, s);
  RDM_CHECK_ARG(!a.g_bf16, "wgrad: bf16 gradient rows are read by the one-product 1x1 kernel of xsplit.hip only");
  const long kslabs = (Mpix + 15) / 16;
  const bool narrow = a.N <= 48;                      // 3x3 convs of the dense layers: 48 output channels
  // a ragged last column tile of the 128 x 96 config wastes up to 25 % of the MFMAs (C = 144: 2 x 96); 256 x 48 tiles fit every
  // C that is a multiple of 48 (PMC: the 1x1 wgrad ran its MFMA pipe at 78 % for 66 % useful - the gap was this waste)
  const bool tall = !narrow && g_variant != 13 && a.C % 96 != 0 && (double)cdiv(a.C, 96) * 96 > 1.04 * a.C;
  // few pixels to contract over (dense_e4, decoder): half-size tiles (1x1: 93 vs 116 us at M = 4560, C = 1224; 3x3: 47 vs 55 us)
  const bool few = Mpix <= 8192 && g_variant != 23;
  const long tiles = few ? (narrow ? (long)cdiv(a.C, 128) * cdiv(a.N, 48) : (long)cdiv(a.C, 48) * cdiv(a.N, 128)) * ntaps :
                     narrow ? (long)cdiv(a.C, 256) * cdiv(a.N, 48) * ntaps : tall ? (long)cdiv(a.C, 48) * cdiv(a.N, 256) * ntaps : (long)cdiv(a.C, 96) * cdiv(a.N, 128) * ntaps;
  // (rounds 1-3 had a dedicated 3x3 "row" kernel for >= 16 K pixels here - one kernel row of taps per block; the plan's many-pixel blocks moved to
  // Winograd in round 3 and to the split-precision kernel of xsplit.hip in round 4, and the row kernel was deleted: the generic tap kernel
  // serves the operator API and RDM_NET_OPT_DIRECT_3X3 at every size)
  if (a.split_k <= 0) a.split_k = pick_split_k(tiles, kslabs, 256 * 3);
  {
    const long gb = ((Mpix - 1) * a.ldg + a.N) * 4, xb = (((long)a.g.B * a.g.H * a.g.W - 1) * a.ldx + a.C) * 4;
    if (gb >= 0xFFFFFFFFL || xb >= 0xFFFFFFFFL) { set_error("wgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
    a.g_bytes = (unsigned)gb; a.x_bytes = (unsigned)xb;
  }
  a.xcd_flat = g_variant == 11;
  ProfScope prof(s, 2.0 * (double)Mpix * a.N * a.C * ntaps);
  prof.kind = taps ? 5 : 4;
  RDM_CENSUS("%s/%s/%s", taps ? "conv_wgrad_kernel/taps" : "conv_wgrad_kernel/1x1",
             few && !narrow ? "128x48" : few ? "48x128" : narrow ? "48x256" : tall ? "256x48" : "128x96", a.split_k > 1 ? "splitK" : "split1");
  if (few && !narrow) launch_wgrad_cfg<2, 3, 4, 1>(a, taps, s);   // 128 x 48
  else if (few) launch_wgrad_cfg<3, 2, 1, 4>(a, taps, s);                //  48 x 128
  else if (narrow) launch_wgrad_cfg<3, 4, 1, 4>(a, taps, s);   //  48 x 256
  else if (tall) launch_wgrad_cfg<4, 3, 4, 1>(a, taps, s);  // 256 x 48: input-channel counts are multiples of 48, not of 96
  else launch_wgrad_cfg<4, 3, 2, 2>(a, taps, s);          // 128 x 96
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
