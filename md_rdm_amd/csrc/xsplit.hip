// Split-precision ("bf16x3") MFMA kernels for the BACKWARD GEMMs of the dense blocks with many pixels (dense_e2 / dense_e3).
//
// Replaces, for those layers, the exact-f32 MFMA kernels of csrc/igemm.hip in the autograd-generated weight / input gradients of
// torchvision's _DenseLayer.conv1 / conv2 (reached from network/RDM_Net.py:526,528).  gfx950 has no reduced-precision fast path for
// f32 operands (no xf32), and v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate.  A float32 value splits EXACTLY into
//     x = hi + lo + r,   hi = bf16(x),  lo = bf16(x - hi),  |r| <= 2^-17 |x|
// and the product of two such values is  a_hi b_hi + a_hi b_lo + a_lo b_hi  up to terms of relative size 2^-16: three
// v_mfma_f32_16x16x32_bf16 (16 cycles each for 16x16x32 MACs) instead of eight v_mfma_f32_16x16x4_f32 (32 cycles each) - 5.3x less
// matrix-pipe time - with float32 accumulation and every bf16 x bf16 product exact in float32.  Measured error against float64:
// 4-6e-6 of the output's maximum (the f32 MFMA kernels: 0.4-1.6e-6); the operator tests hold these kernels to the same 2e-5 as the
// f32 ones (tests/test_gpu_xsplit.py).  Only GRADIENTS go through here: the forward pass, its ordinal indices and its 1e-4 parity
// stay on exact f32.
//
// MI355X mapping:
//  * both operands of a weight gradient are pixel-major ([pixel][channel]) and the contraction runs over PIXELS, i.e. the MFMA's k index is
//    the strided one.  The split values go to LDS as [32 pixels][16-channel tiles] bf16 images and come back through
//    ds_read_b64_tr_b16 - the hardware transpose read: a 16-lane group fetches a 4-pixel x 16-channel block and every lane receives its
//    channel's 4 pixels - so no lane shuffles and no transposed copy in HBM.
//  * image layout: subtile (8 pixels x 16 channels, 256 B) at 256 * (pixel_block * NCT + channel_tile); inside it pixel row p sits at
//    32 * (p ^ 4 * (pixel_block & 1)).  The two 4x16 blocks a 32-lane half reads (k groups 8 pixels apart) then cover all 64 banks
//    (conflict-free), and the staging stores (a 16-lane group writes a 4-pixel x 16-channel patch as 8-byte pieces) cover all 32.
//  * staging = global -> registers (next slab in flight under the MFMAs) -> BatchNorm + ReLU -> split -> LDS; one image set, two
//    barriers per slab, two workgroups per CU so one's staging runs beside the other's MFMAs.
#include <algorithm>
#include <stdlib.h>

#include "rdm_common.h"
#include "elementwise.h"
#include "xsplit.h"
#include "xsplit_dev.h"

namespace rdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr unsigned XOOB = 0xFFFFFFFFu;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xsrd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// one MFMA operand fragment (8 consecutive k of one row / column) from a [pixel][channel] image: two transposed 8-byte reads
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base0, const unsigned char* base1, int imm) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base0 + imm));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base1 + imm));
  const s16x8 ab = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return __builtin_bit_cast(bf16x8, ab);
#else
  return bf16x8{};
#endif
}

// ---------------------------------------------------------------------------------------------
// 1x1 weight gradient:  dW[n][c] += sum_m G[m][n] * f(X[m][c]),  f = ReLU(BatchNorm) or identity.
// Workgroup = 4 waves, 128 (n) x 16*NT (c) outputs over a range of 32-pixel slabs; wave w owns rows 32w .. 32w+31 of the n tile.
// ---------------------------------------------------------------------------------------------
struct XsWgradArgs {
  const float* G; int ldg; int N;
  int g_bf16;                   // NP == 1 only: G is rows of bf16 (ldg in elements)
  int g_split, x_split;         // NP == 3 only: the operand is given as SPLIT ROWS (xsplit_dev.h: [hi x4 | lo x4] per four values, same addresses): staged verbatim
  const float* X; int ldx; int C;
  const float* x_scale; const float* x_shift;
  float* dW; int ldw;
  int M, split_k;
  unsigned g_bytes, x_bytes;
  int n_ctiles;                 // column tiles; tile t covers channels [ct_c0[t], ct_c0[t] + 16 * ct_nt[t])
  int ct_c0[12], ct_nt[12];
};

constexpr int XS_BM = 128, XS_BK = 32, XS_NTMAX = 12;
constexpr int XS_A_IMG = XS_BK * XS_BM * 2;                 // one plane of the gradient tile: 8 KB
constexpr int XS_B_IMG = XS_BK * XS_NTMAX * 16 * 2;         // one plane of the activation tile: <= 12 KB

template <int NT, int NP>
__device__ __forceinline__ void xs_wgrad1x1_body(const XsWgradArgs& p, unsigned char* smem, int c0, int n0, int s_begin, int s_end) {
  constexpr int BN = NT * 16;
  constexpr int BPATCH = 8 * NT;                            // 4-pixel x 16-channel patches of the activation tile
  constexpr int BL = (BPATCH + 15) / 16;                    // ... per 16-lane group
  unsigned char* const Ahi = smem;
  unsigned char* const Alo = smem + XS_A_IMG;
  unsigned char* const Bhi = smem + 2 * XS_A_IMG;
  unsigned char* const Blo = smem + 2 * XS_A_IMG + XS_B_IMG;
  float* const Ssc = reinterpret_cast<float*>(smem + 2 * XS_A_IMG + 2 * XS_B_IMG);     // [BN] scale | [BN] shift

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int grp = tid >> 4, kq = l16 >> 2, jq = l16 & 3;
  const bool bnrelu = p.x_scale != nullptr;
  const __amdgpu_buffer_rsrc_t srdG = xsrd(p.G, p.g_bytes), srdX = xsrd(p.X, p.x_bytes);

  if (bnrelu) {
    for (int i = tid; i < BN; i += 256) {
      const bool ok = c0 + i < p.C;
      Ssc[i] = ok ? p.x_scale[c0 + i] : 0.f;
      Ssc[BN + i] = ok ? p.x_shift[c0 + i] : 0.f;
    }
  }

  // ---- staging maps.  A: patch (rg = (grp >> 3) + 2 it, ct = grp & 7): pixel row 4 rg + kq, channels 16 ct + 4 jq .. + 3 ----
  unsigned a_voff[4]; int a_row[4]; unsigned a_lds[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rg = (grp >> 3) + 2 * it, ct = grp & 7;
    const int row = 4 * rg + kq, n = n0 + 16 * ct + 4 * jq;
    a_row[it] = n < p.N ? row : 0x40000000;
    a_voff[it] = (NP == 1 && p.g_bf16) ? (unsigned)row * (unsigned)(p.ldg * 2) + (unsigned)(n * 2) : (unsigned)row * (unsigned)(p.ldg * 4) + (unsigned)(n * 4);
    const int rb = row >> 3, pr = (row & 7) ^ ((rb & 1) << 2);
    a_lds[it] = (unsigned)(256 * (rb * 8 + ct) + 32 * pr + 8 * jq);
  }
  unsigned b_voff[BL]; int b_row[BL]; unsigned b_lds[BL]; int b_col[BL];
#pragma unroll
  for (int it = 0; it < BL; ++it) {
    const int pi = grp + 16 * it;
    const int rg = pi / NT, ct = pi - rg * NT;
    const int row = 4 * rg + kq, cl = 16 * ct + 4 * jq;
    const bool ok = pi < BPATCH && c0 + cl < p.C;
    b_row[it] = ok ? row : 0x40000000;
    b_col[it] = cl;
    b_voff[it] = (unsigned)row * (unsigned)(p.ldx * 4) + (unsigned)((c0 + cl) * 4);
    const int rb = row >> 3, pr = (row & 7) ^ ((rb & 1) << 2);
    b_lds[it] = pi < BPATCH ? (unsigned)(256 * (rb * NT + ct) + 32 * pr + 8 * jq) : 0u;
  }

  f32x4 ra[4], rb[BL];
  float bhi[BL];                                              // ReLU upper bound: +inf for a live element, 0 for a dead one (zeroes it after BatchNorm)
  auto load_slab = [&](int s) {
    const int m0 = s * XS_BK;
    const unsigned sog = (unsigned)m0 * (unsigned)(p.ldg * ((NP == 1 && p.g_bf16) ? 2 : 4)), sox = (unsigned)m0 * (unsigned)(p.ldx * 4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const bool ok = m0 + a_row[it] < p.M;
      if (NP == 1 && p.g_bf16) {
        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(srdG, (int)(ok ? a_voff[it] + sog : XOOB), 0, 0);
        ra[it] = f32x4{__uint_as_float(h[0]), __uint_as_float(h[1]), 0.f, 0.f};
      } else {
        ra[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdG, (int)(ok ? a_voff[it] + sog : XOOB), 0, 0));
      }
    }
#pragma unroll
    for (int it = 0; it < BL; ++it) {
      const bool ok = m0 + b_row[it] < p.M;
      rb[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (int)(ok ? b_voff[it] + sox : XOOB), 0, 0));
      bhi[it] = ok ? __builtin_inff() : 0.f;
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      u32x2 hi, lo;
      if (NP == 1 && p.g_bf16) hi = u32x2{__float_as_uint(ra[it][0]), __float_as_uint(ra[it][1])};
      else if (NP == 3 && p.g_split) { hi = u32x2{__float_as_uint(ra[it][0]), __float_as_uint(ra[it][1])}; lo = u32x2{__float_as_uint(ra[it][2]), __float_as_uint(ra[it][3])}; }
      else split4(ra[it][0], ra[it][1], ra[it][2], ra[it][3], hi, lo);
      *reinterpret_cast<u32x2*>(Ahi + a_lds[it]) = hi;
      if (NP == 3) *reinterpret_cast<u32x2*>(Alo + a_lds[it]) = lo;
    }
#pragma unroll
    for (int it = 0; it < BL; ++it) {
      if (BPATCH % 16 != 0 && it == BL - 1 && grp + 16 * it >= BPATCH) continue;
      f32x4 v = rb[it];
      u32x2 hi, lo;
      if (NP == 3 && p.x_split) {                               // already activated and split by the producer (rows past M were loaded as zeros)
        hi = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}; lo = u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])};
      } else {
        if (bnrelu) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(Ssc + b_col[it]), sh = *reinterpret_cast<const f32x4*>(Ssc + BN + b_col[it]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(v[e], sc[e], sh[e]), 0.f, bhi[it]);
        }
        split4(v[0], v[1], v[2], v[3], hi, lo);
      }
      *reinterpret_cast<u32x2*>(Bhi + b_lds[it]) = hi;
      if (NP == 3) *reinterpret_cast<u32x2*>(Blo + b_lds[it]) = lo;
    }
  };

  // ---- fragment addresses: k group g reads pixel rows 8g .. 8g+7 as two 4-row blocks; lane (q, p) of a 16-lane group supplies row q,
  // 8 bytes at channel 4p of the tile ----
  const int fq = l16 >> 2, fp = l16 & 3;
  const unsigned frA0 = (unsigned)(256 * (g * 8) + 32 * ((fq) ^ ((g & 1) << 2)) + 8 * fp);
  const unsigned frA1 = (unsigned)(256 * (g * 8) + 32 * ((4 + fq) ^ ((g & 1) << 2)) + 8 * fp);
  const unsigned frB0 = (unsigned)(256 * (g * NT) + 32 * ((fq) ^ ((g & 1) << 2)) + 8 * fp);
  const unsigned frB1 = (unsigned)(256 * (g * NT) + 32 * ((4 + fq) ^ ((g & 1) << 2)) + 8 * fp);

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_slab(s_begin);
  __syncthreads();                                            // the BatchNorm table is in place
  store_slab();
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const bool more = s + 1 < s_end;
    if (more) load_slab(s + 1);                               // in flight under this slab's MFMAs
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int imm = 256 * (wave * 2 + i);
      ah[i] = tr_frag(Ahi + frA0, Ahi + frA1, imm);
      al[i] = tr_frag(Alo + frA0, Alo + frA1, imm);
    }
    bf16x8 bh[2], bl[2];
    bh[0] = tr_frag(Bhi + frB0, Bhi + frB1, 0);
    bl[0] = tr_frag(Blo + frB0, Blo + frB1, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int cur = t & 1;
      if (t + 1 < NT) {
        bh[cur ^ 1] = tr_frag(Bhi + frB0, Bhi + frB1, 256 * (t + 1));
        bl[cur ^ 1] = tr_frag(Blo + frB0, Blo + frB1, 256 * (t + 1));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (NP == 3) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[cur], acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[cur], acc[i][t], 0, 0, 0);
        }
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[cur], acc[i][t], 0, 0, 0);
      }
    }
    __syncthreads();                                          // every wave is past its reads of the images
    if (more) {
      store_slab();
      __syncthreads();
    }
  }

  // ---- split-K epilogue: f32 atomics into the (pre-zeroed) gradient; D row = 4 g + r (n), column = l16 (c) ----
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = c0 + 16 * t + l16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wave * 32 + i * 16 + g * 4 + r;
        if (c < p.C && n < p.N) atomicAdd(p.dW + (long)n * p.ldw + c, acc[i][t][r]);
      }
  }
}

// (a narrow instantiation - column tiles of <= 96 channels, 162 registers, three workgroups per CU - was measured against this one: 0.17 vs 0.19 ms
// at dense_e2 C = 96 and 0.055 vs 0.071 at dense_e3 C = 192, but 0.29 vs 0.24 at C = 144 and 0.57 vs 0.52 at C = 336: occupancy is not what
// holds the kernel; per 32-pixel slab its four waves read 112 KB of fragments from LDS for 1 152 MFMA cycles each - the LDS is as busy as the pipe)
template <int NP>
__global__ __launch_bounds__(256, 2) void xs_wgrad1x1_kernel(XsWgradArgs p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * XS_A_IMG + 2 * XS_B_IMG + 2 * XS_NTMAX * 16 * 4];
  // work item = (column tile, row tile, K split), column tile fastest; XCD x works through a contiguous range of items so that the
  // column tiles sharing one gradient tile meet in one L2 (placement affects speed only)
  const unsigned total = gridDim.x, L = blockIdx.x, x = L & 7u, q = total >> 3, r = total & 7u;
  const long item = x * q + (x < r ? x : r) + (L >> 3);
  const int ntiles = (p.N + XS_BM - 1) / XS_BM;
  const int ct = (int)(item % p.n_ctiles);
  const long t2 = item / p.n_ctiles;
  const int nt = (int)(t2 % ntiles), split = (int)(t2 / ntiles);
  const int nslab = (p.M + XS_BK - 1) / XS_BK;
  const int per = (nslab + p.split_k - 1) / p.split_k;
  const int s_begin = split * per, s_end = min(nslab, s_begin + per);
  if (s_begin >= s_end) return;
  const int c0 = p.ct_c0[ct], n0 = nt * XS_BM;
  switch (p.ct_nt[ct]) {
    case 6: xs_wgrad1x1_body<6, NP>(p, smem, c0, n0, s_begin, s_end); break;
    case 9: xs_wgrad1x1_body<9, NP>(p, smem, c0, n0, s_begin, s_end); break;
    default: xs_wgrad1x1_body<12, NP>(p, smem, c0, n0, s_begin, s_end); break;
  }
}


// =============================================================================================
// 3x3 / stride 1 / pad 1 INPUT gradient of the dense layers' conv2 (48 gradient channels -> Cb = bn_size * growth bottleneck channels):
//     dZ[m][c] = sum_{tap, n} go[pix(m, tap)][n] * W[tap][n][c],   pix(m, (r, q)) = (y + 1 - r, x + 1 - q),
// optionally gated by the ReLU of the forward value (fma(Y[m][c], xs[c], xt[c]) > 0) with the two BatchNorm-backward sums of the gated
// result (sum dz, sum dz * Y) accumulated per channel - the EPI_MASK_STATS epilogue of igemm.hip.
// GEMM view: 432 = 9 x 48 contracted values per output, i.e. 13.5 MFMA k-steps of 32; a lane's 8-value chunk of a k-step never straddles
// a tap (48 = 6 x 8), so chunk ch = 4 j + g of step j belongs to tap ch / 6, channels 8 (ch % 6) ..
//  * a workgroup owns 192 consecutive pixels.  Their gradient tile (+ W + 1 pixels on either side, 48 channels) is split ONCE into two
//    bf16 images [slot][48] in LDS and serves every output channel the workgroup computes: a fragment is one ds_read_b128 at (slot of
//    the lane's pixel + the tap's shift); a tap that leaves the image reads slot 0, which holds zeros.  With 96-byte slots the 16
//    lanes of a ds_read_b128 group (8 consecutive pixels of k-group g, 8 of g + 1: 16 bytes further) cover all 64 banks.
//  * the weights arrive pre-split in FRAGMENT order (k_xs_pack_w3_dgrad: [16-channel tile][k-step][hi | lo][lane][8 bf16], one pass per
//    layer and step over 4.7 MB); a wave owns 32 of the 128 output channels of a column tile and takes ITS fragments straight from
//    global memory (L2) into registers, one k-step ahead - no LDS, no DMA, and therefore NO barrier in the main loop.
//  * MFMA A operand = weights (rows = output channels), B operand = gradient (columns = pixels): a lane then holds 4 CONSECUTIVE
//    channels of one pixel - the gate's Y read and the dZ store are 16-byte accesses, two of them side by side cover a pixel's 128-byte
//    line - and every wave owns its 32 channels for ALL 192 pixels: the BatchNorm-backward sums need no cross-wave reduction.
//  * a work item = (pixel tile, group of column tiles): the image staging and the address set-up are paid once per group.
// =============================================================================================
constexpr int XD_BM = 192, XD_MT = XD_BM / 16, XD_KSTEPS = 14, XD_SLOT = 96, XD_BN = 128, XD_EB = 6;

struct XsDgrad3Args {
  const float* G; int ldg;
  const unsigned char* Wf;
  float* out; int ldc;
  const float* X; int ldx; const float* x_scale; const float* x_shift;
  double* stat0; double* stat1;
  int B, H, W, M, Cb;
  int out_bf16;                            // NP == 1 only: `out` is rows of bf16 (ldc in elements): the consumers round to bf16 anyway
  int nslots, plane_bytes;                 // slot 0 = zeros, slot 1 + s = pixel m0 - (W + 1) + s, s < XD_BM + 2 (W + 1)
  int mtiles, ctiles;                      // pixel tiles, 128-channel column tiles
  int abl;                                 // development builds only (RDM_XD3_ABL): timing-only ablations, results wrong by construction: 1 no gate loads, 2 no stores, 4 no MFMA loop, 8 no statistics (profiles/r05_dgrad3x3_ablation.txt)
  unsigned g_bytes, w_bytes, x_bytes, o_bytes;
};
#ifdef RDM_DEV_VARIANTS
#define XD3_ABL(bit) (p.abl & (bit))
#else
#define XD3_ABL(bit) false
#endif

__global__ __launch_bounds__(256) void k_xs_pack_w3_dgrad(const float* __restrict__ w, long wtap, int ldw, int Cb, unsigned char* __restrict__ Wf) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(idx & 63);
  const long r1 = idx >> 6;
  const int j = (int)(r1 % XD_KSTEPS), ct = (int)(r1 / XD_KSTEPS);
  if (ct >= Cb / 16) return;
  const int c = ct * 16 + (lane & 15), g = lane >> 4;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * j + 8 * g + e, tap = k / 48, n = k - tap * 48;
    v[e] = k < 432 ? w[(long)tap * wtap + (long)n * ldw + c] : 0.f;
  }
  u32x2 h0, l0, h1, l1;
  split4(v[0], v[1], v[2], v[3], h0, l0);
  split4(v[4], v[5], v[6], v[7], h1, l1);
  unsigned char* dst = Wf + ((((long)ct * XD_KSTEPS + j) * 2) * 64 + lane) * 16;
  *reinterpret_cast<u32x4*>(dst) = u32x4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<u32x4*>(dst + 1024) = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

template <bool MASK, int NP>
__global__ __launch_bounds__(256, 2) void xs_dgrad3x3_kernel(XsDgrad3Args p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char xs_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int W = p.W, HW = p.H * p.W;
  // work items = (pixel tile, 128-channel column tile), column tile fastest; workgroup b takes the contiguous range [b T / G, (b + 1) T / G)
  // of them (G = gridDim.x = every workgroup slot of the chip: all resident at once, balanced to +-1 item) and re-stages the gradient
  // images only when its pixel tile changes
  const int T = p.mtiles * p.ctiles;
  const int it0 = (int)((long)blockIdx.x * T / gridDim.x), it1 = (int)((long)(blockIdx.x + 1) * T / gridDim.x);
  unsigned char* const Ahi = xs_smem;
  unsigned char* const Alo = xs_smem + p.plane_bytes;
  const __amdgpu_buffer_rsrc_t srdG = xsrd(p.G, p.g_bytes), srdW = xsrd(p.Wf, p.w_bytes), srdX = xsrd(p.X, p.x_bytes);

  // k-step j -> the lane's chunk: offset of (tap shift, 8-channel group), biased by +32768, two per register
  unsigned dpk[XD_KSTEPS / 2];
  const int hi2 = g >> 1;
#pragma unroll
  for (int j = 0; j < XD_KSTEPS; ++j) {
    const int ch = 4 * j + g, tap = ch < 54 ? ch / 6 : 0, r = tap / 3, q = tap - 3 * r;
    const unsigned d = (unsigned)(((1 - r) * W + (1 - q)) * XD_SLOT + (ch - 6 * (ch / 6)) * 16 + 32768);
    if (j & 1) dpk[j >> 1] |= d << 16; else dpk[j >> 1] = d;
  }
  int base0 = 0; int vm[XD_MT / 3];                          // pixel tile i: image offset base0 + 16 * 96 * i; 9 validity bits at bit 10 * (i % 3) of vm[i / 3]
  int cur_mt = -1, m0 = 0;
  const unsigned wv = (unsigned)(lane * 16);
  for (int it = it0; it < it1; ++it) {
    const int mt = it / p.ctiles, tile = it - mt * p.ctiles;
    if (mt != cur_mt) {
      cur_mt = mt;
      m0 = mt * XD_BM;
      __syncthreads();                                          // every wave is past its reads of the previous images
      // ---- gradient tile -> the two bf16 images ----
      const int total4 = (p.nslots - 1) * 12, pbase = m0 - (W + 1);
      for (int b0 = 0; b0 < total4; b0 += 256 * 6) {
        f32x4 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int idx = b0 + u * 256 + tid, slot = idx / 12, quad = idx - slot * 12;
          const int pix = pbase + slot;
          const bool ok = idx < total4 && pix >= 0 && pix < p.M;
          v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdG, (int)(ok ? (unsigned)pix * (unsigned)(p.ldg * 4) + (unsigned)(quad * 16) : XOOB), 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int idx = b0 + u * 256 + tid, slot = idx / 12, quad = idx - slot * 12;
          if (idx < total4) {
            u32x2 hi, lo;
            split4(v[u][0], v[u][1], v[u][2], v[u][3], hi, lo);
            *reinterpret_cast<u32x2*>(Ahi + (slot + 1) * XD_SLOT + quad * 8) = hi;
            if (NP == 3) *reinterpret_cast<u32x2*>(Alo + (slot + 1) * XD_SLOT + quad * 8) = lo;
          }
        }
      }
      if (tid < 24) *reinterpret_cast<u32x2*>(xs_smem + (tid % 12) * 8 + (tid / 12) * p.plane_bytes) = u32x2{0u, 0u};      // slot 0 of both planes
      // ---- per lane: pixel tile i -> image offset of the pixel's slot (biased by -32768) and the validity of its 9 taps ----
      base0 = (l16 + W + 1 + 1) * XD_SLOT - 32768;
#pragma unroll
      for (int i = 0; i < XD_MT; ++i) {
        const int m = m0 + i * 16 + l16;
        const bool ok = m < p.M;
        const int b = m / HW, rem = m - b * HW, y = rem / W, x = rem - y * W;
        int v = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int r = tap / 3, q = tap - 3 * r;
          const bool valid = ok && (unsigned)(y + 1 - r) < (unsigned)p.H && (unsigned)(x + 1 - q) < (unsigned)W;
          v |= valid ? (1 << tap) : 0;
        }
        if (i % 3 == 0) vm[i / 3] = v; else vm[i / 3] |= v << (10 * (i % 3));      // bit 9 of each field (the zero half-step 432 .. 447) stays clear
      }
      __syncthreads();                                          // the images are complete
    }
    const int ct16 = tile * (XD_BN / 16) + 2 * wave;            // this wave's two 16-channel tiles
    if (ct16 * 16 >= p.Cb) continue;                            // (wave-uniform) ragged last column tile
    asm volatile("" : "+v"(base0));                             // (keeps the 168 fragment addresses from being hoisted out of the item loop into registers)
    f32x4 acc[XD_MT][2];
#pragma unroll
    for (int i = 0; i < XD_MT; ++i) { acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // weight fragments of k-step j: [t][hi | lo], 1 KiB each, one coalesced 16-byte load per lane (a tile past Cb reads as zeros)
    bf16x8 wf[2][2][2];
    auto load_w = [&](int j, bf16x8 (&w)[2][2]) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < (NP == 3 ? 2 : 1); ++pl)
          w[t][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srdW, (int)(wv + (unsigned)((ct16 + t) * (XD_KSTEPS * 2048))), j * 2048 + pl * 1024, 0));
    };
    // (the hardware range check looks at the VECTOR offset: the tile index travels there, so a tile past Cb reads as zeros)
    const bool t1ok = (ct16 + 1) * 16 < p.Cb;
    load_w(0, wf[0]);
    // gradient fragments of (k-step j, pixel tile i): software pipeline - the pair of ds_read_b128 of tile i + 1 is issued BEFORE the six
    // MFMAs of tile i (pinned: left alone, hipcc sinks every read next to its first use and the matrix pipe waits for LDS 12 times a step)
    bf16x8 gq[2][2];
    auto frag = [&](int j, int i, bf16x8 (&f)[2]) {
      const int dj = (int)((dpk[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu);
      const int tapj = (4 * j) / 6 + (((4 * j) % 6 == 4) ? hi2 : 0);      // == (4 j + g) / 6; 9 in the zero half-step -> never valid
      const int a = (base0 + i * (16 * XD_SLOT) + dj) & __builtin_amdgcn_sbfe(vm[i / 3], (unsigned)(tapj + 10 * (i % 3)), 1u);
      f[0] = *reinterpret_cast<const bf16x8*>(Ahi + a);
      if (NP == 3) f[1] = *reinterpret_cast<const bf16x8*>(Alo + a);
    };
    frag(0, 0, gq[0]);
#pragma unroll
    for (int j = 0; j < XD_KSTEPS; ++j) {
      if (XD3_ABL(4)) break;
      if (j + 1 < XD_KSTEPS) load_w(j + 1, wf[(j + 1) & 1]);
#pragma unroll
      for (int i = 0; i < XD_MT; ++i) {
        const int cur = (j * XD_MT + i) & 1;
        if (i + 1 < XD_MT) frag(j, i + 1, gq[cur ^ 1]);
        else if (j + 1 < XD_KSTEPS) frag(j + 1, 0, gq[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (NP == 3) {
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j & 1][t][1], gq[cur][0], acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j & 1][t][0], gq[cur][1], acc[i][t], 0, 0, 0);
          }
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j & 1][t][0], gq[cur][0], acc[i][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- epilogue: D row = 4 g + r (channel), column = l16 (pixel): a lane owns channels c4 .. c4 + 3 of pixel m ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c4 = (ct16 + t) * 16 + 4 * g;
      const bool cok = t == 0 || t1ok;
      f32x4 xs = {0.f, 0.f, 0.f, 0.f}, xt = {0.f, 0.f, 0.f, 0.f};
      if (MASK && cok) { xs = *reinterpret_cast<const f32x4*>(p.x_scale + c4); xt = *reinterpret_cast<const f32x4*>(p.x_shift + c4); }
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      // (gate loads in buffer form: ONE 32-bit offset register, the pixel tile's row offset travels in the scalar offset; a pixel or channel
      // tile past the end becomes an out-of-range vector offset and reads as zeros)
      const unsigned vx = (unsigned)(m0 + l16) * (unsigned)(p.ldx * 4) + (unsigned)(c4 * 4);
#pragma unroll
      for (int i0 = 0; i0 < XD_MT; i0 += XD_EB) {
        f32x4 xv[XD_EB];
        if (MASK && XD3_ABL(1)) {
#pragma unroll
          for (int u = 0; u < XD_EB; ++u) xv[u] = acc[i0 + u][t];
        } else if (MASK) {
#pragma unroll
          for (int u = 0; u < XD_EB; ++u)
            xv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (int)((cok && m0 + (i0 + u) * 16 + l16 < p.M) ? vx : XOOB), (i0 + u) * 16 * p.ldx * 4, 0));
        }
#pragma unroll
        for (int u = 0; u < XD_EB; ++u) {
          const int i = i0 + u;
          const bool ok = cok && m0 + i * 16 + l16 < p.M;
          f32x4 v = acc[i][t];
          if (MASK) {
            const f32x4 x = xv[u];                                 // zeros where !ok: nothing of a dead element reaches the sums
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(x[e], xs[e], xt[e]) > 0.f ? v[e] : 0.f;
            if (ok) { s0 += v; s1 += v * x; }
          }
          // (plain global stores ON PURPOSE: with buffer-form stores here, the shuffles and the exec-masked atomics of the statistics below
          // left wrong values in lanes 12-15 of some stored registers on MI355X / ROCm 7.2 - measured, cause not established; the same
          // epilogue with global_store_dwordx4 is exact)
          if (NP == 1 && p.out_bf16) {
            const bf16x2 lo2 = {(__bf16)v[0], (__bf16)v[1]}, hi2 = {(__bf16)v[2], (__bf16)v[3]};
            if (ok) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + (long)(m0 + i * 16 + l16) * p.ldc + c4) = u32x2{__builtin_bit_cast(unsigned, lo2), __builtin_bit_cast(unsigned, hi2)};
          } else if (ok && !XD3_ABL(2)) *reinterpret_cast<f32x4*>(p.out + (long)(m0 + i * 16 + l16) * p.ldc + c4) = v;
          else if (XD3_ABL(2)) asm volatile("" :: "v"(v));
        }
      }
      if (MASK && !XD3_ABL(8)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = s0[e], b = s1[e];
          a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
          a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
          a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
          a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
          if (l16 == e) { s0[0] = a; s1[0] = b; }               // lane e of each 16 keeps channel c4 + e
        }
        if (l16 < 4 && cok) {
          atomicAdd(p.stat0 + c4 + l16, (double)s0[0]);
          atomicAdd(p.stat1 + c4 + l16, (double)s1[0]);
        }
      }
    }
  }
}


// =============================================================================================
// 1x1 INPUT gradient of the dense layers' conv1:  dX[m][c] = sum_n dY[m][n] * W[n][c]  (n: the Cb bottleneck channels, contiguous in dY's
// rows; c: the layer's input channels), optionally gated by the ReLU of the forward value with the BatchNorm-backward sums (EPI_MASK_STATS).
//  * both operands live in LDS per 32-deep k-step as [row][32 k] bf16 images (64 B per row and plane, 16-byte chunks XOR-swizzled by
//    ((row >> 2) & 1) << 1: the ds_read_b128 lane groups then cover all 64 banks): rows = pixels for the gradient, rows = output
//    channels for the weights.  The gradient slab is staged global -> registers -> split -> LDS (next slab in flight under the MFMAs);
//    the weights come pre-split AND pre-transposed ([plane][k-step][c][32 k], k_xs_pack_w1_dgrad) and are copied verbatim.
//  * one 8-wave workgroup per CU: 2 wave columns x 4 wave rows; a wave owns <= 6 sixteen-channel tiles x <= 5 sixteen-pixel tiles.
//    The tile height (16 .. 320 pixels) is chosen by the host so that the grid fills whole rounds of the 256 CUs.
//  * MFMA A operand = weights (rows = channels), B = gradient (columns = pixels): 16-byte gate loads / stores as in the 3x3 kernel.
// =============================================================================================
constexpr int X1_MTW_BIG = 5, X1_MTW_SMALL = 2, X1_NTW = 6, X1_BNMAX = 2 * X1_NTW * 16;
constexpr int X1_W_IMG = X1_BNMAX * 64;                                      // one plane of one stage
constexpr int x1_lds(int mtw) { return 2 * (2 * (4 * mtw * 16 * 64) + 2 * X1_W_IMG); }

struct XsDgrad1Args {
  const float* G; int ldg; int K;          // dY [M][ldg], K = Cb contracted channels
  int g_bf16;                              // NP == 1 only: dY is rows of bf16 (ldg in elements): staged verbatim, no conversion
  int g_split;                             // NP == 3 only: dY is SPLIT ROWS (xsplit_dev.h): staged verbatim, no conversion
  int acc;                                 // MASK only: out += x_scale * (gated dz) instead of out = gated dz (deferred norm1 backward: out = the block gradient)
  const unsigned char* Wp;                 // [plane][ksteps][C][64 B]
  float* out; int ldc;
  const float* X; int ldx; const float* x_scale; const float* x_shift;
  double* stat0; double* stat1;
  int M, C, ksteps;
  int PT;                                  // 16-pixel tiles per workgroup
  int mtiles, ctiles;                      // pixel tiles x column tiles (<= 12 sixteen-channel tiles each)
  int ct_c0[12], ct_n[12];                 // column tile t: first 16-channel tile, number of 16-channel tiles
  unsigned g_bytes, w_bytes, x_bytes;
};

__global__ __launch_bounds__(256) void k_xs_pack_w1_dgrad(const float* __restrict__ w, int ldw, int K, int C, int ksteps, unsigned char* __restrict__ Wp) {
  // one thread per (k-step, c, 8-k chunk): reads 8 weights of column c (stride ldw), writes 16 B of each plane at the swizzled chunk
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(idx % C);
  const long r1 = idx / C;
  const int ch = (int)(r1 & 3), j = (int)(r1 >> 2);
  if (j >= ksteps) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * j + 8 * ch + e;
    v[e] = k < K ? w[(long)k * ldw + c] : 0.f;
  }
  u32x2 h0, l0, h1, l1;
  split4(v[0], v[1], v[2], v[3], h0, l0);
  split4(v[4], v[5], v[6], v[7], h1, l1);
  const int chs = ch ^ (((c >> 2) & 1) << 1);
  const long plane = (long)ksteps * C * 64;
  unsigned char* dst = Wp + ((long)j * C + c) * 64 + chs * 16;
  *reinterpret_cast<u32x4*>(dst) = u32x4{h0[0], h0[1], h1[0], h1[1]};
  *reinterpret_cast<u32x4*>(dst + plane) = u32x4{l0[0], l0[1], l1[0], l1[1]};
}

// MTW = sixteen-pixel tiles per wave row (4 wave rows: tile heights up to 64 MTW pixels), D = k-steps the global loads run ahead of their LDS
// stores.  (5, 1): the many-pixel blocks.  (2, 1): few pixels (dense_e4's 4 560: 128-pixel tiles; 164 instead of 230 registers).  D = 3 on the
// small tile was built to cover the load latency of a 0.24 us k-step (36 MFMAs) and measured: with an exit per unrolled step hipcc's waitcnt
// insertion drained the queue at the head of the body (s_waitcnt vmcnt(3)); with the steps padded to a multiple of D (one exit) the waits are
// the intended vmcnt(13-14) - and the kernel is NO faster (dense_e4, C = 432 / 1248 / 2064: 44 / 53 / 90 us against 43 / 53 / 63 at D = 1).
// Latency is not what it waits for: at C = 1248 the 252 workgroups move 227 MB of weight and gradient slabs through the L2 in 40 us (5.7 TB/s)
// - every one of the 36 pixel tiles re-reads the layer's packed weights.
template <bool MASK, int NP, int MTW, int D>
__global__ __launch_bounds__(512, 2) void xs_dgrad1x1_kernel(XsDgrad1Args p) {
  constexpr int X1_MTW = MTW, X1_G_IMG = 4 * MTW * 16 * 64, X1_STAGE = 2 * X1_G_IMG + 2 * X1_W_IMG;
  extern __shared__ __attribute__((aligned(1024))) unsigned char x1_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int wc = wave & 1, wp = wave >> 1;
  // work item = (pixel tile, column tile), column tile fastest (the workgroups of one pixel tile run side by side: its gradient rows
  // are fetched from HBM once); XCD x works through a contiguous range of items
  const unsigned total = gridDim.x, L = blockIdx.x, xq = L & 7u, qq = total >> 3, rr = total & 7u;
  const unsigned item = xq * qq + (xq < rr ? xq : rr) + (L >> 3);
  const int mt = (int)(item / (unsigned)p.ctiles), ctile = (int)(item - (unsigned)mt * (unsigned)p.ctiles);
  const int BM = p.PT * 16, m0 = mt * BM;
  const int nct = p.ct_n[ctile], ct0 = p.ct_c0[ctile], BN = nct * 16;
  // this wave: 16-channel tiles [tw0, tw0 + ntw) of the column tile, 16-pixel tiles [pw0, pw0 + npw) of the pixel tile
  const int ntw_max = (nct + 1) >> 1, tw0 = wc * ntw_max, ntw = min(ntw_max, nct - tw0);
  const int npw_max = (p.PT + 3) >> 2, pw0 = wp * npw_max, npw = max(0, min(npw_max, p.PT - pw0));
  const __amdgpu_buffer_rsrc_t srdG = xsrd(p.G, p.g_bytes), srdW = xsrd(p.Wp, p.w_bytes), srdX = xsrd(p.X, p.x_bytes);

  // ---- staging maps.  Gradient: float4 (row = idx >> 3, 4 k at 4 (idx & 7)) -> 8 B of each plane; weights: 16-byte pieces, verbatim ----
  unsigned g_voff[X1_MTW]; unsigned g_lds[X1_MTW]; int g_k[X1_MTW];
#pragma unroll
  for (int u = 0; u < X1_MTW; ++u) {
    const int idx = tid + 512 * u, row = idx >> 3, f4 = idx & 7;
    const bool ok = row < BM && m0 + row < p.M;
    g_k[u] = ok ? 4 * f4 : 0x40000000;
    g_voff[u] = (NP == 1 && p.g_bf16) ? (unsigned)(m0 + row) * (unsigned)(p.ldg * 2) + (unsigned)(f4 * 8)
                                      : (unsigned)(m0 + row) * (unsigned)(p.ldg * 4) + (unsigned)(f4 * 16);
    g_lds[u] = (unsigned)(row * 64 + (((f4 >> 1) ^ (((row >> 2) & 1) << 1)) * 16) + (f4 & 1) * 8);
  }
  const int wpieces = BN * 8;                                  // 16-byte pieces of the weight slab: both planes
  const unsigned w_plane = (unsigned)p.ksteps * (unsigned)p.C * 64u;
  f32x4 rg[D][X1_MTW]; u32x4 rw[D][3];                          // register set (k-step) % D
  auto load_slab = [&](int set, int j) {                        // a k-step past the end reads as zeros (k >= K; weight offsets past the packed buffer)
#pragma unroll
    for (int u = 0; u < X1_MTW; ++u) {
      if (NP == 1 && p.g_bf16) {                                // 4 bf16 = 8 bytes, kept in the first two registers of the set
        const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(srdG, (int)((32 * j + g_k[u] < p.K) ? g_voff[u] + (unsigned)(j * 64) : XOOB), 0, 0);
        rg[set][u] = f32x4{__uint_as_float(h[0]), __uint_as_float(h[1]), 0.f, 0.f};
      } else {
        rg[set][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdG, (int)((32 * j + g_k[u] < p.K) ? g_voff[u] + (unsigned)(j * 128) : XOOB), 0, 0));
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = tid + 512 * u, pl = q >= BN * 4, r = pl ? q - BN * 4 : q;      // piece r of plane pl: row r >> 2, chunk r & 3 (already swizzled in memory)
      rw[set][u] = __builtin_amdgcn_raw_buffer_load_b128(srdW, (int)((q < wpieces && j < p.ksteps) ? (unsigned)pl * w_plane + ((unsigned)j * (unsigned)p.C + (unsigned)(ct0 * 16)) * 64u + (unsigned)r * 16u : XOOB), 0, 0);
    }
  };
  auto store_slab = [&](int set, unsigned char* st) {
#pragma unroll
    for (int u = 0; u < X1_MTW; ++u) {
      if ((tid + 512 * u) >> 3 < BM) {
        u32x2 hi, lo;
        if (NP == 1 && p.g_bf16) hi = u32x2{__float_as_uint(rg[set][u][0]), __float_as_uint(rg[set][u][1])};
        else if (NP == 3 && p.g_split) { hi = u32x2{__float_as_uint(rg[set][u][0]), __float_as_uint(rg[set][u][1])}; lo = u32x2{__float_as_uint(rg[set][u][2]), __float_as_uint(rg[set][u][3])}; }
        else split4(rg[set][u][0], rg[set][u][1], rg[set][u][2], rg[set][u][3], hi, lo);
        *reinterpret_cast<u32x2*>(st + g_lds[u]) = hi;
        if (NP == 3) *reinterpret_cast<u32x2*>(st + X1_G_IMG + g_lds[u]) = lo;
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = tid + 512 * u, pl = q >= BN * 4, r = pl ? q - BN * 4 : q;
      if (q < wpieces) *reinterpret_cast<u32x4*>(st + 2 * X1_G_IMG + pl * X1_W_IMG + r * 16) = rw[set][u];
    }
  };

  f32x4 acc[X1_MTW][X1_NTW];
#pragma unroll
  for (int i = 0; i < X1_MTW; ++i)
#pragma unroll
    for (int t = 0; t < X1_NTW; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned fr = (unsigned)(l16 * 64 + ((g ^ (((l16 >> 2) & 1) << 1)) * 16));      // fragment offset inside a 16-row tile of either image

  load_slab(0, 0);
  store_slab(0, x1_smem);
#pragma unroll
  for (int d = 1; d < D; ++d) load_slab(d, d);                 // k-steps 1 .. D - 1 in flight
  __syncthreads();
  // (the k-steps are padded to a multiple of D with all-zero steps - both operands read as zeros past the end - so that the unrolled body has ONE
  // exit: with an exit per unrolled step hipcc's waitcnt insertion drained the load queue at the head of the body)
  const int ksteps_pad = (p.ksteps + D - 1) / D * D;
  for (int j0 = 0; j0 < ksteps_pad; j0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int j = j0 + u;
      const bool more = j + 1 < ksteps_pad;
      const unsigned char* const st = x1_smem + (j & 1) * X1_STAGE;
      load_slab(u, j + D);                                      // in flight under D k-steps of MFMAs (set u held k-step j: stored one step ago)
      bf16x8 gh[X1_MTW], gl[X1_MTW];
#pragma unroll
      for (int i = 0; i < X1_MTW; ++i)
        if (i < npw) {
          gh[i] = *reinterpret_cast<const bf16x8*>(st + (pw0 + i) * 1024 + fr);
          gl[i] = *reinterpret_cast<const bf16x8*>(st + X1_G_IMG + (pw0 + i) * 1024 + fr);
        }
#pragma unroll
      for (int t = 0; t < X1_NTW; ++t)
        if (t < ntw) {
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(st + 2 * X1_G_IMG + (tw0 + t) * 1024 + fr);
          const bf16x8 wl = *reinterpret_cast<const bf16x8*>(st + 2 * X1_G_IMG + X1_W_IMG + (tw0 + t) * 1024 + fr);
#pragma unroll
          for (int i = 0; i < X1_MTW; ++i)
            if (i < npw) {
              if (NP == 3) {
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, gh[i], acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gl[i], acc[i][t], 0, 0, 0);
              }
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh[i], acc[i][t], 0, 0, 0);
            }
        }
      if (more) store_slab((u + 1) % D, x1_smem + ((j + 1) & 1) * X1_STAGE);   // the other stage: every wave left it at the previous barrier
      __syncthreads();
    }
  }

  // ---- epilogue: a lane owns channels c4 .. c4 + 3 (D rows 4 g + r) of pixel m (D column l16) ----
#pragma unroll
  for (int t = 0; t < X1_NTW; ++t) {
    if (t >= ntw) continue;
    const int c4 = (ct0 + tw0 + t) * 16 + 4 * g;
    const bool cok = c4 < p.C;
    f32x4 xs = {0.f, 0.f, 0.f, 0.f}, xt = {0.f, 0.f, 0.f, 0.f};
    if (MASK && cok) { xs = *reinterpret_cast<const f32x4*>(p.x_scale + c4); xt = *reinterpret_cast<const f32x4*>(p.x_shift + c4); }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    const unsigned vx = (unsigned)(m0 + pw0 * 16 + l16) * (unsigned)(p.ldx * 4) + (unsigned)(c4 * 4);
    f32x4 xv[X1_MTW];
    if (MASK) {
#pragma unroll
      for (int i = 0; i < X1_MTW; ++i)
        xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (int)((cok && i < npw && m0 + (pw0 + i) * 16 + l16 < p.M) ? vx : XOOB), i * 16 * p.ldx * 4, 0));
    }
#pragma unroll
    for (int i = 0; i < X1_MTW; ++i) {
      const int m = m0 + (pw0 + i) * 16 + l16;
      const bool ok = cok && i < npw && m < p.M;
      f32x4 v = acc[i][t];
      if (MASK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(xv[i][e], xs[e], xt[e]) > 0.f ? v[e] : 0.f;
        if (ok) { s0 += v; s1 += v * xv[i]; }
      }
      if (ok) {                                                                    // (plain global stores: see the note in xs_dgrad3x3_kernel)
        float* const o = p.out + (long)m * p.ldc + c4;
        if (MASK && p.acc) {
          f32x4 gv = *reinterpret_cast<const f32x4*>(o);
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[e] = fmaf(xs[e], v[e], gv[e]);
          *reinterpret_cast<f32x4*>(o) = gv;
        } else {
          *reinterpret_cast<f32x4*>(o) = v;
        }
      }
    }
    if (MASK) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s0[e], b = s1[e];
        a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
        a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
        a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
        a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
        if (l16 == e) { s0[0] = a; s1[0] = b; }
      }
      if (l16 < 4 && cok && npw > 0) {
        atomicAdd(p.stat0 + c4 + l16, (double)s0[0]);
        atomicAdd(p.stat1 + c4 + l16, (double)s1[0]);
      }
    }
  }
}


// =============================================================================================
// 3x3 / stride 1 / pad 1 WEIGHT gradient of the dense layers' conv2 (48 output channels):
//     dW[tap][n][c] += sum_m go[m][n] * f(Y[pix(m, tap)][c]),  pix(m, (r, q)) = (y - 1 + r, x - 1 + q),  f = ReLU(BatchNorm) or identity.
// The contraction runs over PADDED positions u of the (H + 2) x (W + 2) frame of every image: with zeros at the frame's border in BOTH
// operands, every tap is a plain shift  u -> u + (r - 1)(W + 2) + (q - 1)  and no product needs a mask (6 - 12 % more k-steps than pixels).
//  * the activations of the workgroup's 64 channels live in an LDS RING of 256 padded positions ([position][16-channel tile] bf16 images as
//    in the 1x1 kernel, hi and lo plane): every slab of 32 positions is loaded, normalised, split and stored ONCE and then read by the
//    nine taps of the seven slabs around it through ds_read_b64_tr_b16 at shifted rows (a lane supplies its own row's address, so the
//    shifts need no alignment); the 48-channel gradient slab is double-buffered.  ONE barrier per slab.
//  * a wave owns one 16-channel tile: 9 taps x 3 gradient tiles = 27 accumulator tiles (108 registers), 81 MFMAs per slab.
//  * K split over the slabs; partial sums leave with f32 atomics into the (pre-zeroed) packed gradient.
// =============================================================================================
constexpr int XW_D = 2, XW_RING = 256, XW_BC = 64, XW_YPLANE = XW_RING * XW_BC * 2, XW_GPLANE = 32 * 48 * 2;      // 32 KB, 3 KB
constexpr int XW_LDS = 2 * XW_YPLANE + 4 * XW_GPLANE;

struct XsWgrad3Args {
  const float* G; int ldg;                 // output gradient [M][ldg], 48 channels
  const float* Y; int ldy; int C;          // forward input (pre BatchNorm) [M][ldy], C channels
  const float* y_scale; const float* y_shift;
  float* dW; long wtap; int ldw; int N;    // packed gradient [tap][n][c]
  int B, H, W;
  int nslab, split, ahead;                 // slabs of 32 padded positions; K splits; slabs the activation ring runs ahead (= behind): ceil((W + 3) / 32)
  int g_frame;                             // NP == 3: G is the FRAME image of the gradient (launch_frame_split_rows): split rows [padded position][48], zeros on the frame's
                                           // border - a slab's gradient rows are 6 KB of consecutive bytes, staged verbatim (no position arithmetic, no split)
  unsigned g_bytes, y_bytes;
};

template <int NP>
__global__ __launch_bounds__(256, 2) void xs_wgrad3x3_kernel(XsWgrad3Args p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[XW_LDS];
  unsigned char* const Yhi = smem;
  unsigned char* const Ylo = smem + XW_YPLANE;
  unsigned char* const Gb = smem + 2 * XW_YPLANE;            // [buffer][hi | lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int grp = tid >> 4, kq = l16 >> 2, jq = l16 & 3;
  const int Wp = p.W + 2, PP = (p.H + 2) * Wp, U = p.B * PP;
  const int c0 = blockIdx.x * XW_BC;
  const int per = (p.nslab + p.split - 1) / p.split;
  const int s_begin = blockIdx.y * per, s_end = min(p.nslab, s_begin + per);
  if (s_begin >= s_end) return;
  const __amdgpu_buffer_rsrc_t srdG = xsrd(p.G, p.g_bytes), srdY = xsrd(p.Y, p.y_bytes);
  const bool bnrelu = p.y_scale != nullptr;

  // ---- staging maps.  Activations: patch (rg = (grp >> 2) + 4 it, ct = grp & 3): slab row 4 rg + kq, channels c0 + 16 ct + 4 jq .. + 3 ----
  const int yc = c0 + 16 * (grp & 3) + 4 * jq;
  const bool ycok = yc < p.C;
  f32x4 ysc = {1.f, 1.f, 1.f, 1.f}, ysh = {0.f, 0.f, 0.f, 0.f};
  if (bnrelu && ycok) { ysc = *reinterpret_cast<const f32x4*>(p.y_scale + yc); ysh = *reinterpret_cast<const f32x4*>(p.y_shift + yc); }
  // offset of (padded position u, this thread's channel quad) or XOOB for a border / out-of-range position
  // (u < 2^21: floor((u + 0.5) / d) through one float multiply is exact - the quotient's error stays far below the 0.5 / d margin)
  const float rPP = 1.0f / (float)PP, rWp = 1.0f / (float)Wp;
  // (a per-geometry table position -> pixel index in place of this arithmetic - ~20 of the ~45 vector instructions a staged float4 costs - was built and
  // measured: 50.9-51.2 vs 50.0-50.4 ms per step - the table load in front of every data load costs more than the divisions it saves.  Timing-only
  // ablations of this kernel at dense_e2, alone: 0.57 ms whole, 0.42 without its MFMAs, 0.38 without its staging, 0.36 staging alone, 0.355 MFMAs alone,
  // 0.19 with neither (prologue, barriers, gradient fragments, 14 M atomic adds of the 12-way K split): staging and MFMA overlap only by a quarter.
  // The ISA says why: a slab is ~200 address VALU + 4 loads, THEN 81 MFMAs in 9 bursts with their 4 ds_reads, THEN ~190 convert / split VALU + 8 ds_writes
  // - the matrix pipe idles through both VALU stretches; only the other workgroup of the CU fills them.  __builtin_amdgcn_sched_group_barrier
  // pipelines (1 MFMA : 2-5 VALU, with and without DS / VMEM groups) left hipcc's order unchanged; a hand-split of the staging into ~7-instruction
  // pieces behind every 3 MFMAs is what it would take.  Cheaper addresses alone do not help: scalar-stepped frame positions (15 instead of ~45
  // vector instructions per address, built and measured) left the kernel at 0.52 ms and the step at 50.0-50.3 ms)
  auto pix_off = [&](int u, int ld, int col, bool colok) -> unsigned {
    const int b = (int)(((float)u + 0.5f) * rPP), rem = u - b * PP, yp = (int)(((float)rem + 0.5f) * rWp), xp = rem - yp * Wp;
    const bool ok = u >= 0 && u < U && colok && yp >= 1 && yp <= p.H && xp >= 1 && xp <= p.W;
    return ok ? (unsigned)((b * p.H + yp - 1) * p.W + xp - 1) * (unsigned)(ld * 4) + (unsigned)(col * 4) : XOOB;
  };
  auto y_lds = [&](int s, int it) -> unsigned {              // ring position of (slab s, this thread's row of patch it)
    const int row = ((s & 7) << 5) + 4 * ((grp >> 2) + 4 * it) + kq;
    const int rb = row >> 3, pr = (row & 7) ^ ((rb & 1) << 2);
    return (unsigned)(256 * (rb * 4 + (grp & 3)) + 32 * pr + 8 * jq);
  };
  // global loads run XW_D slabs ahead of their LDS stores (a slab's 81 MFMAs last ~0.5 us, a load under the step's traffic 2 - 3 us: with ONE
  // slab of lead - the first form of this kernel - every barrier waited for memory and the matrix pipe ran at 12 % of its peak): XW_D register
  // sets, target slab t in set (t - s_begin) % XW_D, indexed statically by the XW_D-fold unrolled slab loop
  f32x4 ry[XW_D][2]; unsigned yin = 0u;                       // bit 2 set + it: that load was inside the image (ReLU upper bound inf; 0 zeroes a border position after BatchNorm)
  auto load_y = [&](int set, int s) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const unsigned off = pix_off(s * 32 + 4 * ((grp >> 2) + 4 * it) + kq, p.ldy, yc, ycok);
      ry[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdY, (int)off, 0, 0));
      yin = (yin & ~(1u << (2 * set + it))) | (off != XOOB ? 1u << (2 * set + it) : 0u);
    }
  };
  auto store_y = [&](int set, int s) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      f32x4 v = ry[set][it];
      if (bnrelu) {
        const float top = (yin >> (2 * set + it)) & 1u ? __builtin_inff() : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(v[e], ysc[e], ysh[e]), 0.f, top);
      }
      u32x2 hi, lo;
      split4(v[0], v[1], v[2], v[3], hi, lo);
      const unsigned a = y_lds(s, it);
      *reinterpret_cast<u32x2*>(Yhi + a) = hi;
      if (NP == 3) *reinterpret_cast<u32x2*>(Ylo + a) = lo;
    }
  };
  // gradient slab: 24 patches (8 row groups x 3 sixteen-channel tiles): group grp takes patch grp, groups 0..7 also patch 16 + grp
  f32x4 rgv[XW_D][2];
  auto load_g = [&](int set, int s) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pi = grp + 16 * it, rg = pi / 3, ct = pi - 3 * rg;
      unsigned off;
      if (NP == 3 && p.g_frame) off = (pi < 24 && s >= 0) ? (unsigned)(s * 32 + 4 * rg + kq) * 192u + (unsigned)((16 * ct + 4 * jq) * 4) : XOOB;      // past the image: out of range -> zeros
      else off = pi < 24 ? pix_off(s * 32 + 4 * rg + kq, p.ldg, 16 * ct + 4 * jq, 16 * ct + 4 * jq < p.N) : XOOB;
      rgv[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdG, (int)off, 0, 0));
    }
  };
  auto store_g = [&](int set, int s) {
    unsigned char* const gb = Gb + (s & 1) * (2 * XW_GPLANE);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pi = grp + 16 * it, rg = pi / 3, ct = pi - 3 * rg;
      if (pi < 24) {
        u32x2 hi, lo;
        if (NP == 3 && p.g_frame) { hi = u32x2{__float_as_uint(rgv[set][it][0]), __float_as_uint(rgv[set][it][1])}; lo = u32x2{__float_as_uint(rgv[set][it][2]), __float_as_uint(rgv[set][it][3])}; }
        else split4(rgv[set][it][0], rgv[set][it][1], rgv[set][it][2], rgv[set][it][3], hi, lo);
        const int row = 4 * rg + kq, rb = row >> 3, pr = (row & 7) ^ ((rb & 1) << 2);
        const unsigned a = (unsigned)(256 * (rb * 3 + ct) + 32 * pr + 8 * jq);
        *reinterpret_cast<u32x2*>(gb + a) = hi;
        if (NP == 3) *reinterpret_cast<u32x2*>(gb + XW_GPLANE + a) = lo;
      }
    }
  };

  // ---- fragment addresses.  Gradient (A operand, rows = n): as in the 1x1 kernel.  Activations (B operand, columns = c of this wave's
  // tile): k group g, half r' reads slab rows 8 g + 4 r' + q shifted by the tap; per tap and half one ring offset (before the slab's base) ----
  const int fq = l16 >> 2, fp = l16 & 3;
  const unsigned frG0 = (unsigned)(256 * (g * 3) + 32 * ((fq) ^ ((g & 1) << 2)) + 8 * fp);
  const unsigned frG1 = (unsigned)(256 * (g * 3) + 32 * ((4 + fq) ^ ((g & 1) << 2)) + 8 * fp);
  // ring byte offset of (tap, half) for slab 0; a slab further = 4 row blocks = 4096 bytes further (mod the 32 KB plane): the row's low
  // three bits and its block's parity - all the swizzle depends on - do not change
  unsigned ybase[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int r = tap / 3, q = tap - 3 * r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = (8 * g + 4 * h + fq + (r - 1) * Wp + (q - 1)) & (XW_RING - 1);
      const int rb = row >> 3, pr = (row & 7) ^ ((rb & 1) << 2);
      ybase[tap][h] = (unsigned)(256 * (rb * 4 + wave) + 32 * pr + 8 * fp);
    }
  }
  auto y_frag = [&](const unsigned char* plane, unsigned soff, int tap) -> bf16x8 {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(plane + ((ybase[tap][0] + soff) & (unsigned)(XW_YPLANE - 1))));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(plane + ((ybase[tap][1] + soff) & (unsigned)(XW_YPLANE - 1))));
    const s16x8 ab = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    return __builtin_bit_cast(bf16x8, ab);
#else
    return bf16x8{};
#endif
  };

  f32x4 acc[9][3];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) acc[tap][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: activation slabs s_begin - ahead .. s_begin + ahead (XW_D loads in flight at a time), gradient slab s_begin; then the loads of
  // targets s_begin + 1 .. s_begin + XW_D - 1 are put in flight ----
  for (int s = s_begin - p.ahead; s <= s_begin + p.ahead; s += XW_D) {
#pragma unroll
    for (int d = 0; d < XW_D; ++d) if (s + d <= s_begin + p.ahead) load_y(d, s + d);
#pragma unroll
    for (int d = 0; d < XW_D; ++d) if (s + d <= s_begin + p.ahead) store_y(d, s + d);
  }
  load_g(0, s_begin);
  store_g(0, s_begin);
#pragma unroll
  for (int d = 1; d < XW_D; ++d) { load_y(d, s_begin + d + p.ahead); load_g(d, s_begin + d); }
  __syncthreads();
  for (int s0 = s_begin; s0 < s_end; s0 += XW_D) {
#pragma unroll
    for (int u = 0; u < XW_D; ++u) {
      const int s = s0 + u;
      if (s >= s_end) break;
      // target s + XW_D (a slab past the end reads as zeros and is never stored) in flight under the MFMAs of XW_D slabs
      load_y(u, s + XW_D + p.ahead); load_g(u, s + XW_D);
      const bool more = s + 1 < s_end;
      const unsigned char* const gb = Gb + (s & 1) * (2 * XW_GPLANE);
      bf16x8 ah[3], al[3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        ah[nt] = tr_frag(gb + frG0, gb + frG1, 256 * nt);
        al[nt] = tr_frag(gb + XW_GPLANE + frG0, gb + XW_GPLANE + frG1, 256 * nt);
      }
      const unsigned soff = ((unsigned)s * 4096u) & (unsigned)(XW_YPLANE - 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bf16x8 bh = y_frag(Yhi, soff, tap), bl = y_frag(Ylo, soff, tap);
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          if (NP == 3) {
            acc[tap][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[nt], bh, acc[tap][nt], 0, 0, 0);
            acc[tap][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bl, acc[tap][nt], 0, 0, 0);
          }
          acc[tap][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bh, acc[tap][nt], 0, 0, 0);
        }
      }
      // ring slot (s + 1 + ahead) & 7 and gradient buffer (s + 1) & 1: last read one slab ago
      if (more) { store_y((u + 1) % XW_D, s + 1 + p.ahead); store_g((u + 1) % XW_D, s + 1); }
      __syncthreads();
    }
  }

  // ---- epilogue: D row = 4 g + r (n), column = l16 (c): f32 atomics into the packed gradient ----
  const int c = c0 + 16 * wave + l16;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nt * 16 + 4 * g + r;
        if (c < p.C && n < p.N) atomicAdd(p.dW + (long)tap * p.wtap + (long)n * p.ldw + c, acc[tap][nt][r]);
      }
}


// =============================================================================================
// 1x1 FORWARD of the dense layers' conv1:  Y[m][n] = sum_c f(X[m][c]) * W[n][c]  (c: the layer's Cin input channels = a prefix of the block
// buffer's rows; n: the Cb bottleneck channels), f = ReLU(BatchNorm), optional per-channel sums of Y and Y^2 (training-mode statistics).
// THREE-way split ("bf16x6"): x = x0 + x1 + x2 exactly (8 + 8 + 8 significant bits), products x0 w0 + x0 w1 + x1 w0 + x0 w2 + x2 w0 + x1 w1 -
// six bf16 MFMAs, float32-equivalent accuracy (dropped terms 2^-24 relative; measured on the reference fixtures: logits within 7.7e-6 of their
// maximum, where the two-way split of the gradient kernels gives 1.0e-4 - too much for the forward's 1e-4 parity bar).  Still 2.7x less
// matrix-pipe time than the f32 MFMA, and this convolution is bound by WRITING Y (758 MB per dense_e2 layer), not by the pipe.
// Structure of xs_dgrad1x1_kernel (8 waves, [row][32 k] images with 3 planes per operand, pre-split weights copied verbatim) with ONE LDS stage
// (84 KB) and two barriers per k-step; 256-pixel x 192-channel tiles.  Measured alternatives (in-process, dense_e2 / dense_e3 shapes): two stages with
// 128-channel tiles - dense_e2 0.42 / 0.61 / 0.94 ms at K = 96 / 192 / 336 against this form's 0.36 / 0.52 / 0.83 (more re-staging of the input per
// output), dense_e3 0.08-0.22 against 0.09-0.25; a persistent workgroup per pixel tile walking its column tiles (stores of tile t draining under
// tile t + 1): 0.47 / 0.63 / 0.96 and half the speed at dense_e3 (one 80-pixel workgroup per CU).  The f32 kernel: 0.45 / 0.70 / 1.15.
// =============================================================================================
// MTW = sixteen-pixel tiles per wave row: 4 -> tiles of <= 256 pixels, one 84 KB stage, ONE workgroup per CU.  (Round 5, measured and not kept: MTW = 2 -
// tiles of <= 128 pixels, a 60 KB stage, TWO workgroups per CU so that one's staging runs beside the other's MFMAs: 9.2 vs 5.8 ms of kernel time per
// step, 52.6 vs 49.3 ms per step - every workgroup re-stages the 36 KB weight slab for half the pixels, and the L2 -> LDS stream of the weights, not the
// staging bubble, is what the kernel waits for.)
constexpr int XF_NTW = 6, XF_BNMAX = 2 * XF_NTW * 16;
constexpr int XF_W_IMG = XF_BNMAX * 64;
constexpr int xf_x_img(int mtw) { return 4 * mtw * 16 * 64; }
constexpr int xf_lds(int mtw) { return 3 * xf_x_img(mtw) + 3 * XF_W_IMG; }

struct XsFwd1Args {
  const float* X; int ldx; int K;          // block buffer [M][ldx], K = Cin contracted channels
  const float* x_scale; const float* x_shift;
  const unsigned char* Wp;                 // [plane 0..2][ksteps][N][64 B]
  float* out; int ldc;
  double* stat0; double* stat1;
  int M, N, ksteps, PT, mtiles, ctiles;
  int abl;                                 // development builds only (RDM_XF1_ABL): timing-only ablations, results wrong by construction: 1 no stores, 2 no MFMAs, 4 no statistics, 8 no split arithmetic (profiles/r05_fwd1x1_ablation.txt)
  unsigned x_bytes, w_bytes;
};
#ifdef RDM_DEV_VARIANTS
#define XF1_ABL(bit) (p.abl & (bit))
#else
#define XF1_ABL(bit) false
#endif

__device__ __forceinline__ void split3x4(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
  const bf16x2 a01 = {(__bf16)v[0], (__bf16)v[1]}, a23 = {(__bf16)v[2], (__bf16)v[3]};
  const float r0 = v[0] - (float)a01[0], r1 = v[1] - (float)a01[1], r2 = v[2] - (float)a23[0], r3 = v[3] - (float)a23[1];
  const bf16x2 b01 = {(__bf16)r0, (__bf16)r1}, b23 = {(__bf16)r2, (__bf16)r3};
  const float q0 = r0 - (float)b01[0], q1 = r1 - (float)b01[1], q2 = r2 - (float)b23[0], q3 = r3 - (float)b23[1];
  const bf16x2 c01 = {(__bf16)q0, (__bf16)q1}, c23 = {(__bf16)q2, (__bf16)q3};
  p0 = u32x2{__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23)};
  p1 = u32x2{__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23)};
  p2 = u32x2{__builtin_bit_cast(unsigned, c01), __builtin_bit_cast(unsigned, c23)};
}

__global__ __launch_bounds__(256) void k_xs_pack_w1_fwd(const float* __restrict__ w, int ldw, int N, int K, int ksteps, unsigned char* __restrict__ Wp) {
  // one thread per (k-step, n, 8-k chunk): 8 consecutive weights of row n -> 16 B of each of the three planes at the swizzled chunk
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int ch = (int)(idx & 3);
  const long r1 = idx >> 2;
  const int n = (int)(r1 % N), j = (int)(r1 / N);
  if (j >= ksteps) return;
  f32x4 v0, v1;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k0 = 32 * j + 8 * ch + e, k1 = k0 + 4;
    v0[e] = k0 < K ? w[(long)n * ldw + k0] : 0.f;
    v1[e] = k1 < K ? w[(long)n * ldw + k1] : 0.f;
  }
  u32x2 a0, a1, a2, b0, b1, b2;
  split3x4(v0, a0, a1, a2);
  split3x4(v1, b0, b1, b2);
  const int chs = ch ^ (((n >> 2) & 1) << 1);
  const long plane = (long)ksteps * N * 64;
  unsigned char* dst = Wp + ((long)j * N + n) * 64 + chs * 16;
  *reinterpret_cast<u32x4*>(dst) = u32x4{a0[0], a0[1], b0[0], b0[1]};
  *reinterpret_cast<u32x4*>(dst + plane) = u32x4{a1[0], a1[1], b1[0], b1[1]};
  *reinterpret_cast<u32x4*>(dst + 2 * plane) = u32x4{a2[0], a2[1], b2[0], b2[1]};
}

template <bool STATS, int NP, int MTW>
__global__ __launch_bounds__(512, MTW == 4 ? 2 : 4) void xs_fwd1x1_kernel(XsFwd1Args p) {
  constexpr int XF_MTW = MTW, XF_X_IMG = xf_x_img(MTW);
  extern __shared__ __attribute__((aligned(1024))) unsigned char xf_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int wc = wave & 1, wp = wave >> 1;
  // work item = (pixel tile, column tile of 192 outputs), column tile fastest: the workgroups of one pixel tile run side by side and its input
  // rows are fetched from HBM once; XCD x works through a contiguous range of items
  const unsigned total = gridDim.x, L = blockIdx.x, xq = L & 7u, qq = total >> 3, rr = total & 7u;
  const unsigned item = xq * qq + (xq < rr ? xq : rr) + (L >> 3);
  const int mt = (int)(item / (unsigned)p.ctiles), ctile = (int)(item - (unsigned)mt * (unsigned)p.ctiles);
  const int BM = p.PT * 16, m0 = mt * BM;
  const int ct0 = ctile * (2 * XF_NTW), nct = min(2 * XF_NTW, p.N / 16 - ct0), BN = nct * 16;
  const int ntw_max = (nct + 1) >> 1, tw0 = wc * ntw_max, ntw = min(ntw_max, nct - tw0);
  const int npw_max = (p.PT + 3) >> 2, pw0 = wp * npw_max, npw = max(0, min(npw_max, p.PT - pw0));
  const __amdgpu_buffer_rsrc_t srdX = xsrd(p.X, p.x_bytes), srdW = xsrd(p.Wp, p.w_bytes);
  const bool bnrelu = p.x_scale != nullptr;
  unsigned char* const Xi = xf_smem;                          // 3 planes of XF_X_IMG
  unsigned char* const Wi = xf_smem + 3 * XF_X_IMG;           // 3 planes of XF_W_IMG

  unsigned x_voff[XF_MTW]; unsigned x_lds[XF_MTW]; bool x_ok[XF_MTW];
  const int f4 = tid & 7;                                     // this thread's float4 of a row's 32 k: the same for all its rows
#pragma unroll
  for (int u = 0; u < XF_MTW; ++u) {
    const int row = (tid + 512 * u) >> 3;
    x_ok[u] = row < BM && m0 + row < p.M;
    x_voff[u] = (unsigned)(m0 + row) * (unsigned)(p.ldx * 4) + (unsigned)(f4 * 16);
    x_lds[u] = (unsigned)(row * 64 + (((f4 >> 1) ^ (((row >> 2) & 1) << 1)) * 16) + (f4 & 1) * 8);
  }
  const int wpieces = BN * 4;                                 // 16-byte pieces of ONE plane of the weight slab
  const unsigned w_plane = (unsigned)p.ksteps * (unsigned)p.N * 64u;
  f32x4 rx[XF_MTW], sc, sh; u32x4 rw[5];                       // weight pieces: 3 planes x BN x 4 <= 2304 = 4.5 per thread
  auto load_slab = [&](int j) {
    const bool kok = 32 * j + 4 * f4 < p.K;
#pragma unroll
    for (int u = 0; u < XF_MTW; ++u)
      rx[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdX, (int)((x_ok[u] && kok) ? x_voff[u] + (unsigned)(j * 128) : XOOB), 0, 0));
    if (bnrelu) {
      sc = kok ? *reinterpret_cast<const f32x4*>(p.x_scale + 32 * j + 4 * f4) : f32x4{0.f, 0.f, 0.f, 0.f};
      sh = kok ? *reinterpret_cast<const f32x4*>(p.x_shift + 32 * j + 4 * f4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int q = tid + 512 * u, pl = q / wpieces, r = q - pl * wpieces;        // piece r of plane pl
      rw[u] = __builtin_amdgcn_raw_buffer_load_b128(srdW, (int)(q < 3 * wpieces ? (unsigned)pl * w_plane + ((unsigned)j * (unsigned)p.N + (unsigned)(ct0 * 16)) * 64u + (unsigned)r * 16u : XOOB), 0, 0);
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int u = 0; u < XF_MTW; ++u) {
      if ((tid + 512 * u) >> 3 < BM) {
        f32x4 v = rx[u];
        if (bnrelu) {
          // a dead element (k past K: loaded as 0, coefficients 0) stays exactly 0; rows past M are never stored
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
        }
        u32x2 p0, p1, p2;
        if (XF1_ABL(8)) { p0 = u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}; p1 = u32x2{__float_as_uint(v[2]), __float_as_uint(v[3])}; p2 = p0; }
        else split3x4(v, p0, p1, p2);
        *reinterpret_cast<u32x2*>(Xi + x_lds[u]) = p0;
        if (NP == 6) {
          *reinterpret_cast<u32x2*>(Xi + XF_X_IMG + x_lds[u]) = p1;
          *reinterpret_cast<u32x2*>(Xi + 2 * XF_X_IMG + x_lds[u]) = p2;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int q = tid + 512 * u, pl = q / wpieces, r = q - pl * wpieces;
      if (q < 3 * wpieces) *reinterpret_cast<u32x4*>(Wi + pl * XF_W_IMG + r * 16) = rw[u];
    }
  };

  f32x4 acc[XF_MTW][XF_NTW];
#pragma unroll
  for (int i = 0; i < XF_MTW; ++i)
#pragma unroll
    for (int t = 0; t < XF_NTW; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned fr = (unsigned)(l16 * 64 + ((g ^ (((l16 >> 2) & 1) << 1)) * 16));

  load_slab(0);
  for (int j = 0; j < p.ksteps; ++j) {
    store_slab();
    __syncthreads();
    if (j + 1 < p.ksteps) load_slab(j + 1);                     // in flight under this slab's MFMAs
    bf16x8 x0[XF_MTW], x1[XF_MTW], x2[XF_MTW];
#pragma unroll
    for (int i = 0; i < XF_MTW; ++i)
      if (i < npw) {
        x0[i] = *reinterpret_cast<const bf16x8*>(Xi + (pw0 + i) * 1024 + fr);
        x1[i] = *reinterpret_cast<const bf16x8*>(Xi + XF_X_IMG + (pw0 + i) * 1024 + fr);
        x2[i] = *reinterpret_cast<const bf16x8*>(Xi + 2 * XF_X_IMG + (pw0 + i) * 1024 + fr);
      }
#pragma unroll
    for (int t = 0; t < XF_NTW; ++t)
      if (t < ntw) {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(Wi + (tw0 + t) * 1024 + fr);
        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(Wi + XF_W_IMG + (tw0 + t) * 1024 + fr);
        const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(Wi + 2 * XF_W_IMG + (tw0 + t) * 1024 + fr);
#pragma unroll
        for (int i = 0; i < XF_MTW; ++i)
          if (i < npw) {                                        // smallest terms first
            if (XF1_ABL(2)) { asm volatile("" :: "v"(w0), "v"(w1), "v"(w2), "v"(x0[i]), "v"(x1[i]), "v"(x2[i])); continue; }
            if (NP == 6) {
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1[i], acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x0[i], acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x2[i], acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x0[i], acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x1[i], acc[i][t], 0, 0, 0);
            }
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0[i], acc[i][t], 0, 0, 0);
          }
      }
    __syncthreads();                                            // every wave is past its reads of the images
  }

  // ---- epilogue: a lane owns output channels n4 .. n4 + 3 (D rows 4 g + r) of pixel m (D column l16) ----
#pragma unroll
  for (int t = 0; t < XF_NTW; ++t) {
    if (t >= ntw) continue;
    const int n4 = (ct0 + tw0 + t) * 16 + 4 * g;
    const bool nok = n4 < p.N;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < XF_MTW; ++i) {
      const int m = m0 + (pw0 + i) * 16 + l16;
      const bool ok = nok && i < npw && m < p.M;
      const f32x4 v = acc[i][t];
      if (ok) {
        if (!XF1_ABL(1)) *reinterpret_cast<f32x4*>(p.out + (long)m * p.ldc + n4) = v;       // (plain global stores: see the note in xs_dgrad3x3_kernel)
        else asm volatile("" :: "v"(v));
        if (STATS) { s0 += v; s1 += v * v; }
      }
    }
    if (STATS && !XF1_ABL(4)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = s0[e], b = s1[e];
        a += __shfl_xor(a, 1); b += __shfl_xor(b, 1);
        a += __shfl_xor(a, 2); b += __shfl_xor(b, 2);
        a += __shfl_xor(a, 4); b += __shfl_xor(b, 4);
        a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
        if (l16 == e) { s0[0] = a; s1[0] = b; }
      }
      if (l16 < 4 && nok && npw > 0) {
        atomicAdd(p.stat0 + n4 + l16, (double)s0[0]);
        atomicAdd(p.stat1 + n4 + l16, (double)s1[0]);
      }
    }
  }
}

}  // namespace

bool xs_wgrad1x1_supported(const WgradArgs& a) {
  const bool one = a.g.KH == 1 && a.g.KW == 1 && a.g.SH == 1 && a.g.SW == 1 && a.g.PH == 0 && a.g.PW == 0 && a.g.H == a.g.Ho && a.g.W == a.g.Wo;
  return one && a.C % 48 == 0 && a.C >= 96 && a.C <= 12 * 192 && a.N % 4 == 0;
}

int launch_xs_wgrad1x1(const WgradArgs& a, hipStream_t s) {
  RDM_CHECK_ARG(xs_wgrad1x1_supported(a), "split-precision 1x1 wgrad: needs a 1x1 / stride 1 convolution with 96 <= C (%d) <= 2304, C a multiple of 48, N (%d) of 4", a.C, a.N);
  RDM_CHECK_ARG(a.ldg % 4 == 0 && a.ldx % 4 == 0 && ((uintptr_t)a.G & 15) == 0 && ((uintptr_t)a.Xs & 15) == 0, "split-precision 1x1 wgrad: strides multiples of 4 floats, operands 16-byte aligned");
  const long M = (long)a.g.B * a.g.Ho * a.g.Wo;
  RDM_CHECK_ARG(!a.g_bf16 || a.xsplit == 1, "1x1 wgrad: bf16 gradient rows exist in the one-product (mixed-precision) form only");
  const long gb = ((M - 1) * a.ldg + a.N) * (a.g_bf16 ? 2 : 4), xb = ((M - 1) * a.ldx + a.C) * 4;
  if (gb >= 0xFFFFFFFFL || xb >= 0xFFFFFFFFL) { set_error("split-precision 1x1 wgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  RDM_CHECK_ARG((!a.g_split && !a.x_split) || a.xsplit != 1, "1x1 wgrad: split rows are operands of the three-product form");
  RDM_CHECK_ARG(!a.x_split || (a.x_scale == nullptr && a.x_shift == nullptr), "1x1 wgrad: split activation rows are already activated (no BatchNorm prologue)");
  XsWgradArgs k{};
  k.g_bf16 = a.g_bf16; k.g_split = a.g_split; k.x_split = a.x_split;
  k.G = a.G; k.ldg = a.ldg; k.N = a.N; k.X = a.Xs; k.ldx = a.ldx; k.C = a.C; k.x_scale = a.x_scale; k.x_shift = a.x_shift;
  k.dW = a.dW; k.ldw = a.ldw; k.M = (int)M; k.g_bytes = (unsigned)gb; k.x_bytes = (unsigned)xb;
  // column tiles of 192 / 144 / 96 channels: the C / 48 units are dealt as evenly as possible over ceil(units / 4) tiles
  const int units = a.C / 48, nct = (units + 3) / 4;
  int c0 = 0;
  for (int t = 0; t < nct; ++t) {
    const int u = units / nct + (t < units % nct ? 1 : 0);
    k.ct_c0[t] = c0; k.ct_nt[t] = 3 * u;
    c0 += 48 * u;
  }
  k.n_ctiles = nct;
  const long tiles = (long)nct * cdiv(a.N, XS_BM), kslabs = (M + XS_BK - 1) / XS_BK;
  // K split: ONE round of the 512 resident workgroups (2 per CU), at least 16 slabs of 32 pixels per split (the f32 atomics of a split's epilogue
  // cost as much as ~8 slabs at dense_e4's widths).  Swept per shape (alone, us; auto = the generic pick_split_k this replaced): dense_e3 C = 480:
  // 102 at 15 splits vs 148 auto; C = 720: 147 at 11 vs 170; dense_e4 C = 432 / 768 / 1248 / 1632: 41 / 55 / 66 / 73 at 8 vs 71 / 80 / 82 / 85;
  // dense_e2 C = 96 / 192: 185 / 291 at 23 vs 206 / 302; a second round (24 splits of 22 tiles) costs 40 % at once
  if (a.split_k > 0) k.split_k = a.split_k;
  else if (tiles > 512) k.split_k = pick_split_k(tiles, kslabs / 2, 256 * 2);
  else k.split_k = (int)std::max<long>(1, std::min<long>(512 / tiles, kslabs / 16));
  if (k.split_k > kslabs) k.split_k = (int)kslabs;
  void* prof = profile_begin(s, 2.0 * (double)M * a.N * a.C, 13);
  RDM_CENSUS("xs_wgrad1x1_kernel/x%d/%s/%s%s", a.xsplit == 1 ? 1 : 3, a.x_scale ? "bn1" : "bn0", k.split_k > 1 ? "splitK" : "split1", a.g_split && a.x_split ? "/rowsGX" : a.g_split ? "/rowsG" : a.x_split ? "/rowsX" : "");
  if (a.xsplit == 1) hipLaunchKernelGGL(xs_wgrad1x1_kernel<1>, dim3((unsigned)(tiles * k.split_k)), dim3(256), 0, s, k);
  else hipLaunchKernelGGL(xs_wgrad1x1_kernel<3>, dim3((unsigned)(tiles * k.split_k)), dim3(256), 0, s, k);
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return 0;
}

size_t xs_dgrad3x3_workspace_bytes(int Cb) { return (size_t)(Cb / 16) * XD_KSTEPS * 2048; }

static int xs_dgrad3_lds_bytes(int W) {
  const int nslots = 1 + XD_BM + 2 * (W + 1);
  const int plane = (nslots * XD_SLOT + 1023) & ~1023;
  return 2 * plane;
}

bool xs_dgrad3x3_supported(const FwdArgs& a) {
  const ConvGeom& g = a.g;
  return g.KH == 3 && g.KW == 3 && g.SH == 1 && g.SW == 1 && g.PH == 1 && g.PW == 1 && g.H == g.Ho && g.W == g.Wo && g.dir == -1 && a.C == 48 &&
         a.N % 16 == 0 && a.N >= 16 && xs_dgrad3_lds_bytes(g.W) <= 156 * 1024 && (g.W + 2) * XD_SLOT < 32768 && a.bias == nullptr && !a.accumulate && !a.add_out;
}

int launch_xs_pack_w3_dgrad(const float* w, long wtap, int ldw, int Cb, void* ws, hipStream_t s) {
  const long threads = (long)(Cb / 16) * XD_KSTEPS * 64;
  hipLaunchKernelGGL(k_xs_pack_w3_dgrad, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, s, w, wtap, ldw, Cb, static_cast<unsigned char*>(ws));
  RDM_LAUNCH_OK();
  return 0;
}

int launch_xs_dgrad3x3(const FwdArgs& a, Epilogue epi, void* ws, size_t ws_bytes, hipStream_t s, int np, bool prepacked) {
  RDM_CHECK_ARG(xs_dgrad3x3_supported(a), "split-precision 3x3 dgrad: needs a 3x3 / stride 1 / pad 1 convolution with 48 gradient channels, N (%d) a multiple of 16, W (%d) <= 339", a.N, a.g.W);
  RDM_CHECK_ARG(epi == EPI_STORE || epi == EPI_MASK_STATS, "split-precision 3x3 dgrad: plain or gate + statistics epilogue only");
  RDM_CHECK_ARG(a.lda % 4 == 0 && a.ldc % 4 == 0 && a.ldw % 4 == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.out & 15) == 0, "split-precision 3x3 dgrad: strides multiples of 4 floats, tensors 16-byte aligned");
  RDM_CHECK_ARG(epi != EPI_MASK_STATS || (a.X && a.x_scale && a.x_shift && a.stat0 && a.stat1 && a.ldx % 4 == 0 && ((uintptr_t)a.X & 15) == 0), "split-precision 3x3 dgrad: the gate needs X, scale, shift and both statistics");
  RDM_CHECK_ARG(ws != nullptr && ((uintptr_t)ws & 15) == 0 && ws_bytes >= xs_dgrad3x3_workspace_bytes(a.N), "split-precision 3x3 dgrad: workspace too small or misaligned (%zu < %zu)", ws_bytes, xs_dgrad3x3_workspace_bytes(a.N));
  const long M = a.M;
  const long gb = ((M - 1) * a.lda + 48) * 4, ob = ((M - 1) * a.ldc + a.N) * 4, xb = a.X ? ((M - 1) * a.ldx + a.N) * 4 : 16;
  if (gb >= 0xFFFFFFFFL || ob >= 0xFFFFFFFFL || xb >= 0xFFFFFFFFL) { set_error("split-precision 3x3 dgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  const int Cb = a.N;
  if (!prepacked) { if (int prc = launch_xs_pack_w3_dgrad(a.Wt, a.wtap, a.ldw, Cb, ws, s)) return prc; }
  XsDgrad3Args k{};
  k.G = a.A; k.ldg = a.lda; k.Wf = static_cast<const unsigned char*>(ws); k.out = a.out; k.ldc = a.ldc;
  k.X = a.X; k.ldx = a.ldx; k.x_scale = a.x_scale; k.x_shift = a.x_shift; k.stat0 = a.stat0; k.stat1 = a.stat1;
  RDM_CHECK_ARG(!a.out_bf16 || np == 1, "3x3 dgrad: bf16 output rows exist in the one-product (mixed-precision) form only");
  k.out_bf16 = a.out_bf16;
  k.B = a.g.B; k.H = a.g.H; k.W = a.g.W; k.M = (int)M; k.Cb = Cb;
  k.nslots = 1 + XD_BM + 2 * (a.g.W + 1);
  k.plane_bytes = (k.nslots * XD_SLOT + 1023) & ~1023;
  k.mtiles = cdiv(M, XD_BM); k.ctiles = cdiv(Cb, XD_BN);
#ifdef RDM_DEV_VARIANTS
  if (getenv("RDM_XD3_ABL")) k.abl = atoi(getenv("RDM_XD3_ABL"));
#endif
  k.g_bytes = (unsigned)gb; k.w_bytes = (unsigned)xs_dgrad3x3_workspace_bytes(Cb); k.x_bytes = (unsigned)xb; k.o_bytes = (unsigned)ob;
  const int lds = xs_dgrad3_lds_bytes(a.g.W);
  // persistent workgroups: as many as are resident at once (2 per CU while two image pairs fit the 160 KB of LDS), each with an equal share of the items
  const long items = (long)k.mtiles * k.ctiles;
  const long slots = 256L * (lds <= 80 * 1024 ? 2 : 1);
  void* prof = profile_begin(s, 2.0 * (double)M * Cb * 432.0, 14);
  RDM_CENSUS("xs_dgrad3x3_kernel/x%d/%s", np == 1 ? 1 : 3, epi == EPI_MASK_STATS ? "MASK_STATS" : "STORE");
  const dim3 grid((unsigned)std::min(items, slots));
#define RDM_XS_D3(MASK_, NP_) do { \
    RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xs_dgrad3x3_kernel<MASK_, NP_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    hipLaunchKernelGGL((xs_dgrad3x3_kernel<MASK_, NP_>), grid, dim3(256), lds, s, k); } while (0)
  if (epi == EPI_MASK_STATS) { if (np == 1) RDM_XS_D3(true, 1); else RDM_XS_D3(true, 3); }
  else { if (np == 1) RDM_XS_D3(false, 1); else RDM_XS_D3(false, 3); }
#undef RDM_XS_D3
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return 0;
}

size_t xs_dgrad1x1_workspace_bytes(int K, int C) { return (size_t)2 * ((K + 31) / 32) * C * 64; }

bool xs_dgrad1x1_supported(const FwdArgs& a) {
  const ConvGeom& g = a.g;
  const bool one = g.KH == 1 && g.KW == 1 && g.SH == 1 && g.SW == 1 && g.PH == 0 && g.PW == 0 && g.H == g.Ho && g.W == g.Wo;
  return one && a.N % 16 == 0 && a.N >= 16 && a.N <= 12 * X1_BNMAX && a.C % 4 == 0 && a.C >= 32 && a.bias == nullptr && !a.accumulate && !a.add_out;
}

int launch_xs_pack_w1_dgrad(const float* w, int ldw, int K, int C, void* ws, hipStream_t s) {
  const int ksteps = (K + 31) / 32;
  const long threads = (long)ksteps * 4 * C;
  hipLaunchKernelGGL(k_xs_pack_w1_dgrad, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, s, w, ldw, K, C, ksteps, static_cast<unsigned char*>(ws));
  RDM_LAUNCH_OK();
  return 0;
}

int launch_xs_dgrad1x1(const FwdArgs& a, Epilogue epi, void* ws, size_t ws_bytes, hipStream_t s, int np, bool prepacked) {
  RDM_CHECK_ARG(xs_dgrad1x1_supported(a), "split-precision 1x1 dgrad: needs a 1x1 / stride 1 convolution, 16 <= N (%d) <= %d a multiple of 16, C (%d) a multiple of 4", a.N, 12 * X1_BNMAX, a.C);
  RDM_CHECK_ARG(epi == EPI_STORE || epi == EPI_MASK_STATS, "split-precision 1x1 dgrad: plain or gate + statistics epilogue only");
  RDM_CHECK_ARG(a.lda % 4 == 0 && a.ldc % 4 == 0 && a.ldw % 4 == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.out & 15) == 0, "split-precision 1x1 dgrad: strides multiples of 4 floats, tensors 16-byte aligned");
  RDM_CHECK_ARG(epi != EPI_MASK_STATS || (a.X && a.x_scale && a.x_shift && a.stat0 && a.stat1 && a.ldx % 4 == 0 && ((uintptr_t)a.X & 15) == 0), "split-precision 1x1 dgrad: the gate needs X, scale, shift and both statistics");
  const int K = a.C, C = a.N, ksteps = (K + 31) / 32;
  RDM_CHECK_ARG(ws != nullptr && ((uintptr_t)ws & 15) == 0 && ws_bytes >= xs_dgrad1x1_workspace_bytes(K, C), "split-precision 1x1 dgrad: workspace too small or misaligned (%zu < %zu)", ws_bytes, xs_dgrad1x1_workspace_bytes(K, C));
  const long M = a.M;
  RDM_CHECK_ARG(!a.a_bf16 || np == 1, "1x1 dgrad: bf16 gradient rows exist in the one-product (mixed-precision) form only");
  const long gb = ((M - 1) * a.lda + K) * (a.a_bf16 ? 2 : 4), xb = a.X ? ((M - 1) * a.ldx + C) * 4 : 16;
  if (gb >= 0xFFFFFFFFL || xb >= 0xFFFFFFFFL) { set_error("split-precision 1x1 dgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  if (!prepacked) { if (int prc = launch_xs_pack_w1_dgrad(a.Wt, a.ldw, K, C, ws, s)) return prc; }
  XsDgrad1Args k{};
  RDM_CHECK_ARG(!a.acc_scaled || epi == EPI_MASK_STATS, "1x1 dgrad: the accumulating epilogue comes with the gate");
  k.acc = a.acc_scaled;
  RDM_CHECK_ARG(!a.a_split || np != 1, "1x1 dgrad: split rows are the operand of the three-product form");
  k.g_split = a.a_split;
  k.G = a.A; k.ldg = a.lda; k.K = K; k.g_bf16 = a.a_bf16; k.Wp = static_cast<const unsigned char*>(ws); k.out = a.out; k.ldc = a.ldc;
  k.X = a.X; k.ldx = a.ldx; k.x_scale = a.x_scale; k.x_shift = a.x_shift; k.stat0 = a.stat0; k.stat1 = a.stat1;
  k.M = (int)M; k.C = C; k.ksteps = ksteps;
  // column tiles of <= 12 sixteen-channel tiles, as even as possible
  const int nct = C / 16, ctiles = (nct + 2 * X1_NTW - 1) / (2 * X1_NTW);
  int c0 = 0;
  for (int t = 0; t < ctiles; ++t) { k.ct_c0[t] = c0; k.ct_n[t] = nct / ctiles + (t < nct % ctiles ? 1 : 0); c0 += k.ct_n[t]; }
  k.ctiles = ctiles;
  // tile height: one workgroup per CU; the cheapest (rounds of 256 workgroups x tile height) wins.  Few pixels: the small-tile instantiation
  // (<= 128 pixels, three k-steps of loads in flight)
  // (tile heights swept per shape, alone: dense_e4 is best at the 128 pixels this picks - 51 / 65 us at C = 1248 / 2064 against 67-83 / 77-126 for
  // 160-320 pixels; at dense_e3 the small instantiation wins below C ~ 500 (90 vs 107-115 us) and loses above (155 vs 139): left on the large one)
  const bool small = M <= 8192;
  const int mtw = small ? X1_MTW_SMALL : X1_MTW_BIG;
  int best_pt = 1; long best_cost = -1;
  for (int pt = 1; pt <= 4 * mtw; ++pt) {
    const long items = (long)cdiv(M, 16 * pt) * ctiles, rounds = (items + 255) / 256;
    const long cost = rounds * (pt * 16 + 24);                  // + a fixed per-tile cost (prologue / epilogue) in pixel units
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_pt = pt; }
  }
  k.PT = best_pt; k.mtiles = cdiv(M, 16 * best_pt);
  k.g_bytes = (unsigned)gb; k.w_bytes = (unsigned)xs_dgrad1x1_workspace_bytes(K, C); k.x_bytes = (unsigned)xb;
  void* prof = profile_begin(s, 2.0 * (double)M * C * K, 16);
  RDM_CENSUS("xs_dgrad1x1_kernel/x%d/%s/%s%s", np == 1 ? 1 : 3, small ? "px128" : "px320", epi == EPI_MASK_STATS ? "MASK_STATS" : "STORE", a.a_split ? "/rowsG" : "");
  const dim3 grid((unsigned)(k.mtiles * k.ctiles));
#define RDM_XS_D1B(MASK_, NP_, MTW_, D_) do { \
    RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xs_dgrad1x1_kernel<MASK_, NP_, MTW_, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, x1_lds(MTW_))); \
    hipLaunchKernelGGL((xs_dgrad1x1_kernel<MASK_, NP_, MTW_, D_>), grid, dim3(512), x1_lds(MTW_), s, k); } while (0)
#define RDM_XS_D1(MASK_, NP_) do { if (small) RDM_XS_D1B(MASK_, NP_, X1_MTW_SMALL, 1); else RDM_XS_D1B(MASK_, NP_, X1_MTW_BIG, 1); } while (0)
  if (epi == EPI_MASK_STATS) { if (np == 1) RDM_XS_D1(true, 1); else RDM_XS_D1(true, 3); }
  else { if (np == 1) RDM_XS_D1(false, 1); else RDM_XS_D1(false, 3); }
#undef RDM_XS_D1
#undef RDM_XS_D1B
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return 0;
}

// Frame image of a 48-channel gradient for xs_wgrad3x3_kernel: dst[u][48] (split rows, xsplit_dev.h) for the padded positions u of the (H + 2) x (W + 2)
// frames of all images, rounded up to whole 32-position slabs; zeros on the border, past the last frame and in channels >= N.
__global__ __launch_bounds__(256) void k_frame_split_rows(const float* __restrict__ G, int ldg, int N, int B, int H, int W, int upad, u32x4* __restrict__ dst) {
  // 252 threads = 21 positions x 12 channel quads per pass; a position's frame coordinates through float reciprocals (exact below 2^21 positions,
  // as in xs_wgrad3x3_kernel) - the first form of this kernel did 64-bit divisions per element and took 172 us per launch for 14 MB
  const int Wp = W + 2, PP = (H + 2) * Wp;
  const float rPP = 1.0f / (float)PP, rWp = 1.0f / (float)Wp;
  const int t = threadIdx.x, pl = (int)((unsigned)t * 2731u >> 15), q = t - pl * 12;          // t / 12, t % 12 for t < 252
  if (t >= 252) return;
  for (int u = blockIdx.x * 21 + pl; u < upad; u += gridDim.x * 21) {
    const int b = (int)(((float)u + 0.5f) * rPP), rem = u - b * PP, yp = (int)(((float)rem + 0.5f) * rWp), xp = rem - yp * Wp;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (b < B && yp >= 1 && yp <= H && xp >= 1 && xp <= W && 4 * q < N) v = *reinterpret_cast<const f32x4*>(G + ((long)(b * H + yp - 1) * W + xp - 1) * ldg + 4 * q);
    dst[(long)u * 12 + q] = split_row4(v[0], v[1], v[2], v[3]);
  }
}

size_t xs_frame_rows_bytes(int B, int H, int W) { return (size_t)(((long)B * (H + 2) * (W + 2) + 31) / 32 * 32) * 192; }

int launch_frame_split_rows(const float* G, int ldg, int N, int B, int H, int W, void* dst, hipStream_t s) {
  RDM_CHECK_ARG(N >= 4 && N <= 48 && N % 4 == 0 && ldg % 4 == 0 && ((uintptr_t)G & 15) == 0 && ((uintptr_t)dst & 15) == 0, "frame_split_rows: N (%d) a multiple of 4 up to 48, ldg a multiple of 4, operands 16-byte aligned", N);
  const long upad = ((long)B * (H + 2) * (W + 2) + 31) / 32 * 32;
  RDM_CHECK_ARG(upad < (1L << 21), "frame_split_rows: %ld padded positions (the float-reciprocal frame arithmetic is exact below 2^21)", upad);
  hipLaunchKernelGGL(k_frame_split_rows, dim3((unsigned)std::min<long>(cdiv(upad, 21), 256 * 16)), dim3(256), 0, s, G, ldg, N, B, H, W, (int)upad, static_cast<u32x4*>(dst));
  RDM_LAUNCH_OK();
  return 0;
}

bool xs_wgrad3x3_supported(const WgradArgs& a) {
  const ConvGeom& g = a.g;
  return g.KH == 3 && g.KW == 3 && g.SH == 1 && g.SW == 1 && g.PH == 1 && g.PW == 1 && g.H == g.Ho && g.W == g.Wo && g.dir == 1 && a.N <= 48 && a.N % 4 == 0 &&
         a.C % 4 == 0 && a.C >= 16 && (g.W + 3 + 31) / 32 <= 3;
}

int launch_xs_wgrad3x3(const WgradArgs& a, hipStream_t s) {
  RDM_CHECK_ARG(xs_wgrad3x3_supported(a), "split-precision 3x3 wgrad: needs a 3x3 / stride 1 / pad 1 convolution with N (%d) <= 48, N and C (%d) multiples of 4, W (%d) <= 93", a.N, a.C, a.g.W);
  RDM_CHECK_ARG(a.ldg % 4 == 0 && a.ldx % 4 == 0 && a.ldw % 4 == 0 && ((uintptr_t)a.G & 15) == 0 && ((uintptr_t)a.Xs & 15) == 0, "split-precision 3x3 wgrad: strides multiples of 4 floats, operands 16-byte aligned");
  const long M = (long)a.g.B * a.g.H * a.g.W;
  RDM_CHECK_ARG(!a.g_frame || a.xsplit != 1, "3x3 wgrad: the frame image is an operand of the three-product form");
  const long gb = a.g_frame ? (long)xs_frame_rows_bytes(a.g.B, a.g.H, a.g.W) : ((M - 1) * a.ldg + a.N) * 4, yb = ((M - 1) * a.ldx + a.C) * 4;
  if (gb >= 0xFFFFFFFFL || yb >= 0xFFFFFFFFL) { set_error("split-precision 3x3 wgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  XsWgrad3Args k{};
  k.g_frame = a.g_frame;
  k.G = a.G; k.ldg = a.ldg; k.Y = a.Xs; k.ldy = a.ldx; k.C = a.C; k.y_scale = a.x_scale; k.y_shift = a.x_shift;
  k.dW = a.dW; k.wtap = a.wtap; k.ldw = a.ldw; k.N = a.N; k.B = a.g.B; k.H = a.g.H; k.W = a.g.W;
  const long U = (long)a.g.B * (a.g.H + 2) * (a.g.W + 2);
  k.nslab = (int)((U + 31) / 32);
  k.ahead = (a.g.W + 3 + 31) / 32;
  const int cblocks = cdiv(a.C, XW_BC);
  // K split: (column blocks x splits) fill the 512 workgroup slots in whole rounds; every split pays 2 * ahead + 1 slabs of prologue
  int best = 1; double best_cost = 1e30;
  const int max_split = a.split_k > 0 ? a.split_k : 64;
  for (int sp = a.split_k > 0 ? a.split_k : 1; sp <= max_split && sp <= std::max(1, k.nslab / 8); ++sp) {
    const long rounds = ((long)cblocks * sp + 511) / 512;
    const double cost = rounds * ((k.nslab + sp - 1) / sp + 2.0 * k.ahead + 3.0 + 8.0);      // + the epilogue's 27 648 atomic adds per workgroup ~ 8 slabs (swept at dense_e2: 11 splits in one round 493 us, 23 in two rounds 511, 12 - one workgroup too many - 730)
    if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
  }
  k.split = best;
  k.g_bytes = (unsigned)gb; k.y_bytes = (unsigned)yb;
  void* prof = profile_begin(s, 2.0 * (double)M * a.N * 9.0 * a.C, 15);
  RDM_CENSUS("xs_wgrad3x3_kernel/x%d/%s%s", a.xsplit == 1 ? 1 : 3, a.x_scale ? "bn1" : "bn0", a.g_frame ? "/frameG" : "");
  if (a.xsplit == 1) hipLaunchKernelGGL(xs_wgrad3x3_kernel<1>, dim3((unsigned)cblocks, (unsigned)k.split), dim3(256), 0, s, k);
  else hipLaunchKernelGGL(xs_wgrad3x3_kernel<3>, dim3((unsigned)cblocks, (unsigned)k.split), dim3(256), 0, s, k);
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return 0;
}

size_t xs_fwd1x1_workspace_bytes(int K, int N) { return (size_t)3 * ((K + 31) / 32) * N * 64; }

bool xs_fwd1x1_supported(const FwdArgs& a) {
  const ConvGeom& g = a.g;
  const bool one = g.KH == 1 && g.KW == 1 && g.SH == 1 && g.SW == 1 && g.PH == 0 && g.PW == 0 && g.H == g.Ho && g.W == g.Wo && g.dir == 1;
  return one && a.N % 16 == 0 && a.N >= 16 && a.C % 4 == 0 && a.C >= 4 && a.bias == nullptr && !a.accumulate && !a.add_out && a.a_sum == nullptr;
}

int launch_xs_pack_w1_fwd(const float* w, int ldw, int N, int K, void* ws, hipStream_t s) {
  const int ksteps = (K + 31) / 32;
  const long threads = (long)ksteps * N * 4;
  hipLaunchKernelGGL(k_xs_pack_w1_fwd, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, s, w, ldw, N, K, ksteps, static_cast<unsigned char*>(ws));
  RDM_LAUNCH_OK();
  return 0;
}

int launch_xs_fwd1x1(const FwdArgs& a, Epilogue epi, void* ws, size_t ws_bytes, hipStream_t s, int np, bool prepacked) {
  RDM_CHECK_ARG(xs_fwd1x1_supported(a), "split-precision 1x1 forward: needs a 1x1 / stride 1 convolution, N (%d) a multiple of 16, C (%d) a multiple of 4, no bias / accumulation", a.N, a.C);
  RDM_CHECK_ARG(epi == EPI_STORE || epi == EPI_STORE_STATS, "split-precision 1x1 forward: plain or statistics epilogue only");
  RDM_CHECK_ARG(a.lda % 4 == 0 && a.ldc % 4 == 0 && a.ldw % 4 == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.out & 15) == 0, "split-precision 1x1 forward: strides multiples of 4 floats, tensors 16-byte aligned");
  RDM_CHECK_ARG((a.a_scale == nullptr) == (a.a_shift == nullptr) && (a.a_scale == nullptr || (((uintptr_t)a.a_scale | (uintptr_t)a.a_shift) & 15) == 0), "split-precision 1x1 forward: scale and shift go together, 16-byte aligned");
  RDM_CHECK_ARG(epi != EPI_STORE_STATS || (a.stat0 && a.stat1), "split-precision 1x1 forward: the statistics epilogue needs both sums");
  const int K = a.C, N = a.N, ksteps = (K + 31) / 32;
  RDM_CHECK_ARG(ws != nullptr && ((uintptr_t)ws & 15) == 0 && ws_bytes >= xs_fwd1x1_workspace_bytes(K, N), "split-precision 1x1 forward: workspace too small or misaligned (%zu < %zu)", ws_bytes, xs_fwd1x1_workspace_bytes(K, N));
  const long M = a.M;
  const long xb = ((M - 1) * a.lda + K) * 4;
  if (xb >= 0xFFFFFFFFL) { set_error("split-precision 1x1 forward: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  if (!prepacked) { if (int prc = launch_xs_pack_w1_fwd(a.Wt, a.ldw, N, K, ws, s)) return prc; }
  XsFwd1Args k{};
  k.X = a.A; k.ldx = a.lda; k.K = K; k.x_scale = a.a_scale; k.x_shift = a.a_shift; k.Wp = static_cast<const unsigned char*>(ws);
  k.out = a.out; k.ldc = a.ldc; k.stat0 = a.stat0; k.stat1 = a.stat1; k.M = (int)M; k.N = N; k.ksteps = ksteps;
  k.ctiles = cdiv(N / 16, 2 * XF_NTW);
#ifdef RDM_DEV_VARIANTS
  if (getenv("RDM_XF1_ABL")) k.abl = atoi(getenv("RDM_XF1_ABL"));
#endif
  constexpr int mtw = 4, slots = 256;
  int best_pt = 1; long best_cost = -1;
  for (int pt = 1; pt <= 4 * mtw; ++pt) {
    const long items = (long)cdiv(M, 16 * pt) * k.ctiles, rounds = (items + slots - 1) / slots;
    const long cost = rounds * (pt * 16 + 24);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_pt = pt; }
  }
  k.PT = best_pt; k.mtiles = cdiv(M, 16 * best_pt);
  k.x_bytes = (unsigned)xb; k.w_bytes = (unsigned)xs_fwd1x1_workspace_bytes(K, N);
  void* prof = profile_begin(s, 2.0 * (double)M * N * K, 17);
  RDM_CENSUS("xs_fwd1x1_kernel/x%d/%s/%s", np == 1 ? 1 : 6, a.a_scale ? "bn1" : "bn0", epi == EPI_STORE_STATS ? "STORE_STATS" : "STORE");
  const dim3 grid((unsigned)(k.mtiles * k.ctiles));
#define RDM_XS_F1M(STATS_, NP_, MTW_) do { \
    RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&xs_fwd1x1_kernel<STATS_, NP_, MTW_>), hipFuncAttributeMaxDynamicSharedMemorySize, xf_lds(MTW_))); \
    hipLaunchKernelGGL((xs_fwd1x1_kernel<STATS_, NP_, MTW_>), grid, dim3(512), xf_lds(MTW_), s, k); } while (0)
#define RDM_XS_F1(STATS_, NP_) RDM_XS_F1M(STATS_, NP_, 4)
  if (epi == EPI_STORE_STATS) { if (np == 1) RDM_XS_F1(true, 1); else RDM_XS_F1(true, 6); }
  else { if (np == 1) RDM_XS_F1(false, 1); else RDM_XS_F1(false, 6); }
#undef RDM_XS_F1
#undef RDM_XS_F1M
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
