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

#include "rdm_common.h"
#include "elementwise.h"
#include "xsplit.h"

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

// x -> (hi, lo) for four values: two packed bf16 pairs each (v_cvt_pk_bf16_f32: round to nearest even)
__device__ __forceinline__ void split4(const float v0, const float v1, const float v2, const float v3, u32x2& hi, u32x2& lo) {
  const bf16x2 h01 = {(__bf16)v0, (__bf16)v1}, h23 = {(__bf16)v2, (__bf16)v3};
  const float r0 = v0 - (float)h01[0], r1 = v1 - (float)h01[1], r2 = v2 - (float)h23[0], r3 = v3 - (float)h23[1];
  const bf16x2 l01 = {(__bf16)r0, (__bf16)r1}, l23 = {(__bf16)r2, (__bf16)r3};
  hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
  lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
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
  const float* X; int ldx; int C;
  const float* x_scale; const float* x_shift;
  float* dW; int ldw;
  int M, split_k;
  unsigned g_bytes, x_bytes;
  int n_ctiles;                 // column tiles; tile t covers channels [ct_c0[t], ct_c0[t] + 16 * ct_nt[t])
  int ct_c0[8], ct_nt[8];
};

constexpr int XS_BM = 128, XS_BK = 32, XS_NTMAX = 12;
constexpr int XS_A_IMG = XS_BK * XS_BM * 2;                 // one plane of the gradient tile: 8 KB
constexpr int XS_B_IMG = XS_BK * XS_NTMAX * 16 * 2;         // one plane of the activation tile: <= 12 KB

template <int NT>
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
    a_voff[it] = (unsigned)row * (unsigned)(p.ldg * 4) + (unsigned)(n * 4);
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
    const unsigned sog = (unsigned)m0 * (unsigned)(p.ldg * 4), sox = (unsigned)m0 * (unsigned)(p.ldx * 4);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const bool ok = m0 + a_row[it] < p.M;
      ra[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srdG, (int)(ok ? a_voff[it] + sog : XOOB), 0, 0));
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
      split4(ra[it][0], ra[it][1], ra[it][2], ra[it][3], hi, lo);
      *reinterpret_cast<u32x2*>(Ahi + a_lds[it]) = hi;
      *reinterpret_cast<u32x2*>(Alo + a_lds[it]) = lo;
    }
#pragma unroll
    for (int it = 0; it < BL; ++it) {
      if (BPATCH % 16 != 0 && it == BL - 1 && grp + 16 * it >= BPATCH) continue;
      f32x4 v = rb[it];
      if (bnrelu) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(Ssc + b_col[it]), sh = *reinterpret_cast<const f32x4*>(Ssc + BN + b_col[it]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(v[e], sc[e], sh[e]), 0.f, bhi[it]);
      }
      u32x2 hi, lo;
      split4(v[0], v[1], v[2], v[3], hi, lo);
      *reinterpret_cast<u32x2*>(Bhi + b_lds[it]) = hi;
      *reinterpret_cast<u32x2*>(Blo + b_lds[it]) = lo;
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
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[cur], acc[i][t], 0, 0, 0);
        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[cur], acc[i][t], 0, 0, 0);
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
    case 6: xs_wgrad1x1_body<6>(p, smem, c0, n0, s_begin, s_end); break;
    case 9: xs_wgrad1x1_body<9>(p, smem, c0, n0, s_begin, s_end); break;
    default: xs_wgrad1x1_body<12>(p, smem, c0, n0, s_begin, s_end); break;
  }
}

}  // namespace

bool xs_wgrad1x1_supported(const WgradArgs& a) {
  const bool one = a.g.KH == 1 && a.g.KW == 1 && a.g.SH == 1 && a.g.SW == 1 && a.g.PH == 0 && a.g.PW == 0 && a.g.H == a.g.Ho && a.g.W == a.g.Wo;
  return one && a.C % 48 == 0 && a.C >= 96 && a.C <= 8 * 192 && a.N % 4 == 0;
}

int launch_xs_wgrad1x1(const WgradArgs& a, hipStream_t s) {
  RDM_CHECK_ARG(xs_wgrad1x1_supported(a), "split-precision 1x1 wgrad: needs a 1x1 / stride 1 convolution with 96 <= C (%d) <= 1536, C a multiple of 48, N (%d) of 4", a.C, a.N);
  RDM_CHECK_ARG(a.ldg % 4 == 0 && a.ldx % 4 == 0 && ((uintptr_t)a.G & 15) == 0 && ((uintptr_t)a.Xs & 15) == 0, "split-precision 1x1 wgrad: strides multiples of 4 floats, operands 16-byte aligned");
  const long M = (long)a.g.B * a.g.Ho * a.g.Wo;
  const long gb = ((M - 1) * a.ldg + a.N) * 4, xb = ((M - 1) * a.ldx + a.C) * 4;
  if (gb >= 0xFFFFFFFFL || xb >= 0xFFFFFFFFL) { set_error("split-precision 1x1 wgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  XsWgradArgs k{};
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
  k.split_k = a.split_k > 0 ? a.split_k : pick_split_k(tiles, kslabs / 2, 256 * 2);     // >= 32 slabs of 32 pixels per split
  if (k.split_k > kslabs) k.split_k = (int)kslabs;
  void* prof = profile_begin(s, 2.0 * (double)M * a.N * a.C, 13);
  RDM_CENSUS("xs_wgrad1x1_kernel/%s/%s", a.x_scale ? "bn1" : "bn0", k.split_k > 1 ? "splitK" : "split1");
  hipLaunchKernelGGL(xs_wgrad1x1_kernel, dim3((unsigned)(tiles * k.split_k)), dim3(256), 0, s, k);
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
