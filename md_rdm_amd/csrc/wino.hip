// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions of the dense layers (48 outputs, K = 9 * Cb), exact-f32 MFMA.
//
// Replaces, for the layers with many pixels (dense_e2 / dense_e3: network/RDM_Net.py:526,528), the direct implicit GEMM of
// csrc/igemm.hip (conv3x3_halo_kernel): the reference's nn.Conv2d(bn_size*growth, growth, 3, padding=1) inside torchvision's
// _DenseLayer.  The 3x3 convolutions are 54 % of the network's multiply-adds; F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input
// patch with 16 multiplies per (input channel, output channel) instead of 36 - 2.25x fewer MFMAs for the same result:
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        (Lavin & Gray; B^T, G, A^T below)
// The float32 error of this form is ~2x the direct f32 sum's (measured 6e-7 of the output's max at K = 24 624), far inside the 2e-5
// operator tolerance and the 1e-4 parity bar.
//
// MI355X mapping (not a cuDNN translation):
//  * the 16 transform positions are 16 INDEPENDENT GEMMs  M_pos[tile][n] = sum_c V_pos[tile][c] * U_pos[n][c]; a 512-thread workgroup
//    (8 wave64s, one per CU: two per SIMD) owns 64 tiles (256 output pixels) x 48 channels; four CONSUMER waves own 4 positions each
//    (4 x (4 x 3) MFMA tiles = 192 accumulator registers; no two waves ever need the same operand fragment), four PRODUCER waves form V.
//  * U (the transformed weights) never touches LDS: it is stored in HBM in FRAGMENT order [slab][pos][n-tile][lane][4] by a tiny
//    transform kernel, so a wave's B fragment is one fully coalesced 1-KiB global_load_dwordx4 (L2 resident: 48 KB per slab, shared
//    by every workgroup).
//  * V (the transformed activations) is formed in registers: a producer thread (tile, channel quad) gathers its 4x4 patch with 16 buffer loads
//    (hardware range check = zero padding outside the tensor), applies the consumer BatchNorm + ReLU, masks the zero-padded positions,
//    runs the two 1-D transforms in place (32 adds per channel) and writes 16 x 16 bytes into a double-buffered, chunk-swizzled LDS image
//    [pos][tile][16 k]: ONE barrier per 16-channel slab.  A fragments are conflict-free ds_read_b128 (4 consecutive k per lane; the
//    k permutation is shared with the B fragments).
//  * measured and NOT built into the plan: the same transform for the dgrad (K = 48 contracted channels, 57 column blocks of outputs at
//    dense_e2).  With three K slabs the expanded operands are not amortised - 149 operand bytes per MFMA against the direct halo
//    kernel's 31 - and the kernel runs at the L2 -> CU feed rate, 1.65 ms against the direct kernel's 1.66 ms (round 3; removed again).
//  * split-K writes per-split partial OUTPUTS (the output transform is linear) with plain stores; a small ordered reduction adds them
//    in a fixed order and takes the per-channel statistics of the sum - deterministic, and 5x the byte rate of f32 atomics.
#include <algorithm>
#include <stdlib.h>

#include "rdm_common.h"
#include "elementwise.h"
#include "wino.h"

namespace rdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned WOOB = 0xFFFFFFFFu;
constexpr int TT = 64;                 // tiles per workgroup
constexpr int LDM = 52;                // epilogue image [pos][32 tiles][LDM]: 4 * LDM = 16 mod 32 -> the four k-groups of a store hit disjoint banks

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wsrd(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// chunk swizzle of the [row][16 floats] LDS image: the 16-byte chunk c of row r sits in slot c ^ f(r), f by (r >> 2) & 3 = {0, 2, 3, 1}.
// With the ds_read_b128 lane groups of gfx950 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; MI355X_MICROARCH.md, LDS) the 16 lanes of
// a group then cover all 16 chunk slots of the 256-byte bank row (checked exhaustively by tools/wino_lds_check.py).
__device__ __forceinline__ int swz(int row) { return (0x78 >> ((row >> 1) & 6)) & 3; }      // {0,2,3,1}[(row >> 2) & 3] packed as 0b01'11'10'00

// ---------------------------------------------------------------------------------------------
// weight transform  U = G g G^T  ->  fragment order [slab][pos][nt][lane = g*16 + l16][e]:  n = nt*16 + l16, c = slab*16 + 4*g + e
// (w_packed is [tap = r*3 + q][n][c])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wino_weight(const float* __restrict__ w, long wtap, int ldw, int N, int C, float* __restrict__ U) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per (slab, nt, lane): produces 16 positions x 4 e
  const int lane = (int)(idx & 63);
  const long r1 = idx >> 6;
  const int nt = (int)(r1 % 3), slab = (int)(r1 / 3);
  if (slab >= C / 16) return;
  const int l16 = lane & 15, g = lane >> 4;
  const int n = nt * 16 + l16;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = slab * 16 + 4 * g + e;
    float gg[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) gg[r][q] = n < N ? w[(long)(r * 3 + q) * wtap + (long)n * ldw + c] : 0.f;
    float t[4][3];                                               // G g
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      t[0][q] = gg[0][q];
      t[1][q] = 0.5f * (gg[0][q] + gg[1][q] + gg[2][q]);
      t[2][q] = 0.5f * (gg[0][q] - gg[1][q] + gg[2][q]);
      t[3][q] = gg[2][q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float u0 = t[i][0], u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]), u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]), u3 = t[i][2];
      const float u[4] = {u0, u1, u2, u3};
#pragma unroll
      for (int j = 0; j < 4; ++j) U[((((long)slab * 16 + (i * 4 + j)) * 3 + nt) * 64 + lane) * 4 + e] = u[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward kernel
// ---------------------------------------------------------------------------------------------
struct WinoFwdArgs {
  const float* A; int lda; int C;                 // raw NHWC input, contracted channels (multiple of 16)
  const float* a_scale; const float* a_shift;     // consumer BatchNorm + ReLU (may be NULL)
  const float* U;                                 // transformed weights, fragment order
  float* out; int ldc; int N;                     // split == 1: the output slice itself;  split > 1: partial[split][M][48]
  int B, H, W, TH, TW, T;                         // T = B * TH * TW tiles
  int split;
  unsigned a_bytes;
};

// Wave specialisation: waves 0-3 are CONSUMERS (one per SIMD; wave w owns transform positions 4w .. 4w+3: 4 x (4 x 3) MFMA tiles = 192
// accumulator registers, 192 MFMAs per 16-channel slab = the SIMD's matrix pipe for 6 144 cycles, next to 16 ds_read_b128 and 12 coalesced
// 1-KiB weight-fragment loads), waves 4-7 are PRODUCERS (the partner wave on each SIMD: gather, BatchNorm + ReLU, transforms, LDS stores
// for slab s+1 while slab s is being multiplied).  The two streams meet at ONE workgroup barrier per slab; the producers' gathers for
// slab s+2 are issued before it and have a whole slab time to land.  (A first version in which all 8 waves did both jobs in lockstep
// ran at 54 % of the matrix pipe: both waves of a SIMD stalled on the same loads at the same time.)
// ABL: timing-only ablations for tools/wino_bench.py (RDM_DEV_VARIANTS builds; results are wrong): 1 no gathers after the prologue,
// 2 no weight-fragment loads in the loop, 4 no MFMAs, 8 no producer arithmetic (BatchNorm, transforms)
template <bool BNRELU, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_fwd_kernel(WinoFwdArgs p) {
  __shared__ __attribute__((aligned(1024))) float smem[2 * 16 * TT * 16];        // V double buffer: 2 x 64 KB; reused by the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int nslab = p.C / 16;
  int s_begin = 0, s_end = nslab;
  if (p.split > 1) {
    const int per = (nslab + p.split - 1) / p.split;
    s_begin = blockIdx.y * per;
    s_end = min(nslab, s_begin + per);
  }
  const int tile0 = blockIdx.x * TT;
  const bool producer = wave >= 4;
  const int pos0 = (wave & 3) * 4;
  f32x4 acc[4][4][3];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[pp][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    if (producer) {
      // ---- producer: thread -> channel quad cq (4 consecutive channels = one 16-byte chunk of the slab) of tile tl ----
      // Everything here competes with the partner wave's MFMAs for the SIMD's issue port (measured: the producers' vector instructions
      // ADD to the slab time, they do not hide under the matrix pipe), so the instruction count is what is minimised: one 16-byte
      // gather per patch position, BatchNorm + ReLU + zero padding in two instructions per element (v_fma, v_med3 against a per-position
      // upper bound that is +inf inside the image and 0 in the padding), the two 1-D transforms in place, one 16-byte LDS store per position.
      const int ptid = tid - 256, tl = ptid >> 2, cq = ptid & 3;
      unsigned voff[16];
      float hi[16];
      {
        const int t = tile0 + tl;
        const bool tok = t < p.T;
        const int tpi = p.TH * p.TW;
        const int b = t / tpi, rem = t - b * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
            const bool ok = tok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            voff[i * 4 + j] = ok ? (unsigned)((b * p.H + y) * p.W + x) * (unsigned)(p.lda * 4) + (unsigned)(cq * 16) : WOOB;
            hi[i * 4 + j] = ok ? __builtin_inff() : 0.f;
          }
      }
      const __amdgpu_buffer_rsrc_t srdA = wsrd(p.A, p.a_bytes);
      const int slot = cq ^ swz(tl);
      float rv[4][16];                               // [channel][patch position]; the SLP vectoriser pairs the transform's adds into v_pk_add_f32
                                                     // (measured 1-3 % faster than scalar adds here; s_setprio for the consumers: no effect)
      auto load_raw = [&](int s) {
        // the slab's channel offset travels in the scalar offset: the range check looks at the vector offset alone, where WOOB marks
        // a padded position (a valid offset + s*64 stays inside the tensor: s*16 + 4*cq + 3 < C)
        const int so = s * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srdA, (int)voff[q], so, 0);
          rv[0][q] = __uint_as_float(v.x); rv[1][q] = __uint_as_float(v.y); rv[2][q] = __uint_as_float(v.z); rv[3][q] = __uint_as_float(v.w);
        }
      };
      auto transform_store = [&](int s, float* Vb) {
        if (BNRELU && !(ABL & 8)) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.a_scale + s * 16 + cq * 4), sh = *reinterpret_cast<const f32x4*>(p.a_shift + s * 16 + cq * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) rv[c][q] = __builtin_amdgcn_fmed3f(fmaf(rv[c][q], sc[c], sh[c]), 0.f, hi[q]);
        }
#define RDM_WINO_1D(v, i0, i1, i2, i3)                                                        \
        { const float d0 = v[i0], d1 = v[i1], d2 = v[i2], d3 = v[i3];                         \
          v[i0] = d0 - d2; v[i1] = d1 + d2; v[i2] = d2 - d1; v[i3] = d1 - d3; }
        if (!(ABL & 8)) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) RDM_WINO_1D(rv[c], j, 4 + j, 8 + j, 12 + j)                      // B^T d: columns
#pragma unroll
            for (int i = 0; i < 4; ++i) RDM_WINO_1D(rv[c], 4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3)      // (.) B: rows
          }
        }
#undef RDM_WINO_1D
#pragma unroll
        for (int pos = 0; pos < 16; ++pos)
          *reinterpret_cast<f32x4*>(Vb + ((pos * TT + tl) * 16 + slot * 4)) = f32x4{rv[0][pos], rv[1][pos], rv[2][pos], rv[3][pos]};
      };
      load_raw(s_begin);
      transform_store(s_begin, smem);
      load_raw(min(s_begin + 1, s_end - 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) {
          transform_store(s + 1, smem + (buf ^ 1) * (16 * TT * 16));      // raw(s+1) has been in flight for a whole slab time
          if (!(ABL & 1)) load_raw(min(s + 2, s_end - 1));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the LDS stores have landed before the barrier releases the readers
        }
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // ---- consumer: positions pos0 .. pos0 + 3 ----
      const f32x4* Uf = reinterpret_cast<const f32x4*>(p.U) + lane;
      auto load_b = [&](int s, int pos, f32x4 (&b)[3]) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) b[nt] = Uf[(((long)s * 16 + pos) * 3 + nt) * 64];
      };
      const int aslot = (g ^ swz(l16)) * 4;
      // Software pipeline, pinned with sched_barrier (left alone, hipcc sinks every load next to its first use and the matrix pipe waits
      // for L2): the B fragments of position pp+1 (three 1-KiB loads) are issued before the 48 MFMAs of position pp, the A fragment of
      // M-tile mt+1 (one ds_read_b128) before the 12 MFMAs of M-tile mt.  MFMA order mt -> e -> nt: an accumulator is revisited after 3
      // MFMAs (96 cycles > the 40-cycle dependent latency).
      f32x4 bq[2][3], aq[2];
      auto load_a1 = [&](const float* Vb, int pos, int mt) { return *reinterpret_cast<const f32x4*>(Vb + ((pos * TT + mt * 16 + l16) * 16 + aslot)); };
      load_b(s_begin, pos0, bq[0]);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int s = s_begin; s < s_end; ++s) {
        const float* Vb = smem + ((s - s_begin) & 1) * (16 * TT * 16);
        aq[0] = load_a1(Vb, pos0, 0);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (!(ABL & 2)) {
            if (pp < 3) load_b(s, pos0 + pp + 1, bq[(pp + 1) & 1]);
            else load_b(min(s + 1, s_end - 1), pos0, bq[0]);
          } else if (s == s_begin) load_b(s, pos0 + pp, bq[(pp + 1) & 1]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int cur = (pp * 4 + mt) & 1;
            if (mt < 3) aq[cur ^ 1] = load_a1(Vb, pos0 + pp, mt + 1);
            else if (pp < 3) aq[cur ^ 1] = load_a1(Vb, pos0 + pp + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int nt = 0; nt < 3; ++nt)
                if (!(ABL & 4)) acc[pp][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[cur][e], bq[pp & 1][nt][e], acc[pp][mt][nt], 0, 0, 0);
                else asm volatile("" :: "v"(aq[cur][e]), "v"(bq[pp & 1][nt][e]));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  }
  __syncthreads();

  // ---- epilogue: M (16 positions, held by the consumers) -> LDS, two halves of 32 tiles; Y = A^T M A; store ----
  float* Ms = smem;
  const long Mtot = (long)p.B * p.H * p.W;
  float* dst = p.split > 1 ? p.out + (long)blockIdx.y * Mtot * 48 : p.out;
  const int ldo = p.split > 1 ? 48 : p.ldc;
  const int tpi = p.TH * p.TW;
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (!producer) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp)
#pragma unroll
        for (int mtl = 0; mtl < 2; ++mtl)
#pragma unroll
          for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Ms[((pos0 + pp) * 32 + mtl * 16 + 4 * g + r) * LDM + nt * 16 + l16] = h == 0 ? acc[pp][mtl][nt][r] : acc[pp][2 + mtl][nt][r];
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
      const int idx = tid + it * 512;
      const int tl2 = idx / 48, n = idx - tl2 * 48;
      const int t = tile0 + h * 32 + tl2;
      if (t < p.T && n < p.N) {
        float m[16];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) m[pos] = Ms[(pos * 32 + tl2) * LDM + n];
        float u[2][4];                                              // A^T M: rows
#pragma unroll
        for (int j = 0; j < 4; ++j) { u[0][j] = m[j] + m[4 + j] + m[8 + j]; u[1][j] = m[4 + j] - m[8 + j] - m[12 + j]; }
        const int b = t / tpi, rem = t - b * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          const float y0 = u[a2][0] + u[a2][1] + u[a2][2], y1 = u[a2][1] - u[a2][2] - u[a2][3];
          const int y = 2 * ty + a2, x = 2 * tx;
          if (y < p.H) {
            float* o = dst + ((long)(b * p.H + y) * p.W + x) * ldo + n;
            o[0] = y0;
            if (x + 1 < p.W) o[ldo] = y1;
          }
        }
      }
    }
    __syncthreads();
  }
}

// ordered sum of the split partials [split][M][48] -> out (pixel stride ldc), optional per-channel statistics of the sum
__global__ __launch_bounds__(256) void k_wino_reduce(const float* __restrict__ part, int split, long M, int N, float* __restrict__ out, int ldc,
                                                     double* stat0, double* stat1, int rows_per_block) {
  const int n4 = threadIdx.x % 12, rl = threadIdx.x / 12;          // 12 float4 columns x 21 row lanes (252 of 256 threads)
  const long m0 = (long)blockIdx.x * rows_per_block;
  const long m1 = min(M, m0 + rows_per_block);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (rl < 21)
    for (long m = m0 + rl; m < m1; m += 21) {
      f32x4 v = *reinterpret_cast<const f32x4*>(part + m * 48 + n4 * 4);
      for (int s = 1; s < split; ++s) v += *reinterpret_cast<const f32x4*>(part + ((long)s * M + m) * 48 + n4 * 4);
      float* o = out + m * ldc + n4 * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n4 * 4 + e < N) o[e] = v[e];
      s0 += v; s1 += v * v;
    }
  if (stat0 == nullptr) return;
  // block reduction: 21 row lanes -> 1
  __shared__ float red0[21][48], red1[21][48];
  if (rl < 21) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red0[rl][n4 * 4 + e] = s0[e]; red1[rl][n4 * 4 + e] = s1[e]; }
  }
  __syncthreads();
  if (threadIdx.x < 48 && (int)threadIdx.x < N) {
    double a = 0, b = 0;
    for (int r = 0; r < 21; ++r) { a += red0[r][threadIdx.x]; b += red1[r][threadIdx.x]; }
    atomicAdd(stat0 + threadIdx.x, a);
    atomicAdd(stat1 + threadIdx.x, b);
  }
}



// =============================================================================================
// The same forward on the bf16 matrix pipe, float32-EQUIVALENT ("bf16x6", round 5): the 16 position GEMMs of F(2x2, 3x3) with both operands split
// three ways AFTER their transforms,  v = v0 + v1 + v2  (bf16 each, 24 significant bits together), u likewise, and
//     v u ~ v0 u0 + v1 u0 + v0 u1 + v2 u0 + v0 u2 + v1 u1            (dropped terms 2^-24 relative; every bf16 x bf16 product exact in float32)
// - the arithmetic of xs_fwd1x1_kernel (xsplit.hip), 6 x 1 / 2.25 = 2.7x the direct convolution's products on a pipe that is 16x faster than the
// f32 MFMA's.  Replaces conv3x3_wino_fwd_kernel for conv2 of dense_e2 / dense_e3 (torchvision _DenseLayer.conv2, network/RDM_Net.py:526,528).
//  * K packing: a slab is 16 channels, a v_mfma_f32_16x16x32_bf16 contracts 32 k-slots - two PLANES of the slab side by side.  Three MFMAs
//    per (position, 16-tile block, 16-output block):   (v0|v0).(u0|u1)   (v1|v1).(u0|u1)   (v2|v0).(u0|u2)   = the six products.
//  * V image in LDS: one 96-byte row per (position, tile) = [v0: 16 ch | v1 | v2]; a fragment's k-group g reads 16 bytes of the row (8 channels of
//    one plane); lanes of k-groups g and g + 2 of (v0|v0) / (v1|v1) read the SAME bytes - an LDS broadcast.  Row stride 96 B = 6 sixteen-byte
//    slots: rows l16 = 0 .. 15 start in distinct even slots (6 l16 mod 16), the second k-half in the odd ones - conflict-free without a swizzle -
//    and so are the producers' 8-byte stores (12 tl + cq mod 32).
//  * 48 tiles per workgroup (3 consumer M-tiles: 144 accumulator registers - 64 tiles would need 192 + 72 of weight fragments), so the image
//    (72 KB) is DOUBLE-buffered like the f32 kernel's; 3 producer waves (192 threads = 48 tiles x 4 channel quads) + 4 consumer waves.
//  * U: k_wino_weight_x6 writes, per (slab, position, 16-output block), the three planes as 512-byte blocks [k-half][n][8 ch]; the two B
//    fragments (u0|u1), (u0|u2) are addressed into them per lane: 2 KiB per (position, output block) into registers, 1.5 KiB of it unique.
// =============================================================================================
constexpr int TX = 48;                                    // tiles per workgroup
constexpr int XROW = 96;                                  // bytes per (position, tile) row of the V image
constexpr int XIMG = 16 * TX * XROW;                      // one buffer: 73 728 B
constexpr int XU_BLOCK = 512, XU_PNT = 3 * XU_BLOCK;      // bytes per plane block / per (slab, position, nt)
typedef short s16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2w __attribute__((ext_vector_type(2)));

// x -> three bf16 planes for four values (round to nearest even at every step: x0 + x1 + x2 = x to 24 bits)
__device__ __forceinline__ void wsplit3x4(const float a, const float b, const float c, const float d, u32x2& p0, u32x2& p1, u32x2& p2) {
  const bf16x2w a01 = {(__bf16)a, (__bf16)b}, a23 = {(__bf16)c, (__bf16)d};
  const float r0 = a - (float)a01[0], r1 = b - (float)a01[1], r2 = c - (float)a23[0], r3 = d - (float)a23[1];
  const bf16x2w b01 = {(__bf16)r0, (__bf16)r1}, b23 = {(__bf16)r2, (__bf16)r3};
  const float q0 = r0 - (float)b01[0], q1 = r1 - (float)b01[1], q2 = r2 - (float)b23[0], q3 = r3 - (float)b23[1];
  const bf16x2w c01 = {(__bf16)q0, (__bf16)q1}, c23 = {(__bf16)q2, (__bf16)q3};
  p0 = u32x2{__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, a23)};
  p1 = u32x2{__builtin_bit_cast(unsigned, b01), __builtin_bit_cast(unsigned, b23)};
  p2 = u32x2{__builtin_bit_cast(unsigned, c01), __builtin_bit_cast(unsigned, c23)};
}

// U = G g G^T, split three ways, block order [slab][pos][nt][plane][k-half h][n = nt*16 + l16][8 ch: c = slab*16 + 8 h + e]
__global__ __launch_bounds__(256) void k_wino_weight_x6(const float* __restrict__ w, long wtap, int ldw, int N, int C, unsigned char* __restrict__ U) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per (slab, nt, l16, h, quad): 4 channels x 16 positions
  const int quad = (int)(idx & 1), h = (int)((idx >> 1) & 1), l16 = (int)((idx >> 2) & 15);
  const long r1 = idx >> 6;
  const int nt = (int)(r1 % 3), slab = (int)(r1 / 3);
  if (slab >= C / 16) return;
  const int n = nt * 16 + l16;
  float u[16][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = slab * 16 + 8 * h + 4 * quad + e;
    float gg[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) gg[r][q] = n < N ? w[(long)(r * 3 + q) * wtap + (long)n * ldw + c] : 0.f;
    float t[4][3];                                               // G g
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      t[0][q] = gg[0][q];
      t[1][q] = 0.5f * (gg[0][q] + gg[1][q] + gg[2][q]);
      t[2][q] = 0.5f * (gg[0][q] - gg[1][q] + gg[2][q]);
      t[3][q] = gg[2][q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i * 4 + 0][e] = t[i][0];
      u[i * 4 + 1][e] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      u[i * 4 + 2][e] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      u[i * 4 + 3][e] = t[i][2];
    }
  }
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) {
    u32x2 p0, p1, p2;
    wsplit3x4(u[pos][0], u[pos][1], u[pos][2], u[pos][3], p0, p1, p2);
    unsigned char* dst = U + (((long)slab * 16 + pos) * 3 + nt) * XU_PNT + h * 256 + l16 * 16 + quad * 8;
    *reinterpret_cast<u32x2*>(dst) = p0;
    *reinterpret_cast<u32x2*>(dst + XU_BLOCK) = p1;
    *reinterpret_cast<u32x2*>(dst + 2 * XU_BLOCK) = p2;
  }
}

struct WinoX6Args {
  const float* A; int lda; int C;
  const float* a_scale; const float* a_shift;
  const unsigned char* U;
  float* out; int ldc; int N;
  int B, H, W, TH, TW, T;
  int split;
  int abl;                 // development builds only: timing-only ablations (RDM_WX6_ABL; results wrong by construction): 1 every gather of a thread from one pixel,
                           // 2 no weight loads in the loop, 4 no MFMAs, 8 no split, 16 no LDS stores (profiles/r05_wino_x6_ablation.txt)
  unsigned a_bytes, u_bytes;
};

#ifdef RDM_DEV_VARIANTS
#define WX6_ABL(bit) (p.abl & (bit))
#else
#define WX6_ABL(bit) false
#endif
template <bool BNRELU>
__global__ __launch_bounds__(448, 2) void conv3x3_wino_x6_kernel(WinoX6Args p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char wx_smem[];      // V double buffer: 2 x XIMG; reused by the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int nslab = p.C / 16;
  int s_begin = 0, s_end = nslab;
  if (p.split > 1) {
    const int per = (nslab + p.split - 1) / p.split;
    s_begin = blockIdx.y * per;
    s_end = min(nslab, s_begin + per);
  }
  const int tile0 = blockIdx.x * TX;
  const bool producer = wave >= 4;
  const int pos0 = (wave & 3) * 4;
  f32x4 acc[4][3][3];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp)
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) acc[pp][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    if (producer) {
      // ---- producer: thread -> channel quad cq of tile tl; gather, BatchNorm + ReLU + padding mask, the two 1-D transforms (all as in the f32
      // kernel), then the three-way split and three 8-byte LDS stores per position ----
      const int ptid = tid - 256, tl = ptid >> 2, cq = ptid & 3;
      unsigned voff[16];
      float hi[16];
      {
        const int t = tile0 + tl;
        const bool tok = t < p.T;
        const int tpi = p.TH * p.TW;
        const int b = t / tpi, rem = t - b * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int y = 2 * ty - 1 + i, x = 2 * tx - 1 + j;
            const bool ok = tok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            voff[i * 4 + j] = ok ? (unsigned)((b * p.H + y) * p.W + x) * (unsigned)(p.lda * 4) + (unsigned)(cq * 16) : WOOB;
            hi[i * 4 + j] = ok ? __builtin_inff() : 0.f;
          }
        if (WX6_ABL(1)) {
#pragma unroll
          for (int q = 0; q < 16; ++q) voff[q] = voff[5];
        }
      }
      const __amdgpu_buffer_rsrc_t srdA = wsrd(p.A, p.a_bytes);
      const unsigned rowoff = (unsigned)(tl * XROW + cq * 8);
      float rv[4][16];
      auto load_raw = [&](int s) {
        const int so = s * 64;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srdA, (int)voff[q], so, 0);
          rv[0][q] = __uint_as_float(v.x); rv[1][q] = __uint_as_float(v.y); rv[2][q] = __uint_as_float(v.z); rv[3][q] = __uint_as_float(v.w);
        }
      };
      auto transform_store = [&](int s, unsigned char* Vb) {
        if (BNRELU) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.a_scale + s * 16 + cq * 4), sh = *reinterpret_cast<const f32x4*>(p.a_shift + s * 16 + cq * 4);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) rv[c][q] = __builtin_amdgcn_fmed3f(fmaf(rv[c][q], sc[c], sh[c]), 0.f, hi[q]);
        }
#define RDM_WINO_1D(v, i0, i1, i2, i3)                                                        \
        { const float d0 = v[i0], d1 = v[i1], d2 = v[i2], d3 = v[i3];                         \
          v[i0] = d0 - d2; v[i1] = d1 + d2; v[i2] = d2 - d1; v[i3] = d1 - d3; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) RDM_WINO_1D(rv[c], j, 4 + j, 8 + j, 12 + j)                      // B^T d: columns
#pragma unroll
          for (int i = 0; i < 4; ++i) RDM_WINO_1D(rv[c], 4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3)      // (.) B: rows
        }
#undef RDM_WINO_1D
        if (tl < TX) {
#pragma unroll
          for (int pos = 0; pos < 16; ++pos) {
            u32x2 p0, p1, p2;
            if (WX6_ABL(8)) { p0 = u32x2{__float_as_uint(rv[0][pos]), __float_as_uint(rv[1][pos])}; p1 = u32x2{__float_as_uint(rv[2][pos]), __float_as_uint(rv[3][pos])}; p2 = p0; }
            else wsplit3x4(rv[0][pos], rv[1][pos], rv[2][pos], rv[3][pos], p0, p1, p2);
            unsigned char* row = Vb + pos * (TX * XROW) + rowoff;
            if (!WX6_ABL(16)) {
            *reinterpret_cast<u32x2*>(row) = p0;
            *reinterpret_cast<u32x2*>(row + 32) = p1;
            *reinterpret_cast<u32x2*>(row + 64) = p2;
            } else asm volatile("" :: "v"(p0), "v"(p1), "v"(p2));
          }
        }
      };
      load_raw(s_begin);
      transform_store(s_begin, wx_smem);
      load_raw(min(s_begin + 1, s_end - 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) {
          transform_store(s + 1, wx_smem + (buf ^ 1) * XIMG);              // raw(s+1) has been in flight for a whole slab time
          load_raw(min(s + 2, s_end - 1));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the LDS stores have landed before the barrier releases the readers
        }
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // ---- consumer: positions pos0 .. pos0 + 3.  Per (position, 16-output block) TWO weight fragments, (u0|u1) and (u0|u2); per (position,
      // 16-tile block) THREE activation fragments from LDS, (v0|v0), (v1|v1), (v2|v0):
      //     (v0|v0).(u0|u1) + (v1|v1).(u0|u1) + (v2|v0).(u0|u2)  =  v0 u0 + v0 u1 + v1 u0 + v1 u1 + v2 u0 + v0 u2.
      // (The first form of this kernel duplicated on the WEIGHT side - three 1-KiB fragments per position and output block, 36 KB per position
      // time of 432 cycles = 85 B/clk per CU through a 64 B/clk L1 - and ran no faster than the f32 kernel: 6.69 vs 6.69 ms per step.  Duplicates
      // now sit on the LDS side, where two lanes reading one address is a broadcast.)
      // The weight fragments run TWO positions ahead of their MFMAs through a ring of three register sets: a position is 27 MFMAs = 432 cycles,
      // less than an L2 round trip under load.  The position walk is unrolled over 12 positions (3 slabs) so that ring slot, accumulator and
      // LDS buffer indices are all compile-time. ----
      const __amdgpu_buffer_rsrc_t srdU = wsrd(p.U, p.u_bytes);
      const unsigned h = (unsigned)(g & 1), hi2 = (unsigned)(g >> 1);
      const unsigned ub1 = (hi2 ? (unsigned)XU_BLOCK : 0u) + h * 256u + (unsigned)l16 * 16u;        // (u0|u1)
      const unsigned ub2 = (hi2 ? 2u * XU_BLOCK : 0u) + h * 256u + (unsigned)l16 * 16u;             // (u0|u2)
      bf16x8w bq[3][3][2];
      auto load_b = [&](int s, int pos, bf16x8w (&b)[3][2]) {
        const int so = ((s * 16 + pos) * 3) * XU_PNT;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          b[nt][0] = __builtin_bit_cast(bf16x8w, __builtin_amdgcn_raw_buffer_load_b128(srdU, (int)ub1, so + nt * XU_PNT, 0));
          b[nt][1] = __builtin_bit_cast(bf16x8w, __builtin_amdgcn_raw_buffer_load_b128(srdU, (int)ub2, so + nt * XU_PNT, 0));
        }
      };
      const unsigned arow = (unsigned)(l16 * XROW);
      const unsigned a0off = arow + h * 16u, a1off = arow + 32u + h * 16u, a2off = arow + (hi2 ? h * 16u : 64u + h * 16u);      // (v0|v0), (v1|v1), (v2|v0)
      bf16x8w aq[2][3];
      auto load_a = [&](const unsigned char* Vb, int pos, int mt, bf16x8w (&a)[3]) {
        const unsigned char* r = Vb + (pos * TX + mt * 16) * XROW;
        a[0] = *reinterpret_cast<const bf16x8w*>(r + a0off);
        a[1] = *reinterpret_cast<const bf16x8w*>(r + a1off);
        a[2] = *reinterpret_cast<const bf16x8w*>(r + a2off);
      };
      const int total = (s_end - s_begin) * 4;                       // positions this wave walks: q -> slab s_begin + (q >> 2), position pos0 + (q & 3)
      load_b(s_begin, pos0, bq[0]);
      load_b(s_begin, pos0 + 1, bq[1]);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int base = 0; base < total; base += 12) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
          const int q = base + u;
          if (q < total) {                                           // (wave-uniform)
            const int pp = u & 3;
            const unsigned char* Vb = wx_smem + ((q >> 2) & 1) * XIMG;
            if (q + 2 < total && !WX6_ABL(2)) load_b(s_begin + ((q + 2) >> 2), pos0 + ((u + 2) & 3), bq[(u + 2) % 3]);
            if (pp == 0) load_a(Vb, pos0, 0, aq[0]);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
              const int cur = (pp * 3 + mt) & 1;
              if (mt < 2) load_a(Vb, pos0 + pp, mt + 1, aq[cur ^ 1]);
              else if (pp < 3) load_a(Vb, pos0 + pp + 1, 0, aq[cur ^ 1]);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int nt = 0; nt < 3; ++nt) {                       // small terms first
                if (WX6_ABL(4)) { asm volatile("" :: "v"(aq[cur][0]), "v"(aq[cur][1]), "v"(aq[cur][2]), "v"(bq[u % 3][nt][0]), "v"(bq[u % 3][nt][1])); continue; }
                acc[pp][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[cur][2], bq[u % 3][nt][1], acc[pp][mt][nt], 0, 0, 0);      // v2 u0 + v0 u2
                acc[pp][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[cur][1], bq[u % 3][nt][0], acc[pp][mt][nt], 0, 0, 0);      // v1 u0 + v1 u1
                acc[pp][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[cur][0], bq[u % 3][nt][0], acc[pp][mt][nt], 0, 0, 0);      // v0 u0 + v0 u1
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (pp == 3) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
            }
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- epilogue: M (16 positions, held by the consumers) -> LDS, one 16-tile block at a time; Y = A^T M A; store ----
  float* Ms = reinterpret_cast<float*>(wx_smem);
  const long Mtot = (long)p.B * p.H * p.W;
  float* dst = p.split > 1 ? p.out + (long)blockIdx.y * Mtot * 48 : p.out;
  const int ldo = p.split > 1 ? 48 : p.ldc;
  const int tpi = p.TH * p.TW;
#pragma unroll 1
  for (int hblk = 0; hblk < 3; ++hblk) {
    if (!producer) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Ms[((pos0 + pp) * 16 + 4 * g + r) * LDM + nt * 16 + l16] = hblk == 0 ? acc[pp][0][nt][r] : hblk == 1 ? acc[pp][1][nt][r] : acc[pp][2][nt][r];
    }
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 448;
      const int tl2 = idx / 48, n = idx - tl2 * 48;
      const int t = tile0 + hblk * 16 + tl2;
      if (tl2 < 16 && t < p.T && n < p.N) {
        float m[16];
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) m[pos] = Ms[(pos * 16 + tl2) * LDM + n];
        float u[2][4];                                              // A^T M: rows
#pragma unroll
        for (int j = 0; j < 4; ++j) { u[0][j] = m[j] + m[4 + j] + m[8 + j]; u[1][j] = m[4 + j] - m[8 + j] - m[12 + j]; }
        const int b = t / tpi, rem = t - b * tpi;
        const int ty = rem / p.TW, tx = rem - ty * p.TW;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          const float y0 = u[a2][0] + u[a2][1] + u[a2][2], y1 = u[a2][1] - u[a2][2] - u[a2][3];
          const int y = 2 * ty + a2, x = 2 * tx;
          if (y < p.H) {
            float* o = dst + ((long)(b * p.H + y) * p.W + x) * ldo + n;
            o[0] = y0;
            if (x + 1 < p.W) o[ldo] = y1;
          }
        }
      }
    }
    __syncthreads();
  }
}

// =============================================================================================
// wgrad:  dW[r*3+q][n][c] = sum_m dOut[m][n] * f(Y[m + (r-1) W + (q-1)][c])     (K = all pixels, 48 x Cb outputs per tap)
// Winograd F(3x3, 2x2): per 2x2 output tile the 3x3 filter gradient is  A3^T [ (G2 dy G2^T) (.) (B^T d B) ] A3  (16 multiplies instead
// of 36), and the sum over tiles commutes with the output transform:  Q[pos][n][c] = sum_tiles Vy[pos][tile][n] * Vd[pos][tile][c]  are 16
// GEMMs whose contraction runs over the TILES; dW = A3^T Q A3 is applied once, to the reduced Q.  Vd = B^T d B is the forward kernel's
// transformed activation (same BatchNorm + ReLU prologue, same padding rule); Vy comes from a pre-pass in A-fragment order.
//   B^T as above,  G2 = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]],  A3^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]
// Workgroup = 16 positions x 48 n x 64 c (4 consumer waves x 192 accumulators) over a range of tile slabs (16 tiles per slab); the
// producers gather the 4x4 patches of the slab's 16 tiles for the workgroup's 64 channels, and the LDS image is [pos][tile][64 c] with the
// 16-float chunks XOR-swizzled by (tile & 3): 16-byte stores and 4-byte fragment reads are both conflict-free.  Partial Q per K split
// goes to scratch with plain stores; k_wino_wgrad_reduce sums the splits in a fixed order and applies A3^T . A3: no atomics anywhere,
// the weight gradient is bit-reproducible (the direct row kernel adds 15 split partials with f32 atomics: 68 MB of fabric writes for a
// 4.7 MB result).
// =============================================================================================
// pre-pass: Vy = G2 dy G2^T, A-fragment order [slab][pos][nt][lane = g*16 + l16][e]: n = nt*16 + l16, tile = slab*16 + 4*e + g.
// One 192-thread workgroup per slab: the slab's 16 tiles x 2x2 pixels x 48 channels go through LDS (coalesced 16-byte loads), thread
// (nt, lane) then owns channel n of the tiles g, 4+g, 8+g, 12+g and writes its float4 of every position: 1-KiB coalesced stores.
__global__ __launch_bounds__(192) void k_wino_gradout_transform(const float* __restrict__ G, int ldg, int N, unsigned g_bytes, int B, int H, int W, int TH, int TW,
                                                                int T, int nslab, float* __restrict__ Vy) {
  __shared__ float gs[16][4][48 + 1];
  const int slab = blockIdx.x, tid = threadIdx.x;
  const int tpi = TH * TW;
  const __amdgpu_buffer_rsrc_t srd = wsrd(G, g_bytes);
  for (int i = tid; i < 16 * 4 * 12; i += 192) {                   // (tile, pixel, channel quad): 12 lanes cover a pixel's 48 channels
    const int quad = i % 12, px = (i / 12) & 3, tl = i / 48;
    const long t = (long)slab * 16 + tl;
    const int b = (int)(t / tpi), rem = (int)(t - (long)b * tpi);
    const int ty = rem / TW, tx = rem - ty * TW;
    const int y = 2 * ty + (px >> 1), x = 2 * tx + (px & 1);
    const bool ok = t < T && y < H && x < W && quad * 4 < N;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srd, ok ? (int)((unsigned)((b * H + y) * W + x) * (unsigned)(ldg * 4) + (unsigned)(quad * 16)) : (int)WOOB, 0, 0);
    gs[tl][px][quad * 4 + 0] = __uint_as_float(v.x); gs[tl][px][quad * 4 + 1] = __uint_as_float(v.y);
    gs[tl][px][quad * 4 + 2] = __uint_as_float(v.z); gs[tl][px][quad * 4 + 3] = __uint_as_float(v.w);
  }
  __syncthreads();
  const int nt = tid >> 6, lane = tid & 63, l16 = lane & 15, g = lane >> 4, n = nt * 16 + l16;
  f32x4 out[16];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int tl = 4 * e + g;
    const float y00 = gs[tl][0][n], y01 = gs[tl][1][n], y10 = gs[tl][2][n], y11 = gs[tl][3][n];
    float t2[4][2];                                               // G2 dy: rows
    t2[0][0] = y00; t2[1][0] = 0.5f * (y00 + y10); t2[2][0] = 0.5f * (y00 - y10); t2[3][0] = y10;
    t2[0][1] = y01; t2[1][1] = 0.5f * (y01 + y11); t2[2][1] = 0.5f * (y01 - y11); t2[3][1] = y11;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[i * 4 + 0][e] = t2[i][0]; out[i * 4 + 1][e] = 0.5f * (t2[i][0] + t2[i][1]);
      out[i * 4 + 2][e] = 0.5f * (t2[i][0] - t2[i][1]); out[i * 4 + 3][e] = t2[i][1];
    }
  }
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) *reinterpret_cast<f32x4*>(Vy + ((((long)slab * 16 + pos) * 3 + nt) * 64 + lane) * 4) = out[pos];
}

struct WinoWgradArgs {
  const float* A; int lda; int C;                 // forward input Y (pre BatchNorm), Cb channels
  const float* a_scale; const float* a_shift;
  const float* Vy;                                // transformed output gradient, fragment order
  float* part;                                    // [split][16 pos][48 n][Cpad]   (Cpad = 64 * column blocks)
  int B, H, W, TH, TW, T, nslab, split, Cpad;
  unsigned a_bytes;
};

template <bool BNRELU>
__global__ __launch_bounds__(512, 2) void conv3x3_wino_wgrad_kernel(WinoWgradArgs p) {
  __shared__ __attribute__((aligned(1024))) float smem[2 * 16 * 16 * 64];        // Vd double buffer [pos][tile][64 c]: 2 x 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int per = (p.nslab + p.split - 1) / p.split;
  const int s_begin = blockIdx.y * per, s_end = min(p.nslab, s_begin + per);
  const int c0 = blockIdx.x * 64;
  const bool producer = wave >= 4;
  const int pos0 = (wave & 3) * 4;
  f32x4 acc[4][3][4];                               // [position][n-tile][c-tile]
#pragma unroll
  for (int pp = 0; pp < 4; ++pp)
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[pp][nt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (s_begin < s_end) {
    if (producer) {
      // ---- producer: thread -> channel quad cq (of the workgroup's 64 channels) of tile tl of the slab; 16 lanes cover a pixel's 256 bytes ----
      const int ptid = tid - 256, tl = ptid >> 4, cq = ptid & 15;
      const bool cok = c0 + cq * 4 < p.C;
      const __amdgpu_buffer_rsrc_t srdA = wsrd(p.A, p.a_bytes);
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (BNRELU && cok) { sc = *reinterpret_cast<const f32x4*>(p.a_scale + c0 + cq * 4); sh = *reinterpret_cast<const f32x4*>(p.a_shift + c0 + cq * 4); }
      const unsigned coff = (unsigned)((c0 + cq * 4) * 4);
      // tile cursor (b, ty, tx) of this thread's tile of the current slab; one slab = 16 tiles further
      int tb, tty, ttx;
      {
        const int t = s_begin * 16 + tl, tpi = p.TH * p.TW;
        tb = t / tpi;
        const int rem = t - tb * tpi;
        tty = rem / p.TW; ttx = rem - tty * p.TW;
      }
      unsigned voff[16];
      float hi[16];
      auto place = [&]() {                           // offsets / padding bounds of the cursor's patch, then advance the cursor by one slab
        const bool tok = tb < p.B && cok;
        const unsigned base = (unsigned)((tb * p.H + 2 * tty - 1) * p.W + 2 * ttx - 1) * (unsigned)(p.lda * 4) + coff;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = tok && (unsigned)(2 * tty - 1 + i) < (unsigned)p.H && (unsigned)(2 * ttx - 1 + j) < (unsigned)p.W;
            voff[i * 4 + j] = ok ? base + (unsigned)((i * p.W + j) * p.lda * 4) : WOOB;
            hi[i * 4 + j] = ok ? __builtin_inff() : 0.f;
          }
        ttx += 16;
        while (ttx >= p.TW) { ttx -= p.TW; ++tty; }
        while (tty >= p.TH) { tty -= p.TH; ++tb; }
      };
      float rv[4][16];
      auto load_raw = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srdA, (int)voff[q], 0, 0);
          rv[0][q] = __uint_as_float(v.x); rv[1][q] = __uint_as_float(v.y); rv[2][q] = __uint_as_float(v.z); rv[3][q] = __uint_as_float(v.w);
        }
      };
      float hic[16];                                 // padding bounds of the patch that is in flight (place() already describes the next one)
      auto transform_store = [&](float* Vb) {
        if (BNRELU) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) rv[c][q] = __builtin_amdgcn_fmed3f(fmaf(rv[c][q], sc[c], sh[c]), 0.f, hic[q]);
        }
#define RDM_WINO_1D(v, i0, i1, i2, i3)                                                        \
        { const float d0 = v[i0], d1 = v[i1], d2 = v[i2], d3 = v[i3];                         \
          v[i0] = d0 - d2; v[i1] = d1 + d2; v[i2] = d2 - d1; v[i3] = d1 - d3; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) RDM_WINO_1D(rv[c], j, 4 + j, 8 + j, 12 + j)
#pragma unroll
          for (int i = 0; i < 4; ++i) RDM_WINO_1D(rv[c], 4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3)
        }
#undef RDM_WINO_1D
        const int chunk = (cq >> 2) ^ (tl & 3);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos)
          *reinterpret_cast<f32x4*>(Vb + ((pos * 16 + tl) * 64 + chunk * 16 + (cq & 3) * 4)) = f32x4{rv[0][pos], rv[1][pos], rv[2][pos], rv[3][pos]};
      };
      place();
      load_raw();
#pragma unroll
      for (int q = 0; q < 16; ++q) hic[q] = hi[q];
      transform_store(smem);
      place();                                         // slab s_begin + 1 (past the end: harmless loads of real tiles or of nothing)
      load_raw();
#pragma unroll
      for (int q = 0; q < 16; ++q) hic[q] = hi[q];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        if (s + 1 < s_end) {
          transform_store(smem + (buf ^ 1) * (16 * 16 * 64));
          place();
          load_raw();
#pragma unroll
          for (int q = 0; q < 16; ++q) hic[q] = hi[q];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // ---- consumer: positions pos0 .. pos0 + 3; A = Vy fragments (global, 1 KiB each), B = Vd fragments (LDS, one float per k-step) ----
      const __amdgpu_buffer_rsrc_t srdY = wsrd(p.Vy, (unsigned)((size_t)p.nslab * 16 * 3 * 1024));
      const int lane16 = lane * 16;
      auto load_a = [&](int s, int pos, f32x4 (&a)[3]) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srdY, lane16, ((s * 16 + pos) * 3 + nt) * 1024, 0);
          a[nt] = f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        }
      };
      // B fragment of c-tile ct, k-step e: tile 4e + g, channel ct*16 + l16 -> chunk ct ^ (tile & 3) = ct ^ g
      auto load_b = [&](const float* Vb, int pos, int ct) {
        f32x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = Vb[(pos * 16 + 4 * e + g) * 64 + ((ct ^ g) * 16) + l16];
        return b;
      };
      f32x4 aq[2][3], bq[2];
      load_a(s_begin, pos0, aq[0]);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int s = s_begin; s < s_end; ++s) {
        const float* Vb = smem + ((s - s_begin) & 1) * (16 * 16 * 64);
        bq[0] = load_b(Vb, pos0, 0);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (pp < 3) load_a(s, pos0 + pp + 1, aq[(pp + 1) & 1]);
          else load_a(min(s + 1, s_end - 1), pos0, aq[0]);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const int cur = (pp * 4 + ct) & 1;
            if (ct < 3) bq[cur ^ 1] = load_b(Vb, pos0 + pp, ct + 1);
            else if (pp < 3) bq[cur ^ 1] = load_b(Vb, pos0 + pp + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int nt = 0; nt < 3; ++nt)
                acc[pp][nt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[pp & 1][nt][e], bq[cur][e], acc[pp][nt][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  }
  // ---- partial Q of this K split: plain stores, [split][pos][n][Cpad] (D layout: row n = nt*16 + 4g + r, column c = ct*16 + l16) ----
  if (!producer) {
    float* dst = p.part + (((long)blockIdx.y * 16 + pos0) * 48) * p.Cpad + c0;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp)
#pragma unroll
      for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) dst[((long)pp * 48 + nt * 16 + 4 * g + r) * p.Cpad + ct * 16 + l16] = acc[pp][nt][ct][r];
  }
}

// Q = sum of the split partials (fixed order), dW = A3^T Q A3 -> packed gradient [tap][n][c]; a thread owns 4 consecutive channels
__global__ __launch_bounds__(256) void k_wino_wgrad_reduce(const float* __restrict__ part, int split, int Cpad, int N, int C, float* __restrict__ dW, long wtap, int ldw) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int c4 = Cpad / 4;
  const int c = (int)(idx % c4) * 4, n = (int)(idx / c4);
  if (n >= N || c >= C) return;
  f32x4 q[16];
  const long stride = (long)16 * 48 * Cpad;
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) {
    const float* src = part + ((long)pos * 48 + n) * Cpad + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    int s = 1;
    for (; s + 3 < split; s += 4) {                                // four loads in flight, added in split order
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + s * stride), a1 = *reinterpret_cast<const f32x4*>(src + (s + 1) * stride);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (s + 2) * stride), a3 = *reinterpret_cast<const f32x4*>(src + (s + 3) * stride);
      v += a0; v += a1; v += a2; v += a3;
    }
    for (; s < split; ++s) v += *reinterpret_cast<const f32x4*>(src + s * stride);
    q[pos] = v;
  }
  f32x4 t[3][4];                                                  // A3^T Q: rows
#pragma unroll
  for (int j = 0; j < 4; ++j) { t[0][j] = q[j] + q[4 + j] + q[8 + j]; t[1][j] = q[4 + j] - q[8 + j]; t[2][j] = q[4 + j] + q[8 + j] - q[12 + j]; }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const f32x4 w3[3] = {t[r][0] + t[r][1] + t[r][2], t[r][1] - t[r][2], t[r][1] + t[r][2] - t[r][3]};
#pragma unroll
    for (int qq = 0; qq < 3; ++qq)
      *reinterpret_cast<f32x4*>(dW + (long)(r * 3 + qq) * wtap + (long)n * ldw + c) = w3[qq];      // C is a multiple of 4: a quad is inside or outside
  }
}

}  // namespace

size_t wino_u_bytes(int C, bool x6) { return x6 ? (size_t)(C / 16) * 16 * 3 * XU_PNT : (size_t)16 * 48 * C * sizeof(float); }

size_t wino_fwd_workspace_bytes(int C, long M, int split, bool x6) {
  const size_t u = wino_u_bytes(C, x6);
  const size_t part = split > 1 ? (size_t)split * M * 48 * sizeof(float) : 0;
  return ((u + 255) & ~(size_t)255) + ((part + 255) & ~(size_t)255);
}

int wino_pick_split(int T, int nslab, bool x6) {
  // one workgroup per CU (128 KB of LDS; 144 KB for the bf16x6 kernel): the grid runs in ceil(blocks / 256) rounds.  Every split pays ~1.2
  // slab-times of prologue and epilogue; pick the split with the smallest (rounds x (slabs per split + 1.2)).
  const int tt = x6 ? TX : TT;
  const long tb = (T + tt - 1) / tt;
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 32 && sp <= nslab / 4 + 1; ++sp) {
    const long rounds = (tb * sp + 255) / 256;
    const double cost = rounds * ((nslab + sp - 1) / sp + 1.2);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
  }
  return best;
}

int launch_wino_weight(const float* w, long wtap, int ldw, int N, int C, float* U, hipStream_t s, bool x6) {
  RDM_CHECK_ARG(C % 16 == 0 && N >= 1 && N <= 48, "winograd weights: C (%d) must be a multiple of 16 and N (%d) <= 48", C, N);
  const long threads = (long)(C / 16) * 3 * 64;
  if (x6) {
    hipLaunchKernelGGL(k_wino_weight_x6, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, s, w, wtap, ldw, N, C, reinterpret_cast<unsigned char*>(U));
    RDM_LAUNCH_OK();
    return 0;
  }
  hipLaunchKernelGGL(k_wino_weight, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, s, w, wtap, ldw, N, C, U);
  RDM_LAUNCH_OK();
  return 0;
}

// tail of both forward kernels: split > 1: ordered sum of the partials into the output slice (+ statistics); split == 1 with statistics: one
// pass over the slice.  Deterministic mode: the statistics come from the ordered column pass instead of the reduction's per-workgroup f64 atomics.
static int wino_fwd_finish(const WinoConv& a, int split, long M, hipStream_t s) {
  if (split > 1 || a.stat0) {
    const bool stats_apart = a.stat0 && (split == 1 || t_deterministic);
    if (split > 1) {
      const int rpb = 84;                                     // 21 row lanes x 4 rows
      hipLaunchKernelGGL(k_wino_reduce, dim3((unsigned)cdiv(M, rpb)), dim3(256), 0, s, a.partial, split, M, a.N, a.out, a.ldc, stats_apart ? nullptr : a.stat0,
                         stats_apart ? nullptr : a.stat1, rpb);
      RDM_LAUNCH_OK();
    }
    if (stats_apart)
      if (int rc = launch_colstats(a.out, a.ldc, (int)M, a.N, a.stat0, a.stat1, s)) return rc;
  }
  return 0;
}

int launch_conv3x3_wino_fwd(const WinoConv& a, hipStream_t s) {
  RDM_CHECK_ARG(a.C % 16 == 0 && a.C > 0 && a.N >= 1 && a.N <= 48, "winograd 3x3: C (%d) must be a multiple of 16, N (%d) <= 48", a.C, a.N);
  RDM_CHECK_ARG(a.lda % 4 == 0 && ((uintptr_t)a.A & 15) == 0, "winograd 3x3: input stride must be a multiple of 4 floats and the tensor 16-byte aligned");
  RDM_CHECK_ARG(a.H >= 1 && a.W >= 1 && a.B >= 1, "winograd 3x3: empty geometry");
  const long M = (long)a.B * a.H * a.W;
  const long ab = ((M - 1) * a.lda + a.C) * 4;
  if (ab >= 0xFFFFFFFFL) { set_error("winograd 3x3: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  const int TH = (a.H + 1) / 2, TW = (a.W + 1) / 2;
  const int T = a.B * TH * TW, nslab = a.C / 16;
  int split = a.split > 0 ? std::min(a.split, nslab) : wino_pick_split(T, nslab, a.x6 != 0);
  if (split > 1 && a.partial == nullptr) split = 1;
  if (split > 1 && a.partial_floats < (size_t)split * M * 48) split = std::max<long>(1, (long)(a.partial_floats / ((size_t)M * 48)));
  if (a.x6) {
    WinoX6Args k{};
    k.A = a.A; k.lda = a.lda; k.C = a.C; k.a_scale = a.a_scale; k.a_shift = a.a_shift; k.U = reinterpret_cast<const unsigned char*>(a.U);
    k.out = split > 1 ? a.partial : a.out; k.ldc = a.ldc; k.N = a.N;
    k.B = a.B; k.H = a.H; k.W = a.W; k.TH = TH; k.TW = TW; k.T = T; k.split = split; k.a_bytes = (unsigned)ab; k.u_bytes = (unsigned)wino_u_bytes(a.C, true);
#ifdef RDM_DEV_VARIANTS
    static const int abl_env = getenv("RDM_WX6_ABL") ? atoi(getenv("RDM_WX6_ABL")) : 0;
    k.abl = abl_env;
#endif
    void* prof = profile_begin(s, 2.0 * M * a.N * 9.0 * a.C, 18);
    RDM_CENSUS("conv3x3_wino_x6_kernel/%s/%s", a.a_scale ? "bn1" : "bn0", split > 1 ? "PARTIAL" : (a.stat0 ? "STORE+stats" : "STORE"));
    dim3 grid((unsigned)cdiv(T, TX), (unsigned)split);
    if (a.a_scale) {
      RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_x6_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * XIMG));
      hipLaunchKernelGGL((conv3x3_wino_x6_kernel<true>), grid, dim3(448), 2 * XIMG, s, k);
    } else {
      RDM_HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_x6_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * XIMG));
      hipLaunchKernelGGL((conv3x3_wino_x6_kernel<false>), grid, dim3(448), 2 * XIMG, s, k);
    }
    profile_end(prof, s);
    RDM_LAUNCH_OK();
    return wino_fwd_finish(a, split, M, s);
  }
  WinoFwdArgs k{};
  k.A = a.A; k.lda = a.lda; k.C = a.C; k.a_scale = a.a_scale; k.a_shift = a.a_shift; k.U = a.U;
  k.out = split > 1 ? a.partial : a.out; k.ldc = a.ldc; k.N = a.N;
  k.B = a.B; k.H = a.H; k.W = a.W; k.TH = TH; k.TW = TW; k.T = T; k.split = split; k.a_bytes = (unsigned)ab;
  void* prof = profile_begin(s, 2.0 * M * a.N * 9.0 * a.C, 9);
  RDM_CENSUS("conv3x3_wino_fwd_kernel/%s/%s", a.a_scale ? "bn1" : "bn0", split > 1 ? "PARTIAL" : (a.stat0 ? "STORE+stats" : "STORE"));
  dim3 grid((unsigned)cdiv(T, TT), (unsigned)split);
#ifdef RDM_DEV_VARIANTS
  const char* abl_env = getenv("RDM_WINO_ABL");
  const int abl = abl_env ? atoi(abl_env) : 0;
#define RDM_ABL(N_) if (abl == N_) hipLaunchKernelGGL((conv3x3_wino_fwd_kernel<true, N_>), grid, dim3(512), 0, s, k); else
  RDM_ABL(1) RDM_ABL(2) RDM_ABL(3) RDM_ABL(4) RDM_ABL(8) RDM_ABL(9) RDM_ABL(11) RDM_ABL(12) RDM_ABL(13) RDM_ABL(15)
#undef RDM_ABL
#endif
  if (a.a_scale) hipLaunchKernelGGL((conv3x3_wino_fwd_kernel<true>), grid, dim3(512), 0, s, k);
  else hipLaunchKernelGGL((conv3x3_wino_fwd_kernel<false>), grid, dim3(512), 0, s, k);
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  return wino_fwd_finish(a, split, M, s);
}

int wino_wgrad_split(int nslab, int cblocks) {
  // one workgroup per CU: (column blocks x splits) should fill the 256 CUs in whole rounds; every split keeps >= 16 slabs
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 64 && sp * 16 <= std::max(nslab, 16); ++sp) {
    const long rounds = ((long)cblocks * sp + 255) / 256;
    const double cost = rounds * ((nslab + sp - 1) / sp + 2.0) + 0.8 * sp;       // + the split's partial Q through HBM (~0.8 slab-times)
    if (cost < best_cost - 1e-9) { best_cost = cost; best = sp; }
  }
  return best;
}
size_t wino_wgrad_vy_floats(int B, int H, int W) {
  const long T = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  return (size_t)((T + 15) / 16) * 16 * 3 * 64 * 4;
}
size_t wino_wgrad_part_floats(int B, int H, int W, int C) {
  const long T = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  const int cblocks = (C + 63) / 64;
  return (size_t)wino_wgrad_split((int)((T + 15) / 16), cblocks) * 16 * 48 * cblocks * 64;
}

int launch_conv3x3_wino_wgrad(const WinoWgrad& a, hipStream_t s) {
  RDM_CHECK_ARG(a.N >= 1 && a.N <= 48 && a.N % 4 == 0 && a.C >= 4 && a.C % 4 == 0, "winograd 3x3 wgrad: N (%d) <= 48, N and C (%d) multiples of 4", a.N, a.C);
  RDM_CHECK_ARG(a.lda % 4 == 0 && a.ldg % 4 == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.G & 15) == 0, "winograd 3x3 wgrad: strides multiples of 4 floats, tensors 16-byte aligned");
  // every argument is validated BEFORE the first launch: a rejected call leaves no work in flight and no scratch half-written
  RDM_CHECK_ARG(a.dW != nullptr && a.ldw % 4 == 0 && a.wtap % 4 == 0 && ((uintptr_t)a.dW & 15) == 0, "winograd 3x3 wgrad: the packed gradient must be 16-byte aligned with strides that are multiples of 4");
  RDM_CHECK_ARG(a.Vy != nullptr && a.part != nullptr, "winograd 3x3 wgrad: scratch is NULL");
  const long M = (long)a.B * a.H * a.W;
  const long ab = ((M - 1) * a.lda + a.C) * 4, gb = ((M - 1) * a.ldg + a.N) * 4;
  if (ab >= 0xFFFFFFFFL || gb >= 0xFFFFFFFFL) { set_error("winograd 3x3 wgrad: operand extent >= 4 GiB is not supported by the 32-bit buffer addressing"); return RDM_ERR_UNSUPPORTED; }
  const int TH = (a.H + 1) / 2, TW = (a.W + 1) / 2;
  const int T = a.B * TH * TW, nslab = cdiv(T, 16), cblocks = cdiv(a.C, 64);
  const int split = wino_wgrad_split(nslab, cblocks);
  RDM_CHECK_ARG(a.vy_floats >= (size_t)nslab * 16 * 3 * 256 && a.part_floats >= (size_t)split * 16 * 48 * cblocks * 64, "winograd 3x3 wgrad: scratch too small");
  hipLaunchKernelGGL(k_wino_gradout_transform, dim3((unsigned)nslab), dim3(192), 0, s, a.G, a.ldg, a.N, (unsigned)gb, a.B, a.H, a.W, TH, TW, T, nslab, a.Vy);
  RDM_LAUNCH_OK();
  WinoWgradArgs k{};
  k.A = a.A; k.lda = a.lda; k.C = a.C; k.a_scale = a.a_scale; k.a_shift = a.a_shift; k.Vy = a.Vy; k.part = a.part;
  k.B = a.B; k.H = a.H; k.W = a.W; k.TH = TH; k.TW = TW; k.T = T; k.nslab = nslab; k.split = split; k.Cpad = cblocks * 64; k.a_bytes = (unsigned)ab;
  void* prof = profile_begin(s, 2.0 * M * a.N * 9.0 * a.C, 10);
  RDM_CENSUS("conv3x3_wino_wgrad_kernel/%s", a.a_scale ? "bn1" : "bn0");
  dim3 grid((unsigned)cblocks, (unsigned)split);
  if (a.a_scale) hipLaunchKernelGGL(conv3x3_wino_wgrad_kernel<true>, grid, dim3(512), 0, s, k);
  else hipLaunchKernelGGL(conv3x3_wino_wgrad_kernel<false>, grid, dim3(512), 0, s, k);
  profile_end(prof, s);
  RDM_LAUNCH_OK();
  hipLaunchKernelGGL(k_wino_wgrad_reduce, dim3((unsigned)cdiv((long)48 * cblocks * 16, 256)), dim3(256), 0, s, a.part, split, cblocks * 64, a.N, a.C, a.dW, a.wtap, a.ldw);
  RDM_LAUNCH_OK();
  return 0;
}

}  // namespace rdm
