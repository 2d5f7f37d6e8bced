// Attainable-peak microbenchmarks: a float4 stream copy for the HBM3E ceiling and a
// register-only v_mfma_f32_16x16x4_f32 loop for the fp32 matrix-core ceiling at the clock the chip
// actually holds.  They give the "measured-attainable" denominators next to the datasheet peaks.
#include <algorithm>

#include "../rdm_common.h"
#include "../../../include/rdm_bench.h"

namespace rdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// (the host pass of hipcc cannot type-check the gfx950-only builtin and silently drops the kernel's host stub if it sees it)
// XCD-aware block order (igemm.hip xcd_block_order), optionally in row groups of G tiles walked column by column
__device__ __forceinline__ void tile_order(int G, int& bx, int& by) {
  const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy, L = blockIdx.x + gx * blockIdx.y;
  const unsigned x = L & 7u, seq = L >> 3, q = total >> 3, r = total & 7u;
  const unsigned Lp = x * q + (x < r ? x : r) + seq;
  if (G >= 2) {
    const unsigned gsz = (unsigned)G * gx, grp = Lp / gsz, rem = Lp - grp * gsz, rows = min((unsigned)G, gy - grp * (unsigned)G), c = rem / rows;
    bx = (int)c; by = (int)(grp * G + (rem - c * rows));
    return;
  }
  bx = (int)(Lp % gx); by = (int)(Lp / gx);
}
__device__ __forceinline__ void lds_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, (int)soff, 0, 0);
#endif
}
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, lds_dst, 16, 0, 0);
#endif
}


__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dst[i] = src[i];
}

// 4 waves per workgroup, 12 independent accumulators per wave (the conv kernels' wave tile), no memory traffic
__global__ __launch_bounds__(256) void k_mfma_loop(float* out, int iters, float seed) {
  f32x4 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[blockIdx.x] = s;     // keep the loop alive without a store in the common case
}

// ---------------------------------------------------------------------------------------------
// EXPERIMENT (round 2): an f32 GEMM  C[m][n] = sum_k A[m][k] * W[n][k]  (both operands k-contiguous, the 1x1 forward's shape) whose
// operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, no staging VALU) as a lane-linear
// [row][16 floats] image, and whose fragments are ONE ds_read_b128 per 16-row tile and 16-deep slab: lane (row, kq) takes
// A[row][4kq .. 4kq+3] and feeds element e to MFMA step e - a permutation of k that A and W share, so the contraction is unchanged.
// 64 lanes x 16 B of such a read cover 1 KiB contiguously: no bank conflicts without padding or swizzling.
// Per 16-deep slab and wave: 7 ds_read_b128 + 3-4 DMA issues + 1 barrier next to 48 MFMAs (conv_fwd_kernel: ~150 other instructions).
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256, 4) void k_gemm_dma_f32(const float* A, int lda, const float* Wt, int ldw, float* Cc,
                                                         int ldc, int M, int N, int K, int grp) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN, BK = 16;
  constexpr int AI = BM / 16, BI = BN / 16;                // 1-KiB pieces (16 rows x 64 B) per slab: one wave-instruction each
  // ONE __shared__ object: with two, hipcc waits vmcnt(0) in front of the first ds_read of every slab (cdna_hip_programming.md, 5, trap 4a)
  __shared__ __attribute__((aligned(1024))) float smem[NBUF * (BM + BN) * BK];
  float* const sm = smem;
  auto As = [&](int buf) { return sm + buf * (BM + BN) * BK; };
  auto Bs = [&](int buf) { return sm + buf * (BM + BN) * BK + BM * BK; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave / WN) * MT * 16, wcol = (wave % WN) * NT * 16;
  // XCD-aware order as in igemm.hip
  int bx, by;
  tile_order(grp, bx, by);
  const int n0 = bx * BN, m0 = by * BM;
  // DMA source pointers of this lane: piece p covers rows 16p .. 16p+15, lane -> (row 16p + lane/4, 16-byte chunk lane%4)
  // 16-byte chunk swizzle: the LDS image is lane-linear (DMA), so the SOURCE chunk is permuted - slot (row, c) holds logical chunk
  // c ^ ((row >> 2) & 3); a 16-lane group of a ds_read_b128 (rows 0..15, one logical chunk) then touches 16 distinct 16-byte slots
  // of the 256-byte bank row instead of 4 (4-way conflict)
  const int prow = lane >> 2, pch = (lane & 3) ^ ((prow >> 2) & 3);
  // buffer form (SRD + one 32-bit offset register per piece, the slab advance in a scalar offset): the flat form's 64-bit address
  // arithmetic per DMA cost 28 % of the MFMA rate in k_mfma_loop_tile (112 vs 144 TFLOP/s)
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)(((long)(M - 1) * lda + K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, (int)(((long)(N - 1) * ldw + K) * 4), 0x00020000);
  unsigned aoff[(AI + 3) / 4], boff[(BI + 3) / 4];
#pragma unroll
  for (int t = 0; t < (AI + 3) / 4; ++t) aoff[t] = (unsigned)(min(m0 + (wave + 4 * t) * 16 + prow, M - 1) * lda + pch * 4) * 4u;
#pragma unroll
  for (int t = 0; t < (BI + 3) / 4; ++t) boff[t] = (unsigned)(min(n0 + (wave + 4 * t) * 16 + prow, N - 1) * ldw + pch * 4) * 4u;
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int t = 0; t < (AI + 3) / 4; ++t) {
      const int piece = wave + 4 * t;
      if (piece < AI) lds_dma16_buf(ra, aoff[t], (unsigned)k0 * 4u, As(buf) + piece * 256);
    }
#pragma unroll
    for (int t = 0; t < (BI + 3) / 4; ++t) {
      const int piece = wave + 4 * t;
      if (piece < BI) lds_dma16_buf(rw, boff[t], (unsigned)k0 * 4u, Bs(buf) + piece * 256);
    }
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
    if (b < nk) issue(b, b * BK);
  for (int s = 0; s < nk; ++s) {
    // slab s must have landed: all but the (NBUF - 2) youngest groups of this wave's DMAs, then the workgroup
    // exact per-wave DMA counts per slab: waves with a piece in the ragged last round issue one more
    constexpr int NA = AI / 4, NB = BI / 4;                           // AI, BI: multiples of 4 except BI = 6 (waves 0, 1 issue NB + 1)
    if (NBUF == 2 || s + NBUF - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (tail: nothing younger than slab s is in flight)
    else if (wave < BI % 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB + 1) * (NBUF - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB) * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NBUF - 1 < nk) issue((s + NBUF - 1) % NBUF, (s + NBUF - 1) * BK);     // into the buffer slab s-1 was read from: every wave is past it
    const int buf = s % NBUF;
    f32x4 a4[MT], b4[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a4[i] = *reinterpret_cast<const f32x4*>(As(buf) + (wrow + i * 16 + l16) * BK + ((g ^ (l16 >> 2)) & 3) * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Bs(buf) + (wcol + j * 16 + l16) * BK + ((g ^ (l16 >> 2)) & 3) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + l16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow + i * 16 + g * 4 + r;
        if (n < N && m < M) Cc[(long)m * ldc + n] = acc[i][j][r];
      }
  }
}

// EXPERIMENT 1c: ISSUING the staging loads is what costs MFMA throughput (k_mfma_loop_tile: 152 TFLOP/s with fragment reads and a barrier
// per slab, 112 with four LDS-DMA instructions per wave and slab on top - waited for or not - 124 with two), so this variant stages half
// as many bytes per MFMA: wave tile 128 x 96 (MT = 8, NT = 6: 192 accumulator registers, 2 waves per SIMD), workgroup tile 256 x 192,
// 7 DMA instructions per wave and slab next to 192 MFMAs instead of 3.5 next to 48.
template <int NBUF, int NT, int MINB>
__global__ __launch_bounds__(256, MINB) void k_gemm_dma_big_f32(const float* A, int lda, const float* Wt, int ldw, float* Cc, int ldc, int M, int N, int K, int grp) {
  constexpr int MT = 8, BM = 256, BN = NT * 32, BK = 16, NA = 4, NB = BN / 64;
  __shared__ __attribute__((aligned(1024))) float smem[NBUF * (BM + BN) * BK];
  float* const sm = smem;
  auto As = [&](int buf) { return sm + buf * (BM + BN) * BK; };
  auto Bs = [&](int buf) { return sm + buf * (BM + BN) * BK + BM * BK; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave >> 1) * 128, wcol = (wave & 1) * (NT * 16);
  int bx, by;
  tile_order(grp, bx, by);
  const int n0 = bx * BN, m0 = by * BM;
  const int prow = lane >> 2, pch = (lane & 3) ^ ((prow >> 2) & 3);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)(((long)(M - 1) * lda + K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, (int)(((long)(N - 1) * ldw + K) * 4), 0x00020000);
  unsigned aoff[NA], boff[NB];
#pragma unroll
  for (int t = 0; t < NA; ++t) aoff[t] = (unsigned)(min(m0 + (wave + 4 * t) * 16 + prow, M - 1) * lda + pch * 4) * 4u;
#pragma unroll
  for (int t = 0; t < NB; ++t) boff[t] = (unsigned)(min(n0 + (wave + 4 * t) * 16 + prow, N - 1) * ldw + pch * 4) * 4u;
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int t = 0; t < NA; ++t) lds_dma16_buf(ra, aoff[t], (unsigned)k0 * 4u, As(buf) + (wave + 4 * t) * 256);
#pragma unroll
    for (int t = 0; t < NB; ++t) lds_dma16_buf(rw, boff[t], (unsigned)k0 * 4u, Bs(buf) + (wave + 4 * t) * 256);
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
    if (b < nk) issue(b, b * BK);
  const int sl = ((g ^ (l16 >> 2)) & 3) * 4;
  for (int s = 0; s < nk; ++s) {
    if (NBUF == 2 || s + NBUF - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB) * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NBUF - 1 < nk) issue((s + NBUF - 1) % NBUF, (s + NBUF - 1) * BK);
    const int buf = s % NBUF;
    f32x4 b4[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Bs(buf) + (wcol + j * 16 + l16) * BK + sl);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                        // the A fragments in two halves: 16 + 24 fragment registers beside 192 accumulators
      f32x4 a4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const f32x4*>(As(buf) + (wrow + (h * 4 + i) * 16 + l16) * BK + sl);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[h * 4 + i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + l16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow + i * 16 + g * 4 + r;
        if (n < N && m < M) Cc[(long)m * ldc + n] = acc[i][j][r];
      }
  }
}

// EXPERIMENT 1b: the same with 32-deep slabs - every staged row is a FULL 128-byte line (the 16-deep slab's 64-byte half lines cost the
// texture addresser twice the work per byte), half as many barriers per MFMA.  Lane (row, kq) reads chunks kq and kq + 4 of its row.
template <int MT, int NT, int WM, int WN, int NBUF>
__global__ __launch_bounds__(256, 2) void k_gemm_dma32_f32(const float* A, int lda, const float* Wt, int ldw, float* Cc, int ldc, int M, int N, int K, int grp) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN, BK = 32;
  constexpr int AI = BM / 8, BI = BN / 8;                  // 1-KiB pieces (8 rows x 128 B)
  static_assert(AI % 4 == 0 && BI % 4 == 0, "pieces must divide evenly over the 4 waves");
  constexpr int NA = AI / 4, NB = BI / 4;
  __shared__ __attribute__((aligned(1024))) float smem[NBUF * (BM + BN) * BK];
  float* const sm = smem;
  auto As = [&](int buf) { return sm + buf * (BM + BN) * BK; };
  auto Bs = [&](int buf) { return sm + buf * (BM + BN) * BK + BM * BK; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave / WN) * MT * 16, wcol = (wave % WN) * NT * 16;
  int bx, by;
  tile_order(grp, bx, by);
  const int n0 = bx * BN, m0 = by * BM;
  // piece p = rows 8p .. 8p+7; lane -> (row 8p + lane/8, slot lane%8); slot c of row r holds logical chunk c ^ ((r >> 1) & 7)
  const int prow = lane >> 3;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)(((long)(M - 1) * lda + K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, (int)(((long)(N - 1) * ldw + K) * 4), 0x00020000);
  unsigned aoff[NA], boff[NB];
#pragma unroll
  for (int t = 0; t < NA; ++t) {
    const int r = (wave + 4 * t) * 8 + prow;
    aoff[t] = (unsigned)(min(m0 + r, M - 1) * lda + (((lane & 7) ^ ((r >> 1) & 7)) * 4)) * 4u;
  }
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    const int r = (wave + 4 * t) * 8 + prow;
    boff[t] = (unsigned)(min(n0 + r, N - 1) * ldw + (((lane & 7) ^ ((r >> 1) & 7)) * 4)) * 4u;
  }
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int t = 0; t < NA; ++t) lds_dma16_buf(ra, aoff[t], (unsigned)k0 * 4u, As(buf) + (wave + 4 * t) * 256);
#pragma unroll
    for (int t = 0; t < NB; ++t) lds_dma16_buf(rw, boff[t], (unsigned)k0 * 4u, Bs(buf) + (wave + 4 * t) * 256);
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;                                   // (K a multiple of 32 in this experiment)
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
    if (b < nk) issue(b, b * BK);
  // this lane's two 16-byte slots of row l16 (+16 i): logical chunks g and g + 4, XOR-swizzled by (row >> 1) & 7 = (l16 >> 1)
  const int sl0 = ((g ^ (l16 >> 1)) & 7) * 4, sl1 = (((g + 4) ^ (l16 >> 1)) & 7) * 4;
  for (int s = 0; s < nk; ++s) {
    if (NBUF == 2 || s + NBUF - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB) * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NBUF - 1 < nk) issue((s + NBUF - 1) % NBUF, (s + NBUF - 1) * BK);
    const int buf = s % NBUF;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 a4[MT], b4[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a4[i] = *reinterpret_cast<const f32x4*>(As(buf) + (wrow + i * 16 + l16) * BK + (h ? sl1 : sl0));
#pragma unroll
      for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Bs(buf) + (wcol + j * 16 + l16) * BK + (h ? sl1 : sl0));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + l16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow + i * 16 + g * 4 + r;
        if (n < N && m < M && (ldc > 0 || acc[i][j][r] == 123.456f)) Cc[(long)m * (ldc > 0 ? ldc : -ldc) + n] = acc[i][j][r];   // ldc < 0: timing probe without the output stores
      }
  }
}

// EXPERIMENT 1d: k_gemm_dma32_f32 on PRE-TILED operands - A as [m-tile][k-slab][128 rows][32 floats], W as [n-tile][k-slab][96][32], each
// (tile, slab) image contiguous (16 / 12 KB) and already chunk-swizzled - so every DMA wave-instruction copies 1 KiB of CONSECUTIVE bytes
// instead of 8 row pieces 128 bytes long and a row stride apart.  Same LDS image, same fragment reads, same MFMA loop: what is the
// access SHAPE of the staging loads worth?
template <int NBUF>
__global__ __launch_bounds__(256, 2) void k_gemm_dma32_tiled_f32(const float* At, const float* Wt, float* Cc, int ldc, int M, int N, int K, int grp) {
  constexpr int MT = 4, NT = 3, BM = 128, BN = 96, BK = 32, NA = 4, NB = 3;
  __shared__ __attribute__((aligned(1024))) float smem[NBUF * (BM + BN) * BK];
  float* const sm = smem;
  auto As = [&](int buf) { return sm + buf * (BM + BN) * BK; };
  auto Bs = [&](int buf) { return sm + buf * (BM + BN) * BK + BM * BK; };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave >> 1) * MT * 16, wcol = (wave & 1) * NT * 16;
  int bx, by;
  tile_order(grp, bx, by);
  const int n0 = bx * BN, m0 = by * BM;
  const int nk = K / BK;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(At), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned abase = (unsigned)by * (unsigned)nk * (BM * BK * 4u), wbase = (unsigned)bx * (unsigned)nk * (BN * BK * 4u);
  auto issue = [&](int buf, int s) {
#pragma unroll
    for (int t = 0; t < NA; ++t) lds_dma16_buf(ra, (unsigned)((wave + 4 * t) * 1024 + lane * 16), abase + (unsigned)s * (BM * BK * 4u), As(buf) + (wave + 4 * t) * 256);
#pragma unroll
    for (int t = 0; t < NB; ++t) lds_dma16_buf(rw, (unsigned)((wave + 4 * t) * 1024 + lane * 16), wbase + (unsigned)s * (BN * BK * 4u), Bs(buf) + (wave + 4 * t) * 256);
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < NBUF - 1; ++b)
    if (b < nk) issue(b, b);
  const int sl0 = ((g ^ (l16 >> 1)) & 7) * 4, sl1 = (((g + 4) ^ (l16 >> 1)) & 7) * 4;
  for (int s = 0; s < nk; ++s) {
    if (NBUF == 2 || s + NBUF - 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA + NB) * (NBUF - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NBUF - 1 < nk) issue((s + NBUF - 1) % NBUF, s + NBUF - 1);
    const int buf = s % NBUF;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 a4[MT], b4[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a4[i] = *reinterpret_cast<const f32x4*>(As(buf) + (wrow + i * 16 + l16) * BK + (h ? sl1 : sl0));
#pragma unroll
      for (int j = 0; j < NT; ++j) b4[j] = *reinterpret_cast<const f32x4*>(Bs(buf) + (wcol + j * 16 + l16) * BK + (h ? sl1 : sl0));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + l16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow + i * 16 + g * 4 + r;
        if (n < N && m < M) Cc[(long)m * ldc + n] = acc[i][j][r];
      }
  }
}

// EXPERIMENT 2: no LDS and no workgroup barrier at all - every wave loads its own MFMA fragments straight from global memory
// (lane (row, kq) <- 16 bytes A[row][4kq .. 4kq+3], the same k permutation) NS slabs ahead into registers.  The A rows are loaded by
// the WN waves that share them and the W rows by the WM waves (through L1 / L2); what it buys is waves that never wait for each other.
template <int MT, int NT, int WM, int WN, int NS>
__global__ __launch_bounds__(256, 4) void k_gemm_direct_f32(const float* A, int lda, const float* Wt, int ldw, float* Cc, int ldc, int M, int N, int K, int grp) {
  constexpr int BM = MT * 16 * WM, BN = NT * 16 * WN, BK = 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wrow = (wave / WN) * MT * 16, wcol = (wave % WN) * NT * 16;
  int bx, by;
  tile_order(grp, bx, by);
  const int n0 = bx * BN, m0 = by * BM;
  const float* ap[MT];
  const float* bp[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) ap[i] = A + (long)min(m0 + wrow + i * 16 + l16, M - 1) * lda + g * 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) bp[j] = Wt + (long)min(n0 + wcol + j * 16 + l16, N - 1) * ldw + g * 4;
  f32x4 fa[NS][MT], fb[NS][NT];
  auto load = [&](int st, int k0) {
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[st][i] = *reinterpret_cast<const f32x4*>(ap[i] + k0);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[st][j] = *reinterpret_cast<const f32x4*>(bp[j] + k0);
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) load(st, st * BK);
  for (int s0 = 0; s0 < nk; s0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int s = s0 + u;
      if (s < nk) {
        if (s + NS - 1 < nk) load((u + NS - 1) % NS, (s + NS - 1) * BK);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][i][e], fb[u][j][e], acc[i][j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wcol + j * 16 + l16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow + i * 16 + g * 4 + r;
        if (n < N && m < M) Cc[(long)m * ldc + n] = acc[i][j][r];
      }
  }
}

int launch_gemm_dma(const float* a, int lda, const float* w, int ldw, float* c, int ldc, int m, int n, int k, int variant, hipStream_t stream) {
  const int grp = variant / 100;                    // variant = 100 * row-group size + kernel id
  variant %= 100;
  dim3 grid(cdiv(n, 96), cdiv(m, 128));
  if (variant == 50) hipLaunchKernelGGL((k_gemm_dma32_f32<4, 3, 2, 2, 2>), grid, dim3(256), 0, stream, a, lda, w, ldw, c, -ldc, m, n, k, grp);   // no output stores
  else if (variant == 40) hipLaunchKernelGGL(k_gemm_dma32_tiled_f32<2>, grid, dim3(256), 0, stream, a, w, c, ldc, m, n, k, grp);     // a, w: PRE-TILED images (lda / ldw unused)
  else if (variant == 30) hipLaunchKernelGGL((k_gemm_dma_big_f32<2, 6, 2>), dim3(cdiv(n, 192), cdiv(m, 256)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else if (variant == 31) hipLaunchKernelGGL((k_gemm_dma_big_f32<3, 6, 2>), dim3(cdiv(n, 192), cdiv(m, 256)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else if (variant == 32) hipLaunchKernelGGL((k_gemm_dma_big_f32<4, 4, 1>), dim3(cdiv(n, 128), cdiv(m, 256)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);   // 256 x 128, 4 buffers, ONE workgroup per CU
  else if (variant == 33) hipLaunchKernelGGL((k_gemm_dma_big_f32<3, 4, 2>), dim3(cdiv(n, 128), cdiv(m, 256)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);   // 256 x 128, 3 buffers, two per CU
  else if (variant == 34) hipLaunchKernelGGL((k_gemm_dma_big_f32<6, 4, 1>), dim3(cdiv(n, 128), cdiv(m, 256)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);   // 6 buffers, one per CU
  else if (variant == 20) hipLaunchKernelGGL((k_gemm_dma32_f32<4, 3, 2, 2, 2>), grid, dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else if (variant == 21) hipLaunchKernelGGL((k_gemm_dma32_f32<2, 3, 2, 2, 3>), dim3(cdiv(n, 96), cdiv(m, 64)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else if (variant == 22) hipLaunchKernelGGL((k_gemm_dma32_f32<2, 3, 2, 2, 2>), dim3(cdiv(n, 96), cdiv(m, 64)), dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else if (variant == 10) hipLaunchKernelGGL((k_gemm_direct_f32<4, 3, 2, 2, 2>), grid, dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else if (variant == 3) hipLaunchKernelGGL((k_gemm_dma_f32<4, 3, 2, 2, 3>), grid, dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  else hipLaunchKernelGGL((k_gemm_dma_f32<4, 3, 2, 2, 2>), grid, dim3(256), 0, stream, a, lda, w, ldw, c, ldc, m, n, k, grp);
  RDM_LAUNCH_OK();
  return RDM_OK;
}

// register-only MFMA loop in the conv kernels' exact operand pattern: per 16-deep slab 4 k-steps x (4 A x 3 B) MFMAs, 16 + 12 DISTINCT
// operand registers (the plain k_mfma_loop feeds every MFMA the same two registers).  No memory traffic.
// MODE 0: registers only.  1: + the slab's 7 ds_read_b128 fragment reads (static LDS image, conflict-free).  2: + one workgroup barrier
// per slab.  3: + the slab's LDS-DMA staging (4 x global_load_lds_dwordx4 per wave from an L2-resident 64 KB source) with the 2-buffer
// wait/barrier of k_gemm_dma_f32.  Each step adds exactly one ingredient of the real kernels to the 48-MFMA slab.
template <int MODE>
__global__ __launch_bounds__(256, 4) void k_mfma_loop_tile(float* out, int iters, float seed, const float* src, unsigned span_mask) {
  __shared__ __attribute__((aligned(1024))) float smem[2 * (128 + 96) * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x; i < 2 * (128 + 96) * 16; i += 256) smem[i] = seed + i * 1e-6f;
  __syncthreads();
  f32x4 acc[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 a[4], b[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = f32x4{seed + i, seed - i, seed * i, seed + 2 * i} + threadIdx.x * 1e-3f;
#pragma unroll
  for (int j = 0; j < 3; ++j) b[j] = f32x4{seed - j, seed + j, seed * 2 * j, seed - 3 * j} - threadIdx.x * 1e-3f;
  const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 48;
  const int sl = ((g ^ (l16 >> 2)) & 3) * 4;
  const float* gsrc = src + (size_t)(blockIdx.x & 3) * 4096 + wave * 1024 + lane * 4;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    float* base = smem + buf * (128 + 96) * 16;
    if (MODE >= 3) {
      if (MODE == 6) {                                                          // 6: buffer form: SRD + one 32-bit offset VGPR instead of a 64-bit address pair
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float* nb6 = smem + (buf ^ 1) * (128 + 96) * 16;
        // source: a window of (span_mask + 1) floats walked in 16 KB steps, a different phase per workgroup - span 64 KB: L1-resident,
        // a few MB: L2-resident, hundreds of MB: Infinity Cache / HBM (every step reads 14 KB like a real 128 x 96 x 16 slab)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7FFFFFFF, 0x00020000);
        const unsigned vo = (unsigned)((wave * 1024 + lane * 4) * 4);
        const unsigned so = (((unsigned)blockIdx.x * 40503u + (unsigned)it) * 4096u & span_mask) * 4u;
#pragma unroll
        for (int t = 0; t < 2; ++t) lds_dma16_buf(rs, vo, so + (unsigned)(t * 256 * 4), nb6 + (wave + 4 * t) * 256);
        lds_dma16_buf(rs, vo, so + 128 * 4, nb6 + 128 * 16 + wave * 256);
        if (wave < 2) lds_dma16_buf(rs, vo, so + 64 * 4, nb6 + 128 * 16 + (4 + wave) * 256);
      } else {
      if (MODE != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // 5: never wait for the DMA (timing only: what is ISSUING it worth?)
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      float* nb = smem + (buf ^ 1) * (128 + 96) * 16;
#pragma unroll
      for (int t = 0; t < 2; ++t) lds_dma16(gsrc + t * 256 + (it & 7) * 512, nb + (wave + 4 * t) * 256);
      if (MODE != 4) {                                                          // 4: the A tile only (8 of the 14 KB)
        lds_dma16(gsrc + 128, nb + 128 * 16 + wave * 256);
        if (wave < 2) lds_dma16(gsrc + 64, nb + 128 * 16 + (4 + wave) * 256);
      }
      }
    } else if (MODE == 2) {
      __builtin_amdgcn_s_barrier();
    }
    if (MODE >= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(base + (wrow + i * 16 + l16) * 16 + sl);
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = *reinterpret_cast<const f32x4*>(base + 128 * 16 + (wcol + j * 16 + l16) * 16 + sl);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    if (MODE == 0) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));     // operands "change" every slab
  }
  float sacc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (sacc == 123.456f) out[blockIdx.x] = sacc;
}

int launch_mfma_tile(float* scratch, int blocks, int iters, hipStream_t stream, int mode_in = -1, unsigned span_mask = 16383u) {
  const int mode = mode_in >= 0 ? mode_in : (iters & 7);       // (legacy: low three bits of the iteration count select the ingredient set)
  const float* src = scratch + 4096;                // the scratch buffer behind the first 16 KB is the DMA source
  if (mode == 0) hipLaunchKernelGGL(k_mfma_loop_tile<0>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  else if (mode == 1) hipLaunchKernelGGL(k_mfma_loop_tile<1>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  else if (mode == 2) hipLaunchKernelGGL(k_mfma_loop_tile<2>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  else if (mode == 3) hipLaunchKernelGGL(k_mfma_loop_tile<3>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  else if (mode == 4) hipLaunchKernelGGL(k_mfma_loop_tile<4>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  else if (mode == 5) hipLaunchKernelGGL(k_mfma_loop_tile<5>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  else hipLaunchKernelGGL(k_mfma_loop_tile<6>, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f, src, span_mask);
  RDM_LAUNCH_OK();
  return RDM_OK;
}

}  // namespace rdm

using namespace rdm;

extern "C" {

int rdm_microbench_copy(const float* src, float* dst, int64_t n_floats, rdm_stream_t stream) {
  RDM_CHECK_ARG(src && dst && n_floats > 0 && n_floats % 4 == 0, "microbench_copy: need a positive multiple of 4 floats");
  hipLaunchKernelGGL(k_stream_copy, dim3(256 * 8), dim3(256), 0, stream, reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), (long)(n_floats / 4));
  RDM_LAUNCH_OK();
  return RDM_OK;
}

/* launches `blocks` workgroups x 4 waves x iters x 12 MFMAs; FLOPs = blocks*4*iters*12*2048 */
int rdm_microbench_mfma_f32(float* scratch, int32_t blocks, int32_t iters, rdm_stream_t stream) {
  RDM_CHECK_ARG(scratch && blocks > 0 && iters != 0, "microbench_mfma: bad argument");
  if (iters < 0) return launch_mfma_tile(scratch, blocks, -iters / 4, stream);      // negative: the tile-pattern loop (48 MFMAs per iteration)
  hipLaunchKernelGGL(k_mfma_loop, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f);
  RDM_LAUNCH_OK();
  return RDM_OK;
}

/* the tile-pattern MFMA loop with one staging ingredient (mode 0..6, k_mfma_loop_tile); mode 6 reads its DMA source from a window of
 * span_floats (a power of two, <= scratch_floats - 8192) of the scratch buffer: where the data comes from (L1 / L2 / Infinity Cache / HBM) */
int rdm_microbench_mfma_staged_f32(float* scratch, int64_t scratch_floats, int32_t blocks, int32_t slabs, int32_t mode, int64_t span_floats,
                                   rdm_stream_t stream) {
  RDM_CHECK_ARG(scratch && blocks > 0 && slabs > 0 && mode >= 0 && mode <= 6, "microbench_mfma_staged: bad argument");
  RDM_CHECK_ARG(span_floats >= 16384 && (span_floats & (span_floats - 1)) == 0 && span_floats + 8192 <= scratch_floats && span_floats <= (1LL << 29),
                "microbench_mfma_staged: span must be a power of two of floats inside the scratch buffer");
  return launch_mfma_tile(scratch, blocks, slabs, stream, mode, (unsigned)(span_floats - 1));
}

/* EXPERIMENT: LDS-DMA staged f32 GEMM C[M][N] = A[M][K] * W[N][K]^T (K multiple of 16); variant = LDS buffers (2 or 3) */
int rdm_microbench_gemm_dma_f32(const float* a, int32_t lda, const float* w, int32_t ldw, float* c, int32_t ldc, int32_t m, int32_t n, int32_t k,
                                int32_t variant, rdm_stream_t stream) {
  RDM_CHECK_ARG(a && w && c && m > 0 && n > 0 && k > 0 && k % 16 == 0 && lda % 4 == 0 && ldw % 4 == 0, "microbench_gemm_dma: bad argument");
  return launch_gemm_dma(a, lda, w, ldw, c, ldc, m, n, k, variant, stream);
}

}  // extern "C"
