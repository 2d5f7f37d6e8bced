// Issue rate of v_mfma_f32_16x16x32_bf16 on MI355X as a function of the number of INDEPENDENT accumulators a wave cycles through (D = 1: every MFMA waits
// for the previous one's result) and of the waves per SIMD.  Operands in registers, no memory traffic; shader cycles per MFMA and wave from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_rate tools/mfma_rate/mfma_bf16_rate.hip && ./mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void k(float* out, long* cyc, int iters) {
  f32x4 acc[D];
#pragma unroll
  for (int d = 0; d < D; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  const long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / (D > 16 ? 16 : D); ++r)
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[d], 0, 0, 0);
  }
  const long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) s += acc[d][0] + acc[d][1] + acc[d][2] + acc[d][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int D>
void run(int waves_per_simd, float* out, long* cyc) {
  const int iters = 2000, threads = 256 * waves_per_simd;      // 4 SIMDs per CU
  hipLaunchKernelGGL(k<D>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<D>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long h[8]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const int per = (D >= 16 ? 16 : (16 / D) * D);
  const double n = (double)iters * per;
  printf("D = %2d independent accumulators, %d wave(s) per SIMD: %6.1f shader cycles per MFMA and wave, %7.1f TFLOP/s on the chip\n", D, waves_per_simd, (double)h[0] / n,
         256.0 * 4 * waves_per_simd * n * 16384.0 / (ms * 1e-3) / 1e12);
}

// The pattern of the split kernels: NA x NB operand fragments in registers (all distinct), every MFMA a different (a, b, accumulator) triple, six products per accumulator
// as in the bf16x6 forward - no memory traffic either.  Does feeding DIFFERENT registers cost issue cycles?
template <int NI, int NT>
__global__ __launch_bounds__(256) void k_pattern(float* out, long* cyc, int iters) {
  f32x4 acc[NI][NT];
  bf16x8 x0[NI], x1[NI], x2[NI], w0[NT], w1[NT], w2[NT];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) { x0[i][e] = (__bf16)(float)((threadIdx.x + i + e) & 3); x1[i][e] = (__bf16)0.5f; x2[i][e] = (__bf16)0.25f; }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) { w0[t][e] = (__bf16)(float)((threadIdx.x + t) & 1); w1[t][e] = (__bf16)0.125f; w2[t][e] = (__bf16)1.0f; }
  const long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#define TERM(W_, X_) _Pragma("unroll") for (int i = 0; i < NI; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_[t], X_[i], acc[i][t], 0, 0, 0)
      TERM(w1, x1); TERM(w2, x0); TERM(w0, x2); TERM(w1, x0); TERM(w0, x1); TERM(w0, x0);
#undef TERM
    }
    asm volatile("" ::: "memory");
  }
  const long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) s += acc[i][t][0] + acc[i][t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NI, int NT>
void run_pattern(float* out, long* cyc) {
  const int iters = 500;
  hipLaunchKernelGGL((k_pattern<NI, NT>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess) { printf("pattern %d x %d: %s\n", NI, NT, hipGetErrorString(e)); return; }
  long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  fflush(stdout); printf("tile pattern %d x %d accumulators, six products each, distinct operand registers, 1 wave per SIMD: %6.1f shader cycles per MFMA\n", NI, NT, (double)h[0] / ((double)iters * NI * NT * 6));
}

int main() {
  float* out; long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  for (int w = 1; w <= 2; ++w) { run<1>(w, out, cyc); run<2>(w, out, cyc); run<4>(w, out, cyc); run<8>(w, out, cyc); run<16>(w, out, cyc); }
  run_pattern<4, 6>(out, cyc); run_pattern<4, 10>(out, cyc); run_pattern<2, 6>(out, cyc);
  return 0;
}
