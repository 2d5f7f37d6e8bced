// Attainable-peak microbenchmarks: a float4 stream copy for the HBM3E ceiling and a
// register-only v_mfma_f32_16x16x4_f32 loop for the fp32 matrix-core ceiling at the clock the chip
// actually holds.  They give the "measured-attainable" denominators next to the datasheet peaks.
#include <algorithm>

#include "rdm_common.h"

namespace rdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  RDM_CHECK_ARG(scratch && blocks > 0 && iters > 0, "microbench_mfma: bad argument");
  hipLaunchKernelGGL(k_mfma_loop, dim3(blocks), dim3(256), 0, stream, scratch, iters, 1.0f);
  RDM_LAUNCH_OK();
  return RDM_OK;
}

}  // extern "C"
