// Measurement kernel (librdm_bench.so, NOT the product): the price of a PER-XCD barrier among the workgroups that really share an XCD, and
// whether a same-XCD hand-off through the XCD's L2 (plain stores, s_waitcnt vmcnt(0), counter, sc1 loads) reads fresh data - the two numbers
// the review's "one image per XCD" persistent kernel for the few-pixel blocks of the bf16 forward stands on.
//
//  * grouping is by the hardware's own XCC_ID register, not by blockIdx % 8: every workgroup reads it, takes a rank in its XCD with an
//    agent-scope atomic and waits (bounded) until the whole grid has registered, so "same XCD" is a checked fact, not a placement guess;
//  * a round = every workgroup writes `payload_floats` floats of its slot (value = round number), drains its stores, one lane adds to the
//    XCD's counter and polls it with sc1 loads until all n_k members of the XCD have arrived; then the workgroup reads the slot of its
//    right-hand neighbour IN THE SAME XCD with sc1 loads (L1 bypassed, L2 served) and counts values that are not the round's;
//  * every spin is bounded (an exhausted budget sets a flag and the workgroup leaves every later wait at once): the grid always drains.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/rdm_bench.h"

namespace {

constexpr int kMaxPolls = 1 << 20;        // x ~0.1-0.2 us per poll: far beyond any real wait

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// state (u32): [0..7] workgroups registered per XCC | [8] total registered | [9] timeout flag | [32 * (1 + k)] barrier counter of XCC k (own 128-B line)
__global__ __launch_bounds__(256) void k_xcd_sync(unsigned* state, float* slots, int payload_floats, int rounds, unsigned* result) {
  __shared__ unsigned sh[4];
  const int tid = threadIdx.x;
  unsigned xcc = 0, rank = 0, nk = 0, dead = 0;
  if (tid == 0) {
    xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;      // HW_REG_XCC_ID, bits 3:0
    rank = __hip_atomic_fetch_add(state + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(state + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int polls = 0;
    while (ld_sc1(state + 8) < gridDim.x && ld_sc1(state + 9) == 0u) {
      if (++polls > kMaxPolls) { __hip_atomic_store(state + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(2);
    }
    nk = ld_sc1(state + xcc);
    dead = ld_sc1(state + 9);
    sh[0] = xcc; sh[1] = rank; sh[2] = nk; sh[3] = dead;
  }
  __syncthreads();
  xcc = sh[0]; rank = sh[1]; nk = sh[2]; dead = sh[3];
  // slot of workgroup (xcc, rank): 8 XCCs x 256 ranks x payload (any placement of <= 256 workgroups fits)
  float* const mine = slots + ((size_t)xcc * 256 + rank) * payload_floats;
  const float* const theirs = slots + ((size_t)xcc * 256 + (rank + 1) % (nk ? nk : 1)) * payload_floats;
  unsigned* const ctr = state + 32 * (1 + xcc);
  unsigned stale = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  const uint64_t w0 = wall_clock64();
  for (int r = 1; r <= rounds && !dead; ++r) {
    for (int i = tid * 4; i < payload_floats; i += 1024) *reinterpret_cast<float4*>(mine + i) = make_float4((float)r, (float)r, (float)r, (float)r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have reached the L2
    __syncthreads();                                          // ... and every other wave's of the workgroup
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * nk;
      int polls = 0;
      while (ld_sc1(ctr) < want) {
        if (++polls > kMaxPolls || ld_sc1(state + 9) != 0u) { __hip_atomic_store(state + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh[3] = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    dead = sh[3];
    if (dead) break;
    // the neighbour's slot, L1 bypassed (sc1), 16 bytes per lane
    for (int i = tid * 4; i < payload_floats; i += 1024) {
      const unsigned* q = reinterpret_cast<const unsigned*>(theirs + i);
      for (int e = 0; e < 4; ++e) stale += __uint_as_float(ld_sc1(q + e)) != (float)r;
    }
    // a second barrier keeps round r + 1's stores behind every read of round r (the product kernel has two syncs per layer as well)
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)r * nk;
      int polls = 0;
      while (ld_sc1(ctr + 8) < want) {
        if (++polls > kMaxPolls || ld_sc1(state + 9) != 0u) { __hip_atomic_store(state + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); sh[3] = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    dead = sh[3];
  }
  const uint64_t w1 = wall_clock64();
  const uint64_t t1 = __builtin_readcyclecounter();
  // per-workgroup record: xcc, rank, n_k, stale values seen, wall-clock ticks (100 MHz), shader cycles
  __shared__ unsigned sst;
  if (tid == 0) sst = 0u;
  __syncthreads();
  if (stale) atomicAdd(&sst, stale);
  __syncthreads();
  if (tid == 0) {
    unsigned* o = result + (size_t)blockIdx.x * 8;
    o[0] = xcc; o[1] = rank; o[2] = nk; o[3] = sst;
    o[4] = (unsigned)(w1 - w0); o[5] = (unsigned)(t1 - t0); o[6] = dead; o[7] = 0;
  }
}

}  // namespace

extern "C" int rdm_microbench_xcd_sync(uint32_t* state, float* slots, int32_t payload_floats, int32_t rounds, int32_t blocks, uint32_t* result,
                                       rdm_stream_t stream) {
  if (!state || !slots || !result || payload_floats < 0 || payload_floats % 4 != 0 || rounds < 0 || blocks < 1 || blocks > 256) return RDM_ERR_BAD_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(state, 0, 384 * 4, s) != hipSuccess) return RDM_ERR_HIP;      // 384 u32
  hipLaunchKernelGGL(k_xcd_sync, dim3(blocks), dim3(256), 0, s, state, slots, payload_floats, rounds, result);
  return hipGetLastError() == hipSuccess ? RDM_OK : RDM_ERR_HIP;
}
