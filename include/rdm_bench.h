/* librdm_bench.so - measurement kernels of the development tools (tools/*.py, bench_ops.py).  NOT part of the product: the shipped
 * library is md_rdm_amd/librdm_hip.so (include/rdm_hip.h); this one is built next to it from md_rdm_amd/csrc/bench/ and links against it. */
#ifndef RDM_BENCH_H_
#define RDM_BENCH_H_
#include "rdm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Attainable-peak microbenchmarks: float4 stream copy (HBM) and a register-only
 * v_mfma_f32_16x16x4_f32 loop (blocks x 4 waves x iters x 12 MFMAs of 2048 FLOP). */
int rdm_microbench_copy(const float* src, float* dst, int64_t n_floats, rdm_stream_t stream);
int rdm_microbench_mfma_f32(float* scratch, int32_t blocks, int32_t iters, rdm_stream_t stream);
/* The conv kernels' 48-MFMA slab (4 x 3 tiles, distinct operands) as a loop with ONE staging ingredient added per mode: 0 registers only,
 * 1 + the slab's ds_read_b128 fragment reads, 2 + a workgroup barrier, 3 + flat-address LDS-DMA staging, 4 the A tile only, 5 issued but
 * never waited for, 6 buffer-form LDS-DMA from a window of span_floats (power of two) of `scratch` - L1-, L2-, Infinity-Cache- or
 * HBM-resident depending on the span.  FLOPs = blocks * 4 waves * slabs * 48 * 2048. */
int rdm_microbench_mfma_staged_f32(float* scratch, int64_t scratch_floats, int32_t blocks, int32_t slabs, int32_t mode, int64_t span_floats,
                                   rdm_stream_t stream);
/* Pipeline experiment: f32 GEMM c[m][n] = sum_k a[m][k] * w[n][k] with LDS-DMA staging (global_load_lds_dwordx4) and
 * ds_read_b128 fragments, 128 x 96 tiles; k a multiple of 16; variant = LDS buffers (2 or 3). */
int rdm_microbench_gemm_dma_f32(const float* a, int32_t lda, const float* w, int32_t ldw, float* c, int32_t ldc, int32_t m, int32_t n, int32_t k,
                                int32_t variant, rdm_stream_t stream);

/* Per-XCD barrier + same-XCD hand-off probe (csrc/bench/xcd_sync.hip): `blocks` (<= 256) workgroups group themselves by the hardware XCC_ID, then run
 * `rounds` rounds of {write payload_floats floats, drain, arrive on the XCD's counter, poll it with sc1 loads, read the right-hand neighbour's payload with
 * sc1 loads, second barrier}.  state: 384 u32 (zeroed here); slots: 8 * 256 * payload_floats floats; result: 8 u32 per workgroup = xcc, rank, members of
 * its XCD, values read that were not the round's, wall-clock ticks (100 MHz) and shader cycles of the rounds, timeout flag, 0.  Every wait is bounded. */
int rdm_microbench_xcd_sync(uint32_t* state, float* slots, int32_t payload_floats, int32_t rounds, int32_t blocks, uint32_t* result, rdm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RDM_BENCH_H_ */
