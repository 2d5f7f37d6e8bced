// Shared internals of librdm_hip.so (gfx950 only).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/rdm_hip.h"

namespace rdm {

void set_error(const char* fmt, ...);

#define RDM_CHECK_ARG(cond, ...)                         \
  do {                                                   \
    if (!(cond)) {                                       \
      ::rdm::set_error(__VA_ARGS__);                     \
      return RDM_ERR_BAD_ARGUMENT;                       \
    }                                                    \
  } while (0)

#define RDM_HIP_OK(expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      ::rdm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RDM_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

// every launcher ends with this: catches bad launch configs without synchronising (and counts the launcher calls for
// rdm_launch_count(): bench.py reports launches per step)
extern long long g_launches;
#define RDM_LAUNCH_OK()                                                               \
  do {                                                                                \
    ++::rdm::g_launches;                                                              \
    hipError_t e_ = hipGetLastError();                                                \
    if (e_ != hipSuccess) {                                                           \
      ::rdm::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return RDM_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

// Development A/B switch behind rdm_debug_variant (the measured alternatives DESIGN.md cites).  It exists only in builds made with
// RDM_DEV_VARIANTS=1 (`RDM_DEV_VARIANTS=1 python -m md_rdm_amd.build`); in the shipped library it is the constant 0, every
// `g_variant == N` branch folds away at compile time and no launcher reads mutable process-wide state.
#ifdef RDM_DEV_VARIANTS
extern int g_variant;
#else
constexpr int g_variant = 0;
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Deterministic reduction mode (rdm_net_set_option RDM_NET_OPT_DETERMINISTIC; tests): while a plan call that asked for it is running on
// this thread, the launchers never split K (one workgroup owns every output element), take channel statistics with ONE row chunk per
// column group (a fixed summation order instead of f64 atomics from many workgroups) and gate / reduce the dgrad results in a separate
// ordered pass.  Slower, bit-reproducible.  Thread-local and scoped (DetScope): no process-wide mutable state.
extern thread_local bool t_deterministic;
struct DetScope {
  bool prev;
  explicit DetScope(bool on) : prev(t_deterministic) { t_deterministic = on; }
  ~DetScope() { t_deterministic = prev; }
};

// Launch census (rdm_census_*): while enabled, every MFMA launcher records WHICH kernel variant it picked (tile size, halo length,
// epilogue, split) under a readable name, so the parity tests can assert that the variants the headline geometry selects were the
// ones they compared with the oracle.  Off by default: one predictable branch per launch.
extern bool g_census_on;
void census_hit(const char* fmt, ...);
#define RDM_CENSUS(...) do { if (::rdm::g_census_on) ::rdm::census_hit(__VA_ARGS__); } while (0)

// ---------------------------------------------------------------------------------
// internal launchers (igemm.hip) - the C-ABI conv entry points and the network plan
// both go through these
// ---------------------------------------------------------------------------------
struct ConvGeom {
  int B, H, W;          // input spatial extent (pixels the A operand is gathered from)
  int Ho, Wo;           // output spatial extent (GEMM rows M = B*Ho*Wo)
  int KH, KW, SH, SW, PH, PW;
  int dir;              // +1: iy = oy*SH - PH + r (forward);  -1: iy = oy + PH - r (dgrad of a stride-1 conv)
};

enum Epilogue { EPI_STORE = 0, EPI_STORE_STATS = 1, EPI_MASK_STATS = 2, EPI_ATOMIC = 3, EPI_MASK_STATS_ATOMIC = 4 };

struct FwdArgs {            // C[m][n] = sum_{tap,c} f(A[pix(m,tap)][c]) * Wt[tap][n][c]     (B_KSTRIDED=false)
                            // C[m][n] = sum_{tap,c} f(A[pix(m,tap)][c]) * Wt[tap][c][n]     (B_KSTRIDED=true, dgrad)
  ConvGeom g;
  const float* A; int lda; int C;             // contracted channels per tap (multiple of 16)
  const float* a_scale; const float* a_shift; // optional BN-ReLU prologue on A (per contracted channel)
  const float* Wt; long wtap; int ldw;        // weight tap stride / row stride (floats)
  float* out; int ldc; int M, N;
  const float* bias;                          // EPI_STORE only
  double* stat0; double* stat1;               // STORE_STATS: sum v, sum v^2;  MASK_STATS: sum dz, sum dz*x
  const float* X; int ldx; const float* x_scale; const float* x_shift;  // MASK_STATS: forward pre-BN value + its affine
  int split_k;                                // >1 => EPI_ATOMIC into pre-zeroed out
  int xcd_flat;                               // 1: keep the hardware block order (A/B switch; default 0 = XCD-aware order)
  int add_out;                                // EPI_STORE_STATS: out = out + v (the statistics are taken of the SUM): completes a K-partial
  int accumulate;                             // 1: add into `out` (f32 atomics), never zero it - the caller owns the initial value
  unsigned a_bytes, w_bytes;                  // set by the launcher: addressable extents of A / Wt (buffer descriptors)
  // RAW BatchNorm prologue (training, few-pixel blocks): instead of finished (a_scale, a_shift) the kernel gets the channel sums and forms
  // the affine itself - the k_bn_finalize launch (5-9 us + a dependent-launch gap) leaves the critical chain; C <= RAWBN_MAX_C
  const double* a_sum; const double* a_sq; const float* a_gamma; const float* a_beta; double a_count;
  // mixed-precision arithmetic mode, xsplit.hip kernels with one bf16 MFMA per product only: `out` (3x3 dgrad) / `A` (1x1 dgrad) are rows of bf16
  // (ldc / lda in elements of that type)
  int out_bf16, a_bf16;
  int a_split;                                // xs 1x1 dgrad, three-product form: A (the gradient operand) is SPLIT ROWS (xsplit_dev.h)
  // xs 1x1 dgrad with the gate epilogue only: `out` is the block gradient and receives  out += x_scale * (gated dz)  (deferred norm1 backward)
  int acc_scaled;
  // conv3x3_halo_kernel, raw-BatchNorm form with the accumulating (K-split) epilogue only: one zeroed counter per pixel tile.  When set, the LAST
  // split of a tile to arrive takes the channel statistics (stat0 / stat1) of the finished tile in the same launch (no separate reduction pass).
  unsigned* tickets;
};
constexpr int RAWBN_MAX_C = 768;

// scale / shift of a training-mode BatchNorm from the channel sums: THE arithmetic of k_bn_finalize (elementwise.hip), shared so that the
// in-kernel prologue and the finalisation kernel agree bit for bit
__device__ __forceinline__ void bn_affine_from_sums(double sum, double sq, double count, float gamma, float beta, float eps, float& scale, float& shift,
                                                    float& mean, float& rstd, double& var_out) {
  const double mu = sum / count;
  double var = sq / count - mu * mu;
  if (var < 0) var = 0;
  mean = (float)mu;
  rstd = (float)(1.0 / sqrt(var + (double)eps));
  scale = gamma * rstd;
  shift = beta - mean * scale;
  var_out = var;
}

struct WgradArgs {          // dW[tap][n][c] += sum_m G[m][n] * f(Xs[pix(m,tap)][c])
  ConvGeom g;
  const float* G; int ldg; int N;             // gradient wrt conv output, rows n
  const float* Xs; int ldx; int C;            // forward conv input (pre BN-ReLU), rows c
  const float* x_scale; const float* x_shift; // optional BN-ReLU prologue (per c)
  float* dW; long wtap; int ldw;              // pre-zeroed, atomically accumulated
  int split_k;
  long n_items;                               // set by the launcher: column tiles x row tiles x splits
  unsigned g_bytes, x_bytes;                  // set by the launcher: addressable extents of G / Xs
  int xcd_flat;                               // 1: keep the hardware block order (A/B switch)
  int xsplit;                                 // != 0: the kernels of xsplit.hip may serve this launch (gradients only): 3 (or any value but 1) = split precision,
                                              // three bf16 MFMAs per product; 1 = operands rounded to bf16, one MFMA (the mixed-precision mode)
  int g_bf16;                                 // xsplit == 1 only: G is rows of bf16 (ldg in elements of that type)
  int g_split, x_split;                       // xsplit == 3, 1x1 only: G / Xs are SPLIT ROWS (xsplit_dev.h; Xs then already activated: no x_scale / x_shift)
  int g_frame;                                // xsplit == 3, 3x3 only: G is the frame image launch_frame_split_rows wrote (xsplit.h)
};

int launch_conv_fwd(const FwdArgs& a, bool b_kstrided, Epilogue epi, hipStream_t s);
int launch_conv_wgrad(const WgradArgs& a, hipStream_t s);
int pick_split_k(long tiles, long kslabs, int slots);
void profile_enable(bool on);
void* profile_begin(hipStream_t s, double flops, int kind, double bytes = 0);   // event bracket around one launch (no-op unless profiling is enabled)
double profile_kind_bytes(int kind);
void profile_end(void* ticket, hipStream_t s);
int profile_read(double* ms_sum, double* ms_union, double* flops, int* launches);
int profile_kind(int kind, const char** name, double* ms_sum, double* flops, int* launches);
size_t nyu_preprocess_workspace_bytes(int B, int H, int W, int h1, int w1, int out_w);
int launch_nyu_preprocess(const unsigned char* rgb, const float* depth, const void* aug_dev, int B, int H, int W, int h1, int w1, int oh, int ow,
                          float* x, float* y, void* ws, size_t ws_bytes, hipStream_t s);


// ---- 128-bit buffer stores with a SCALAR offset: MI355X needs a wait state the compiler does not insert --------------------------------
// A MUBUF store of more than 64 bits must not have its data VGPRs overwritten in the next 1-2 issue slots.  hipcc (LLVM
// GCNHazardRecognizer::createsVALUHazard) inserts those wait states EXCEPT when the store's soffset is an SGPR - the ISA manual's exemption.
// On gfx950 the exemption does not hold: tools/store_hazard/store_hazard_probe.hip (profiles/r05_store_hazard_probe.txt) stores the
// OVERWRITTEN dword 0 in lanes 12-15 of every 16 when a VALU write follows the store directly, and stores correctly behind one s_nop.  (Found in
// round 4 as "wrong values in lanes 12-15" of xs_dgrad3x3_kernel's gated output.)  Every >64-bit buffer store with a runtime scalar offset goes
// through this helper: the store and its two wait states are one asm block, so no schedule can separate them.
// tools/store_hazard/scan_isa.py audits the compiled ISA of the whole library for the unprotected pattern (tests/test_boundary.py runs it).
typedef unsigned int rdm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buffer_store_b128_soffset(rdm_u32x4 data, __amdgpu_buffer_rsrc_t srd, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(data), "v"(voffset), "s"(srd), "s"(soffset) : "memory");
#endif
}

}  // namespace rdm
