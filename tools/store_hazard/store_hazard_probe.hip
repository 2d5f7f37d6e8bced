// Reproducer for DESIGN.md 4.1c's "wrong values in lanes 12-15 after raw_buffer_store_b128": a 128-bit BUFFER store whose soffset is an SGPR,
// followed IMMEDIATELY by a VALU write of one of its data registers.  The ISA manual (and LLVM's GCNHazardRecognizer::createsVALUHazard) say a
// >64-bit VMEM store needs 1-2 wait states before its data VGPRs are overwritten, EXCEPT for MUBUF stores with an SGPR soffset, which need none.
// Variants: 0 = SGPR soffset, no wait state (what hipcc emits); 1 = SGPR soffset + s_nop 0; 2 = SGPR soffset + s_nop 1; 3 = immediate soffset +
// s_nop 1 (what hipcc emits for the non-exempt form).  Build: hipcc --offload-arch=gfx950 -O2 store_hazard_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V> __global__ void k(float* out, unsigned bytes, int rounds) {
  const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)bytes, 0x00020000);
  const unsigned tid = threadIdx.x, lane4 = (blockIdx.x * blockDim.x + tid) * 4;
  for (int r = 0; r < rounds; ++r) {
    const unsigned voff = tid * 16, soff = (blockIdx.x * blockDim.x + (unsigned)r * gridDim.x * blockDim.x) * 16;
    const float a = (float)(lane4 + 0), b = (float)(lane4 + 1), c = (float)(lane4 + 2), d = (float)(lane4 + 3), poison = -1.0f;
#define PRE "v_mov_b32 v100, %0\n v_mov_b32 v101, %1\n v_mov_b32 v102, %2\n v_mov_b32 v103, %3\n s_nop 4\n"
#define POST "v_mov_b32 v100, %7\n v_mov_b32 v101, %7\n v_mov_b32 v102, %7\n v_mov_b32 v103, %7\n"
#define ARGS : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(srd), "s"(soff), "v"(poison) : "v100", "v101", "v102", "v103", "memory"
    if (V == 0) asm volatile(PRE "buffer_store_dwordx4 v[100:103], %4, %5, %6 offen\n" POST ARGS);
    if (V == 1) asm volatile(PRE "buffer_store_dwordx4 v[100:103], %4, %5, %6 offen\n s_nop 0\n" POST ARGS);
    if (V == 2) asm volatile(PRE "buffer_store_dwordx4 v[100:103], %4, %5, %6 offen\n s_nop 1\n" POST ARGS);
    if (V == 3) { const unsigned vo2 = voff + soff; asm volatile(PRE "buffer_store_dwordx4 v[100:103], %4, %5, 0 offen\n s_nop 1\n" POST : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(vo2), "s"(srd), "s"(soff), "v"(poison) : "v100", "v101", "v102", "v103", "memory"); }
  }
}
template <int V> static void run(const char* what) {
  const int blocks = 4096, threads = 256, rounds = 8;
  const size_t n = (size_t)blocks * threads * rounds * 4;
  float* d; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, (unsigned)(n * 4), rounds);
  hipDeviceSynchronize();
  std::vector<float> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  long bad = 0, hist[16] = {0}, dw[4] = {0};
  for (size_t i = 0; i < n; ++i) {
    const size_t t = (i / 4) % ((size_t)blocks * threads);
    if (h[i] != (float)(t * 4 + i % 4)) { ++bad; ++hist[t % 16]; ++dw[i % 4]; }
  }
  printf("variant %d (%s): %ld of %zu stored dwords wrong; by lane %% 16:", V, what, bad, n);
  for (int i = 0; i < 16; ++i) printf(" %ld", hist[i]);
  printf("; by dword: %ld %ld %ld %ld\n", dw[0], dw[1], dw[2], dw[3]);
  hipFree(d);
}
int main() {
  run<0>("SGPR soffset, data overwritten in the next instruction");
  run<1>("SGPR soffset, s_nop 0");
  run<2>("SGPR soffset, s_nop 1");
  run<3>("immediate soffset, s_nop 1");
  return 0;
}
