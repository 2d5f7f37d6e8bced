// AddressSanitizer check of the HOST side of librdm_hip ("compile-time -fsanitize=address host build of the
// C-ABI shim").  Built by tests/host_asan/Makefile from the library's own sources with --cuda-host-only (no device code, no GPU
// needed) and run by tests/test_host_asan.py: every entry point that does host-side work without launching - the tensor registry,
// plan construction and workspace layout for several geometries, the backward stage table, workspace-size queries - plus the
// argument-validation paths of the launching entry points (they must return a status code BEFORE touching the device).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/rdm_hip.h"

#define CHECK(c)                                                             \
  do {                                                                       \
    if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
  } while (0)

int main() {
  CHECK(rdm_version() > 0);
  const int nt = rdm_net_num_tensors();
  CHECK(nt == 968);
  long params = 0;
  for (int i = 0; i < nt; ++i) {
    const char* name = rdm_net_tensor_name(i);
    CHECK(name != nullptr && strlen(name) > 3);
    CHECK(rdm_net_tensor_numel(i) >= 0);
    if (rdm_net_tensor_is_param(i) == 1) params += rdm_net_tensor_numel(i);
  }
  CHECK(params == 90529721);
  CHECK(rdm_net_tensor_name(-1) == nullptr && rdm_net_tensor_name(nt) == nullptr && rdm_net_tensor_numel(nt) == -1);

  const int geoms[][3] = {{1, 226, 226}, {2, 228, 228}, {16, 228, 304}, {8, 352, 1216}, {3, 33, 33}, {1, 1000, 37}};
  for (auto& g : geoms) {
    rdm_net* net = nullptr;
    CHECK(rdm_net_create(g[0], g[1], g[2], &net) == RDM_OK && net);
    int32_t h = 0, w = 0;
    CHECK(rdm_net_output_hw(net, &h, &w) == RDM_OK && h > 0 && w > 0);
    CHECK(rdm_net_workspace_bytes(net) > 0 && rdm_net_bf16_workspace_bytes(net) > 0 && rdm_net_bf16_weight_bytes(net) > 180000000);
    CHECK(rdm_net_forward_flops(net) > 0 && rdm_net_backward_flops(net) > rdm_net_forward_flops(net) && rdm_net_bf16_forward_bytes(net) > 0);
    int64_t off = 0, numel = 0;
    const char* names[] = {"blk0", "blk3", "G1", "logits", "e1", "dZ", "dZ1", "Y0_0", "Y2_35", "bn1_3_23", "bn2_1_11", "P2"};
    for (const char* nm : names) CHECK(rdm_net_buffer(net, nm, &off, &numel) == RDM_OK && off >= 0 && (size_t)off < rdm_net_workspace_bytes(net));
    CHECK(rdm_net_buffer(net, "no_such_buffer", &off, &numel) == RDM_ERR_BAD_ARGUMENT);
    CHECK(rdm_net_buffer(net, "Y2_36", &off, &numel) == RDM_ERR_BAD_ARGUMENT);
    CHECK(rdm_net_set_option(net, RDM_NET_OPT_PACKED_3X3, 1) == RDM_OK && rdm_net_set_option(net, 77, 1) == RDM_ERR_BAD_ARGUMENT);
    // launching entry points must refuse bad arguments on the host, before any device call
    std::vector<void*> T(nt, nullptr);
    float dummy[4];
    CHECK(rdm_net_forward(net, nullptr, T.data(), dummy, 16, dummy, 1, nullptr) == RDM_ERR_BAD_ARGUMENT);
    CHECK(rdm_net_forward(net, dummy, T.data(), dummy, 16, dummy, 1, nullptr) == RDM_ERR_WORKSPACE_TOO_SMALL);
    CHECK(rdm_net_backward(net, dummy, T.data(), T.data(), dummy, 16, 2, 1, nullptr) == RDM_ERR_BAD_ARGUMENT);
    CHECK(rdm_net_backward_stage(net, dummy, T.data(), T.data(), dummy, 16, 99, nullptr) == RDM_ERR_BAD_ARGUMENT);
    CHECK(rdm_net_backward_stage(net, dummy, T.data(), T.data(), dummy, 16, 0, nullptr) == RDM_ERR_WORKSPACE_TOO_SMALL);
    CHECK(rdm_net_bf16_prepare(net, T.data(), dummy, 16, nullptr) == RDM_ERR_WORKSPACE_TOO_SMALL);
    CHECK(strlen(rdm_last_error_string()) > 0);
    rdm_net_destroy(net);
  }
  rdm_net* bad = nullptr;
  CHECK(rdm_net_create(0, 228, 304, &bad) == RDM_ERR_BAD_ARGUMENT && rdm_net_create(1, 16, 16, &bad) == RDM_ERR_BAD_ARGUMENT);
  CHECK(rdm_net_create(1, 228, 304, nullptr) == RDM_ERR_BAD_ARGUMENT);

  int32_t a = 0, b = 0, prev = -1;
  for (int s = 0; s < 4; ++s) CHECK(rdm_net_segment_range(s, &a, &b) == RDM_OK && a <= b);
  CHECK(rdm_net_segment_range(4, &a, &b) == RDM_ERR_BAD_ARGUMENT);
  const int ns = rdm_net_num_backward_stages();
  CHECK(ns >= 10 && ns <= 20);
  for (int s = 0; s < ns; ++s) {
    CHECK(rdm_net_backward_stage_range(s, &a, &b) == RDM_OK && a <= b && (prev < 0 || b == prev - 1));
    prev = a;
  }
  CHECK(prev == 0 && rdm_net_backward_stage_range(ns, &a, &b) == RDM_ERR_BAD_ARGUMENT);

  CHECK(rdm_als_workspace_bytes(64, 16, 256, 64, 100) > 0);
  CHECK(rdm_conv3x3_bf16_workspace_bytes(2736, 8, 57, 76) > 0 && rdm_conv3x3_bf16_workspace_bytes(0, 8, 57, 76) == 0);
  CHECK(rdm_padavgpool2_bwd_workspace_bytes(384) >= 2 * 384 * 8 + 3 * 384 * 4 && rdm_padavgpool2_bwd_workspace_bytes(0) == 0);
  CHECK(rdm_nyu_preprocess_workspace_bytes(16, 480, 640, 250, 333, 304) > 0);
  rdm_conv_desc d = {2, 8, 10, 384, 384, 48, 48, 3, 3, 1, 1, 1, 1};
  CHECK(rdm_conv2d_fwd(&d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == RDM_ERR_BAD_ARGUMENT);
  d.in_c = 0;
  float x[4];
  CHECK(rdm_conv2d_fwd(&d, x, x, nullptr, nullptr, nullptr, x, nullptr, nullptr, nullptr) == RDM_ERR_BAD_ARGUMENT);
  CHECK(rdm_gemm_bf16(x, 8, 7, nullptr, nullptr, x, 8, nullptr, x, 8, 4, 4, 0, nullptr, 0, nullptr) == RDM_ERR_BAD_ARGUMENT);   // K not a multiple of 8
  CHECK(rdm_adamw_fused(x, x, x, x, -1, 1e-4f, .9f, .999f, 1e-8f, .01f, 1, 1.f, nullptr) == RDM_ERR_BAD_ARGUMENT);
  CHECK(rdm_bn_stats(x, 6, 10, 6, nullptr, nullptr, nullptr) == RDM_ERR_BAD_ARGUMENT);
  CHECK(rdm_launch_count() == 0);                         // nothing above reached a launch
  printf("host-side ASAN check ok: %d tensors, %ld parameters, %d backward stages\n", nt, params, ns);
  return 0;
}
