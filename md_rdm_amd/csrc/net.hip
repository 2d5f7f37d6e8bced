// Native execution plan for the convolutional stack of DepthEstimationNet:
//   encoder (network/RDM_Net.py:73-94, built at :493-532) + Decoder d_1 up to conv2 (:137-159),
// forward and backward, as one host-side walk that enqueues every kernel on the caller's stream.
//
// MI355X-first structure (this is NOT how the reference runs it):
//  * concat-free DenseNet: one NHWC buffer per dense block; each layer's 3x3 conv writes its 48
//    channels in place (torch.cat in torchvision's _DenseBlock re-copies the growing tensor).
//  * BatchNorm+ReLU never materialise: per-channel sum / sum^2 come out of the PRODUCING conv's
//    epilogue, a tiny finalise kernel turns them into (scale, shift), the CONSUMING conv applies
//    them in its staging registers.  Channel statistics of a block buffer are computed once and
//    shared by every later layer that normalises the same channels.
//  * transitions pool before the (linear) 1x1 conv: 4x fewer GEMM rows.
//  * everything needed by backward is kept (288 GB HBM: ~7 GB at B=16 228x304) instead of the
//    reference's memory_efficient recomputation.
#include <stdlib.h>
#include <string>
#include <mutex>
#include <vector>

#include "rdm_common.h"
#include "elementwise.h"
#include "bf16.h"
#include "wino.h"
#include "xsplit.h"

namespace rdm {


namespace {

constexpr int GROWTH = 48;
struct BlockDef { const char* name; int layers; int cin; int bn_size; };
struct TransDef { const char* name; int cin; int cout; };
const BlockDef kBlocks[4] = {{"encoder.dense_e2", 6, 96, 57}, {"encoder.dense_e3", 12, 192, 29}, {"encoder.dense_e4", 36, 384, 15},
                             {"d_1.dense_layer", 24, 1056, 8}};
const TransDef kTrans[3] = {{"encoder.trans_e2", 384, 192}, {"encoder.trans_e3", 768, 384}, {"encoder.trans_e4", 2112, 1056}};

struct TensorInfo { std::string name; int64_t numel; int is_param; };
struct BnIdx { int w, b, rm, rv, nbt; };
struct LayerIdx { BnIdx bn1; int conv1; BnIdx bn2; int conv2; };
struct Registry {
  std::vector<TensorInfo> t;
  int stem_w, stem_b;
  std::vector<LayerIdx> layers[4];
  BnIdx trans_bn[3]; int trans_conv[3];
  int conv1_w, conv1_b, conv2_w, conv2_b;
  int seg_first[4], seg_last[4];
  // backward STAGES: the 4 segments cut into runs of dense layers holding >= ~6.5 M gradient floats (~25 MB, the DDP bucket size
  // the reference's Lightning DDP uses) so a gradient exchange can start every few layers and the LAST one is small
  struct Stage { int seg, block, i_hi, i_lo; bool first_of_seg, last_of_seg; int first_t, last_t; };
  std::vector<Stage> stages;
  long layer_floats(int b, int i) const {
    const LayerIdx& L = layers[b][i];
    return t[L.bn1.w].numel * 2 + t[L.conv1].numel + t[L.bn2.w].numel * 2 + t[L.conv2].numel;
  }
  void build_stages() {
    const long kBucket = 6500000;
    for (int seg = 0; seg < 4; ++seg) {
      const int b = 3 - seg, nl = (int)layers[b].size();
      int hi = nl - 1;
      const size_t first_stage = stages.size();
      while (hi >= 0) {
        long acc = 0;
        int lo = hi;
        for (;;) {
          acc += layer_floats(b, lo);
          if (acc >= kBucket || lo == 0) break;
          --lo;
        }
        if (lo > 0 && lo <= 1) lo = 0;                           // do not leave a one-layer tail
        Stage st{seg, b, hi, lo, false, false, layers[b][lo].bn1.w, layers[b][hi].conv2};
        stages.push_back(st);
        hi = lo - 1;
      }
      Stage& f = stages[first_stage];
      Stage& l = stages.back();
      f.first_of_seg = true; l.last_of_seg = true;
      f.last_t = seg_last[seg];                                  // head (seg 0) / the transition after the block (segs 1..3) run first
      l.first_t = seg_first[seg];                                // the stem's tensors ride with the last stage of segment 3
    }
  }
  int add(const std::string& n, int64_t numel, int p) { t.push_back({n, numel, p}); return (int)t.size() - 1; }
  BnIdx add_bn(const std::string& p, int c) {
    BnIdx b;
    b.w = add(p + ".weight", c, 1); b.b = add(p + ".bias", c, 1);
    b.rm = add(p + ".running_mean", c, 0); b.rv = add(p + ".running_var", c, 0); b.nbt = add(p + ".num_batches_tracked", 1, 0);
    return b;
  }
  void add_block(int bi) {
    const BlockDef& d = kBlocks[bi];
    for (int i = 0; i < d.layers; ++i) {
      const std::string p = std::string(d.name) + ".denselayer" + std::to_string(i + 1);
      const int c = d.cin + i * GROWTH, cb = d.bn_size * GROWTH;
      LayerIdx L;
      L.bn1 = add_bn(p + ".norm1", c);
      L.conv1 = add(p + ".conv1.weight", (int64_t)cb * c, 1);
      L.bn2 = add_bn(p + ".norm2", cb);
      L.conv2 = add(p + ".conv2.weight", (int64_t)GROWTH * cb * 9, 1);
      layers[bi].push_back(L);
    }
  }
  void add_trans(int ti) {
    trans_bn[ti] = add_bn(std::string(kTrans[ti].name) + ".norm", kTrans[ti].cin);
    trans_conv[ti] = add(std::string(kTrans[ti].name) + ".conv.weight", (int64_t)kTrans[ti].cout * kTrans[ti].cin, 1);
  }
  Registry() {
    stem_w = add("encoder.conv_e1.weight", 96 * 3 * 49, 1);
    stem_b = add("encoder.conv_e1.bias", 96, 1);
    const int e2_first = (int)t.size();
    add_block(0); add_trans(0);
    const int e3_first = (int)t.size();
    add_block(1); add_trans(1);
    const int e4_first = (int)t.size();
    add_block(2); add_trans(2);
    const int d_first = (int)t.size();
    add_block(3);
    conv1_w = add("d_1.conv1.weight", 2208, 1); conv1_b = add("d_1.conv1.bias", 1, 1);
    conv2_w = add("d_1.conv2.weight", 180 * 2208, 1); conv2_b = add("d_1.conv2.bias", 180, 1);
    const int d_last = (int)t.size() - 1;
    const char* wl[8] = {"d0", "f1", "f2", "f3", "f4", "f5", "f6", "f7"};
    for (int i = 0; i < 8; ++i) add(std::string("weight_layer.") + wl[i], i < 4 ? 1 : 0, 1);
    seg_first[0] = d_first; seg_last[0] = d_last;
    seg_first[1] = e4_first; seg_last[1] = d_first - 1;
    seg_first[2] = e3_first; seg_last[2] = e4_first - 1;
    seg_first[3] = 0; seg_last[3] = e3_first - 1;
    build_stages();
  }
};
const Registry& reg() { static Registry r; return r; }

struct Bump {
  size_t off = 0;
  template <typename T> size_t take(size_t n) { off = (off + 255) & ~(size_t)255; size_t o = off; off += n * sizeof(T); return o; }
};

struct LayerWs { size_t Y, statY, bn1, bn2, w2p, bs2, bs1, dw3; };   // bs*: backward [sum dz | sum dz*x] f64 slots      // bn*: [scale|shift|mean|rstd] x C floats
struct BlockGeom { int H, W, M, ctot, cb; };

}  // namespace

// The weight-gradient stream is ONE per device for the whole process, shared by every plan: the runtime maps streams onto a handful of hardware
// queues round-robin, and with a stream per plan the FOURTH plan of a process got the hardware queue of the caller's stream - its weight
// gradients then ran in series with the dgrad chain (71 ms instead of 52 ms per step, measured) although nothing in the program had changed.
static hipStream_t shared_side_stream(bool default_priority) {
  static std::mutex mu;
  static hipStream_t streams[2][64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  hipStream_t& s = streams[default_priority ? 1 : 0][dev];
  if (!s) {
    // lowest priority: the side stream carries bulk work (weight gradients) that should fill what the dependent chain on the
    // caller's stream leaves free, not compete with it for workgroup slots
    int prio_least = 0, prio_greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (e == hipSuccess) e = default_priority ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking) : hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio_least);
    if (e != hipSuccess) { set_error("side stream: %s", hipGetErrorString(e)); s = nullptr; }
  }
  return s;
}

// pixel counts from which layout() sizes the per-layer pack buffers of the split kernels; xs_block() / xf_block() can never select below them
constexpr int XS_LAYOUT_MIN_PIXELS = 1024, XF_LAYOUT_MIN_PIXELS = 8192;

struct NetImpl {
  int B, H0, W0, H1, W1;
  BlockGeom bg[4];
  int M1;
  // workspace offsets
  size_t patches, e1, argmax, blk[4], blkstat[4], stats_begin, stats_end, stem_wp, logits, w2pad, tickets;
  std::vector<LayerWs> lws[4];
  size_t transP[3], transBn[3];
  // backward scratch
  size_t G[4], dZ[2], dZ1, cA, cB, cC, dP, gE1, dWstem, dL, tmp64, bwd_stats_begin, bwd_stats_end, transBs[3];
  // weight gradients run on a library-owned side stream, fenced with events against the caller's
  // stream: wgrad of a layer only depends on tensors that are final when its dgrad chain starts
  hipStream_t side = nullptr;
  hipEvent_t ev_go = nullptr, ev_dy = nullptr, ev_dz[2] = {nullptr, nullptr}, ev_side = nullptr, ev_fs[2] = {nullptr, nullptr}, ev_fa[2] = {nullptr, nullptr};
  bool dz_busy[2] = {false, false};
  int ensure_side() {
    if (side) return 0;
    side = shared_side_stream(g_variant == 14);
    if (!side) return RDM_ERR_HIP;
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_go, hipEventDisableTiming));
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_dy, hipEventDisableTiming));
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_dz[0], hipEventDisableTiming));
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_dz[1], hipEventDisableTiming));
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_pkf, hipEventDisableTiming));
    RDM_HIP_OK(hipEventCreateWithFlags(&ev_pkb, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
      RDM_HIP_OK(hipEventCreateWithFlags(&ev_fs[i], hipEventDisableTiming));
      RDM_HIP_OK(hipEventCreateWithFlags(&ev_fa[i], hipEventDisableTiming));
    }
    return 0;
  }
  ~NetImpl() {
    if (side) { hipStreamSynchronize(side); hipEventDestroy(ev_go); hipEventDestroy(ev_dy); hipEventDestroy(ev_dz[0]); hipEventDestroy(ev_dz[1]); hipEventDestroy(ev_side); hipEventDestroy(ev_pkf); hipEventDestroy(ev_pkb); hipEventDestroy(ev_fs[0]); hipEventDestroy(ev_fs[1]); hipEventDestroy(ev_fa[0]); hipEventDestroy(ev_fa[1]); }
  }
  // Winograd F(2x2, 3x3) for the 3x3 convs of the blocks with many pixels (wino.hip): transformed weights per layer (formed on the side
  // stream at the start of forward) and one scratch for the per-split partial outputs
  bool wino_fwd[4] = {false, false, false, false}, wino_wg[4] = {false, false, false, false};
  int wino_split[4] = {1, 1, 1, 1}, wino_split_x6[4] = {1, 1, 1, 1};
  int opt_wino_x6 = 0;         // RDM_NET_OPT_WINO_X6: the Winograd forward of those blocks on the bf16x6 kernel (float32-equivalent, bf16 matrix pipe)
  bool wino_x6(int b) const { return opt_wino_x6 && opt_split_fwd && !opt_det && wino_fwd[b]; }
  size_t winoPartial = 0, winoPartialFloats = 0;
  size_t winoVy = 0, winoVyFloats = 0, winoQ = 0, winoQFloats = 0;     // weight-gradient scratch (side stream: one launch at a time)
  size_t xsW = 0, xsWBytes = 0, xfW = 0, xfWBytes = 0;
  // per-layer packed weights of the split kernels (formed once per training step on the side stream at the start of the forward, off the
  // dependent chains): 3x3 dgrad / 1x1 dgrad / 1x1 forward images; empty where the block is below the kernels' default pixel thresholds
  std::vector<size_t> xsP3[4], xsP1[4], xfP[4];
  bool pk_fwd_valid = false, pk_bwd_valid = false;
  hipEvent_t ev_pkf = nullptr, ev_pkb = nullptr;
  int opt_prepack = 1;
  size_t deferBC = 0; int deferLd = 0;      // deferred norm1 backward: [parity 2][B | C][deferLd] running per-channel sums of the block being walked
  std::vector<size_t> winoU[4];
  size_t total;
  int training_saved = 1;
  int opt_join_seg = 0;        // RDM_NET_OPT_JOIN_PER_SEGMENT: the side stream joins the caller's stream at the end of each of the 4 segments only
  int opt_det = 0;             // RDM_NET_OPT_DETERMINISTIC: ordered reductions everywhere (tests), see DetScope
  int opt_no_wino = 0;         // RDM_NET_OPT_DIRECT_3X3: keep the direct implicit-GEMM kernels everywhere (A/B and tests)
  int opt_packed3x3 = 0;       // RDM_NET_OPT_PACKED_3X3: the 78 3x3 weights (and their gradients) are handed over as [tap][out][in]
  int opt_prezeroed = 0;       // RDM_NET_OPT_GRADS_PREZEROED: every gradient tensor is zero when backward stage 0 starts
  int opt_split_bwd = 0;       // RDM_NET_OPT_SPLIT_BWD: the backward GEMMs of the many-pixel blocks run the split-precision (bf16x3) kernels of xsplit.hip
  // blocks whose gradient GEMMs run the split-precision kernels: measured per block at B=16 228x304 - dense_e4 (4 560 pixels) gains on the 1x1
  // weight / input gradients and the 3x3 input gradient (173 -> 100, 134 -> 86, 63 -> 45 us per layer) but not on the 3x3 weight gradient (51 -> 56 us);
  // the decoder (1 280 pixels) lost overall while every layer paid two weight-pack launches and the norm1 pass on its chain (53.3 vs 52.6 ms per step);
  // with RDM_NET_OPT_PREPACK and RDM_NET_OPT_DEFER_NORM1 it gains (49.8-50.1 vs 50.4-50.7): the threshold is the kernels' own minimum
  int xs_min_pixels = XS_LAYOUT_MIN_PIXELS;
  bool xs_block(int b) const { return opt_split_bwd && !opt_det && bg[b].M >= xs_min_pixels; }
  int opt_split_rows = 1;      // RDM_NET_OPT_SPLIT_ROWS: dY and relu1(norm1(x)) reach the split 1x1 gradient kernels as SPLIT ROWS (xsplit_dev.h) written once by their producers
  size_t xsGf = 0;             // frame image of the layer's 48-channel output gradient for the split 3x3 weight gradient (side stream: one at a time)
  size_t xsXh = 0;             // split rows of the activation operand of the layer whose 1x1 weight gradient is running (side stream: one at a time)
  int opt_defer_norm1 = 1;     // RDM_NET_OPT_DEFER_NORM1: see k_bn_bwd_defer (elementwise.hip); blocks on the xs 1x1 dgrad only
  int opt_gemm_bf16 = 0;       // RDM_NET_OPT_GEMM_BF16: the launches routed to xsplit.hip round their operands to bf16 (one MFMA per product) - mixed-precision arithmetic
  int xs_np() const { return (opt_gemm_bf16 & 2) ? 1 : 3; }      // value bits: 1 = the forward GEMMs, 2 = the gradient GEMMs (3 = both)
  int opt_fuse_stats3 = 1;     // RDM_NET_OPT_FUSE_STATS3: the K-split 3x3 conv of the few-pixel blocks takes the channel statistics of its output in the same launch
  int opt_split_fwd = 0;       // RDM_NET_OPT_SPLIT_FWD: conv1 of the many-pixel blocks on the three-way-split bf16x6 forward kernel
  int xf_min_pixels = XF_LAYOUT_MIN_PIXELS;
  bool xf_block(int b) const { return opt_split_fwd && !opt_det && bg[b].M >= xf_min_pixels; }
  int xs_wg3_min_pixels = 8192;
  bool xs_block_wgrad3(int b) const { return xs_block(b) && bg[b].M >= xs_wg3_min_pixels; }
  // ---- reduced-precision forward (bf16.hip): prepared-weight buffer layout + activation workspace layout ----
  struct Bf16Layer { size_t w1, w3, bn1, bn2; };
  std::vector<Bf16Layer> bfl[4];
  size_t bf_wt[3], bf_tbn[3], bf_stem_w, bf_head_w, bf_wtotal = 0;
  size_t bf_patches, bf_e1, bf_blk[4], bf_Y, bf_P[3], bf_logits, bf_partial, bf_partial_floats, bf_counters, bf_total = 0;
  int bf_cbp[4];            // bottleneck width padded to a multiple of 32 (zero weight rows / columns, zero affine: the pad channels of Y are exact zeros)
  bool bf_act3[4];          // the block's 3x3 runs conv3x3_act_bf16_kernel (LDS-DMA, fragment-order weight image); else conv3x3_bf16_kernel without prologue
  static constexpr int kBfCounters = 4096;
  double bf_bytes = 0;      // algorithmic HBM bytes of one bf16 forward (every activation written once, read by its consumers once; weights once)
  void plan_bf16() {
    Bump w;
    for (int b = 0; b < 4; ++b) {
      const int cbp = (bg[b].cb + 31) / 32 * 32;
      bf_cbp[b] = cbp;
      // many-pixel blocks (dense_e2 / e3): the DMA kernel; few-pixel blocks are bound by launch + memory round trips, where the
      // register-staged kernel with its separate reduction launch measured faster (bf16_microbench: 15.3 vs 19.7 us at dense_e4).
      // By BLOCK, not by geometry: the prepared weight buffer (3x3 layout) must serve every plan of the same weights.
      bf_act3[b] = b < 2;
      bfl[b].resize(kBlocks[b].layers);
      for (int i = 0; i < kBlocks[b].layers; ++i) {
        const int cin = kBlocks[b].cin + i * GROWTH;
        bfl[b][i].w1 = w.take<unsigned short>((size_t)cbp * cin);
        bfl[b][i].w3 = w.take<unsigned short>((size_t)9 * GROWTH * cbp);             // either layout: 27 KiB per 32 channels
        bfl[b][i].bn1 = w.take<float>(4 * (size_t)cin);
        bfl[b][i].bn2 = w.take<float>(4 * (size_t)cbp);
      }
    }
    for (int t = 0; t < 3; ++t) {
      bf_wt[t] = w.take<unsigned short>((size_t)kTrans[t].cout * kTrans[t].cin);
      bf_tbn[t] = w.take<float>(4 * (size_t)kTrans[t].cin);
    }
    bf_stem_w = w.take<unsigned short>(96 * 160);
    bf_head_w = w.take<unsigned short>((size_t)180 * 2208);
    bf_wtotal = (w.off + 255) & ~(size_t)255;
    Bump a;
    size_t maxY = 0;
    bf_patches = a.take<unsigned short>((size_t)M1 * 160);
    bf_e1 = a.take<unsigned short>((size_t)M1 * 96);
    for (int b = 0; b < 4; ++b) {
      bf_blk[b] = a.take<unsigned short>((size_t)bg[b].M * bg[b].ctot);
      maxY = std::max(maxY, (size_t)bg[b].M * bf_cbp[b]);
    }
    bf_Y = a.take<unsigned short>(maxY);
    for (int t = 0; t < 3; ++t) bf_P[t] = a.take<unsigned short>((size_t)bg[t + 1].M * kTrans[t].cin);
    bf_logits = a.take<float>((size_t)bg[3].M * 192);
    bf_partial_floats = 0;                                        // K-split partial sums of the 3x3 convs
    for (int b = 0; b < 4; ++b) {
      if (bf_act3[b]) { bf_partial_floats = std::max(bf_partial_floats, conv3x3_act_partial_floats(bf_cbp[b], B, bg[b].H, bg[b].W)); continue; }
      const int split = std::min(std::max(cdiv(bf_cbp[b], 32) / 2, 1), 16);       // the launcher sizes the split to what it is given
      bf_partial_floats = std::max(bf_partial_floats, (size_t)split * bg[b].M * GROWTH);
    }
    bf_partial_floats = std::max(bf_partial_floats, (size_t)8 * bg[3].M * bf_cbp[3]);      // K-split of the decoder's 1x1 convs
    bf_partial = a.take<float>(bf_partial_floats);
    bf_counters = a.take<unsigned>(kBfCounters);
    bf_total = (a.off + 255) & ~(size_t)255;
    // algorithmic traffic: x (f32) + patches w+r + e1 w+r + per layer [prefix read + Y write + Y read + 48-slice write] + transitions + head
    double by = (double)B * 3 * H0 * W0 * 4 + 2.0 * M1 * 160 * 2 + 2.0 * M1 * 96 * 2 + (double)bg[0].M * 96 * 2;
    for (int b = 0; b < 4; ++b)
      for (int i = 0; i < kBlocks[b].layers; ++i)
        by += (double)bg[b].M * 2 * ((kBlocks[b].cin + i * GROWTH) + 2.0 * bg[b].cb + GROWTH) + 2.0 * ((double)bg[b].cb * (kBlocks[b].cin + i * GROWTH) + 9.0 * GROWTH * bg[b].cb);
    for (int t = 0; t < 3; ++t)
      by += (double)bg[t].M * kTrans[t].cin * 2 + 2.0 * bg[t + 1].M * kTrans[t].cin * 2 + (double)bg[t + 1].M * kTrans[t].cout * 2 + 2.0 * kTrans[t].cin * kTrans[t].cout;
    by += (double)bg[3].M * 2208 * 2 + 2.0 * 180 * 2208 + 2.0 * bg[3].M * 180 * 4;
    bf_bytes = by;
  }

  void plan() {
    H1 = (H0 + 6 - 7) / 2 + 1; W1 = (W0 + 6 - 7) / 2 + 1;
    M1 = B * H1 * W1;
    int h = (H1 + 2 - 3) / 2 + 1, w = (W1 + 2 - 3) / 2 + 1;
    for (int b = 0; b < 4; ++b) {
      bg[b].H = h; bg[b].W = w; bg[b].M = B * h * w;
      bg[b].ctot = kBlocks[b].cin + kBlocks[b].layers * GROWTH;
      bg[b].cb = kBlocks[b].bn_size * GROWTH;
      h = (h + 1) / 2; w = (w + 1) / 2;
    }
    Bump a;
    patches = a.take<float>((size_t)M1 * 160);
    e1 = a.take<float>((size_t)M1 * 96);
    argmax = a.take<unsigned char>((size_t)bg[0].M * 96);
    stem_wp = a.take<float>(96 * 160);
    for (int b = 0; b < 4; ++b) blk[b] = a.take<float>((size_t)bg[b].M * bg[b].ctot);
    // all f64 statistics live in one region so a single memset zeroes them per forward
    stats_begin = a.take<double>(0);
    for (int b = 0; b < 4; ++b) blkstat[b] = a.take<double>(2 * (size_t)bg[b].ctot);
    tickets = a.take<double>(64);                                     // 128 unsigned counters (conv3x3_halo_kernel's in-launch statistics): zeroed with the statistics, left zero by every launch
    for (int b = 0; b < 4; ++b) {
      lws[b].resize(kBlocks[b].layers);
      for (auto& L : lws[b]) L.statY = a.take<double>(2 * (size_t)bg[b].cb);
    }
    stats_end = a.take<double>(0);
    for (int b = 0; b < 4; ++b)
      for (int i = 0; i < kBlocks[b].layers; ++i) {
        LayerWs& L = lws[b][i];
        L.Y = a.take<float>((size_t)bg[b].M * bg[b].cb);
        L.bn1 = a.take<float>(4 * (size_t)(kBlocks[b].cin + i * GROWTH));
        L.bn2 = a.take<float>(4 * (size_t)bg[b].cb);
        L.w2p = a.take<float>(9 * (size_t)GROWTH * bg[b].cb);
      }
    for (int t = 0; t < 3; ++t) {
      transP[t] = a.take<float>((size_t)bg[t + 1].M * kTrans[t].cin);
      transBn[t] = a.take<float>(4 * (size_t)kTrans[t].cin);
    }
    logits = a.take<float>((size_t)bg[3].M * 192);
    w2pad = a.take<float>((size_t)192 * 2208);
    for (int b = 0; b < 4; ++b) {
      // Winograd where the 2.25x fewer MFMAs outweigh the transforms (tools/wino_bench.py, Cb = 720): forward 0.061 vs 0.078 ms direct at
      // 9 120 pixels, 0.087 vs 0.106 at 13 376 (KITTI's dense_e4), but 0.052 vs 0.044 at 4 560 (NYU's dense_e4); the weight gradient
      // ties at 9 120 (0.103 vs 0.100) and wins from 13 376 (0.113 vs 0.136).  dense_e2 / dense_e3 of the headline geometry: 1.5-1.65x.
      wino_fwd[b] = bg[b].M >= 8192 && bg[b].cb % 16 == 0;
      wino_wg[b] = bg[b].M >= 12288 && bg[b].cb % 16 == 0;
      if (!wino_fwd[b]) continue;
      const int T = B * ((bg[b].H + 1) / 2) * ((bg[b].W + 1) / 2);
      wino_split[b] = wino_pick_split(T, bg[b].cb / 16);
      wino_split_x6[b] = wino_pick_split(T, bg[b].cb / 16, true);
      const int sp = std::max(wino_split[b], wino_split_x6[b]);              // options arrive after the layout: sized for either kernel
      if (sp > 1) winoPartialFloats = std::max(winoPartialFloats, (size_t)sp * bg[b].M * 48);
      winoU[b].resize(kBlocks[b].layers);
      for (auto& o : winoU[b]) o = a.take<unsigned char>(std::max(wino_u_bytes(bg[b].cb, false), wino_u_bytes(bg[b].cb, true)));
    }
    winoPartial = a.take<float>(winoPartialFloats);
    for (int b = 0; b < 4; ++b)
      if (wino_wg[b]) {
        winoVyFloats = std::max(winoVyFloats, wino_wgrad_vy_floats(B, bg[b].H, bg[b].W));
        winoQFloats = std::max(winoQFloats, wino_wgrad_part_floats(B, bg[b].H, bg[b].W, bg[b].cb));
      }
    winoVy = a.take<float>(winoVyFloats);
    winoQ = a.take<float>(winoQFloats);
    // split-precision backward (xsplit.hip): fragment-order split weights of the layer whose 3x3 input gradient is running (main stream)
    for (int b = 0; b < 4; ++b)
      if (bg[b].M >= XS_LAYOUT_MIN_PIXELS) xsWBytes = std::max({xsWBytes, xs_dgrad3x3_workspace_bytes(bg[b].cb), xs_dgrad1x1_workspace_bytes(bg[b].cb, bg[b].ctot)});
    xsW = a.take<unsigned char>(xsWBytes);
    for (int b = 0; b < 4; ++b) {
      if (bg[b].M >= XS_LAYOUT_MIN_PIXELS) {
        xsP3[b].resize(kBlocks[b].layers); xsP1[b].resize(kBlocks[b].layers);
        for (int i = 0; i < kBlocks[b].layers; ++i) {
          xsP3[b][i] = a.take<unsigned char>(xs_dgrad3x3_workspace_bytes(bg[b].cb));
          xsP1[b][i] = a.take<unsigned char>(xs_dgrad1x1_workspace_bytes(bg[b].cb, kBlocks[b].cin + i * GROWTH));
        }
      }
      if (bg[b].M >= XF_LAYOUT_MIN_PIXELS) {
        xfP[b].resize(kBlocks[b].layers);
        for (int i = 0; i < kBlocks[b].layers; ++i) xfP[b][i] = a.take<unsigned char>(xs_fwd1x1_workspace_bytes(kBlocks[b].cin + i * GROWTH, bg[b].cb));
      }
    }
    {
      size_t xh = 0;                                             // [M][cin] split rows of the widest layer of every block on the split kernels
      for (int b = 0; b < 4; ++b)
        if (bg[b].M >= XS_LAYOUT_MIN_PIXELS) xh = std::max(xh, (size_t)bg[b].M * (size_t)(kBlocks[b].cin + (kBlocks[b].layers - 1) * GROWTH));
      xsXh = a.take<float>(xh);
      size_t gf = 0;
      for (int b = 0; b < 4; ++b)
        if (bg[b].M >= XS_LAYOUT_MIN_PIXELS) gf = std::max(gf, xs_frame_rows_bytes(B, bg[b].H, bg[b].W));
      xsGf = a.take<unsigned char>(gf);
    }
    for (int b = 0; b < 4; ++b) deferLd = std::max(deferLd, (bg[b].ctot + 63) / 64 * 64);
    deferBC = a.take<float>((size_t)4 * deferLd);
    for (int b = 0; b < 4; ++b)
      if (bg[b].M >= XS_LAYOUT_MIN_PIXELS) xfWBytes = std::max(xfWBytes, xs_fwd1x1_workspace_bytes(bg[b].ctot, bg[b].cb));
    xfW = a.take<unsigned char>(xfWBytes);
    // backward scratch
    size_t maxMC = 0, maxMCin = 0, maxC = 0, maxP = 0, maxCb = 0;
    for (int b = 0; b < 4; ++b) {
      G[b] = a.take<float>((size_t)bg[b].M * bg[b].ctot);
      maxMC = std::max(maxMC, (size_t)bg[b].M * bg[b].cb);
      maxMCin = std::max(maxMCin, (size_t)bg[b].M * bg[b].ctot);
      maxC = std::max(maxC, (size_t)std::max(bg[b].cb, bg[b].ctot));
      maxCb = std::max(maxCb, (size_t)bg[b].cb);
    }
    for (int t = 0; t < 3; ++t) maxP = std::max(maxP, (size_t)bg[t + 1].M * kTrans[t].cin);
    dZ[0] = a.take<float>(maxMC);
    dZ[1] = a.take<float>(maxMC);
    dZ1 = a.take<float>(maxMCin);
    for (int b = 0; b < 4; ++b)
      for (auto& L : lws[b]) L.dw3 = a.take<float>(9 * (size_t)GROWTH * bg[b].cb);
    // backward reductions: one slot per BatchNorm, zeroed by ONE memset at the start of backward
    bwd_stats_begin = a.take<double>(0);
    for (int b = 0; b < 4; ++b)
      for (int i = 0; i < kBlocks[b].layers; ++i) {
        lws[b][i].bs2 = a.take<double>(2 * (size_t)bg[b].cb);
        lws[b][i].bs1 = a.take<double>(2 * (size_t)(kBlocks[b].cin + i * GROWTH));
      }
    for (int t = 0; t < 3; ++t) transBs[t] = a.take<double>(2 * (size_t)kTrans[t].cin);
    bwd_stats_end = a.take<double>(0);
    cA = a.take<float>(maxC); cB = a.take<float>(maxC); cC = a.take<float>(maxC);
    dP = a.take<float>(maxP);
    gE1 = a.take<float>((size_t)M1 * 96);
    dWstem = a.take<float>(96 * 160);
    dL = a.take<float>((size_t)bg[3].M * 192);
    tmp64 = a.take<double>(512);
    total = (a.off + 255) & ~(size_t)255;
    plan_bf16();
  }
};

namespace {

template <typename T> T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }
inline float* F(void* const* tensors, int i) { return static_cast<float*>(tensors[i]); }

ConvGeom geom1x1(int B, int H, int W) { return ConvGeom{B, H, W, H, W, 1, 1, 1, 1, 0, 0, 1}; }
ConvGeom geom3x3(int B, int H, int W, int dir) { return ConvGeom{B, H, W, H, W, 3, 3, 1, 1, 1, 1, dir}; }

// statistics of rows of a matrix are fused in the conv epilogue on big layers; split-K layers
// (few rows) reduce with a separate pass
inline bool fuse_stats(int M, int N) {
  return (long)cdiv(M, 256) * cdiv(N, 48) >= 1536;   // flat between 512 and 3072 (swept)
}

// norm1 finalisation of layer i restricted to channels [c_lo, c_hi) of the block buffer
int finalize_norm1(NetImpl& n, int b, int i, int c_lo, int c_hi, bool count_batch, void* ws, void* const* T, int training, hipStream_t s) {
  const BlockGeom& g = n.bg[b];
  const LayerIdx& L = reg().layers[b][i];
  const int cin = kBlocks[b].cin + i * GROWTH;
  double* bst = at<double>(ws, n.blkstat[b]);
  float* bn1 = at<float>(ws, n.lws[b][i].bn1);
  return launch_bn_finalize(bst + c_lo, bst + g.ctot + c_lo, (double)g.M, F(T, L.bn1.w) + c_lo, F(T, L.bn1.b) + c_lo, F(T, L.bn1.rm) + c_lo,
                            F(T, L.bn1.rv) + c_lo, count_batch ? static_cast<long long*>(T[L.bn1.nbt]) : nullptr, bn1 + c_lo, bn1 + cin + c_lo,
                            bn1 + 2 * cin + c_lo, bn1 + 3 * cin + c_lo, c_hi - c_lo, training, s);
}

// conv1 (1x1) of layer i over the input channels [c_lo, c_hi); accumulate => atomically added into a zeroed Y
int conv1_range(NetImpl& n, int b, int i, int c_lo, int c_hi, bool accumulate, bool fuse, void* ws, void* const* T, hipStream_t s, bool add_out = false,
                bool raw_bn = false) {
  const BlockGeom& g = n.bg[b];
  const LayerIdx& L = reg().layers[b][i];
  const LayerWs& W = n.lws[b][i];
  const int cin = kBlocks[b].cin + i * GROWTH;
  float* bn1 = at<float>(ws, W.bn1);
  double* sty = at<double>(ws, W.statY);
  FwdArgs a{};
  a.g = geom1x1(n.B, g.H, g.W);
  a.A = at<float>(ws, n.blk[b]) + c_lo; a.lda = g.ctot; a.C = c_hi - c_lo; a.a_scale = bn1 + c_lo; a.a_shift = bn1 + cin + c_lo;
  a.Wt = F(T, L.conv1) + c_lo; a.wtap = 0; a.ldw = cin;
  a.out = at<float>(ws, W.Y); a.ldc = g.cb; a.M = g.M; a.N = g.cb;
  a.stat0 = sty; a.stat1 = sty + g.cb;
  a.accumulate = accumulate ? 1 : 0;
  a.add_out = add_out ? 1 : 0;
  if (raw_bn) {          // the consumer forms the BatchNorm affine of its channels from the block's channel sums (no finalisation launch on the chain)
    double* bst = at<double>(ws, n.blkstat[b]);
    a.a_scale = nullptr; a.a_shift = nullptr;
    a.a_sum = bst + c_lo; a.a_sq = bst + g.ctot + c_lo; a.a_gamma = F(T, L.bn1.w) + c_lo; a.a_beta = F(T, L.bn1.b) + c_lo; a.a_count = (double)g.M;
  }
  if (n.xf_block(b) && !accumulate && !add_out && !raw_bn && xs_fwd1x1_supported(a)) {
    const bool pk = n.pk_fwd_valid && !n.xfP[b].empty() && c_lo == 0 && c_hi == cin;
    return launch_xs_fwd1x1(a, fuse ? EPI_STORE_STATS : EPI_STORE, at<unsigned char>(ws, pk ? n.xfP[b][i] : n.xfW), pk ? xs_fwd1x1_workspace_bytes(cin, g.cb) : n.xfWBytes, s,
                            (n.opt_gemm_bf16 & 1) ? 1 : 6, pk);
  }
  const int rc = launch_conv_fwd(a, false, fuse ? EPI_STORE_STATS : EPI_STORE, s);
  return rc < 0 ? rc : 0;
}

int forward_block(NetImpl& n, int b, void* ws, void* const* T, int training, hipStream_t s) {
  const BlockGeom& g = n.bg[b];
  float* blk = at<float>(ws, n.blk[b]);
  double* bst = at<double>(ws, n.blkstat[b]);
  const int layers = kBlocks[b].layers;
  // Small-M blocks (dense_e4, decoder) are a strictly serial chain of short kernels that cannot fill the
  // chip.  DenseNet structure to the rescue: conv1 of layer i+1 contracts over ALL earlier channels, and
  // everything except the newest 48 is already final while layer i is still running - so that bulk
  // ("part A") runs one layer ahead on the side stream, and only the 48-channel remainder ("part B",
  // 3 K-slabs) stays on the critical path.  Both parts add atomically into a zeroed Y.
  const bool pipelined = !fuse_stats(g.M, g.cb) && layers > 1 && g_variant != 8 && !n.opt_det;   // part A / part B add atomically: not in deterministic mode
  hipStream_t side = n.side;
  int rc;
  if (b == 0 && n.pk_fwd_valid) RDM_HIP_OK(hipStreamWaitEvent(s, n.ev_pkf, 0));                 // the packed conv1 weights of the split forward kernel
  // Few-pixel blocks in training (round 3): the two BatchNorm finalisations of a layer leave the dependent chain - the consuming conv
  // forms (scale, shift) from the channel sums itself (RAW prologue), the running statistics and the coefficients backward needs are written
  // by BATCHED finalisation launches (24 BatchNorms each) - and the 48-channel output slices of ALL layers are zeroed by ONE launch at the start
  // of the block (their K-split 3x3 convs then only add).  Per layer: 6 -> 3 launches on the chain, ~2.9 launches fewer in all.
  const bool raw = pipelined && training && !n.opt_det && g.cb <= RAWBN_MAX_C && !(n.wino_fwd[b] && !n.opt_no_wino) && g.M <= 8192 && 2 * (g.W + 1) <= 128;
  if (raw && (rc = launch_zero_rows(blk + kBlocks[b].cin, g.M, g.ctot - kBlocks[b].cin, g.ctot, s))) return rc;
  // RAW mode bookkeeping (running statistics + the coefficients backward reads): nothing in forward waits for it, so it is BATCHED - up to
  // BN_BATCH BatchNorms per launch, enqueued on the caller's stream once their sums are final (the sums stay untouched until the next forward)
  BnBatch batch;
  int nbatch = 0, batch_maxc = 0;
  auto flush_batch = [&]() -> int {
    const int r = launch_bn_finalize_batch(batch, nbatch, batch_maxc, s);
    nbatch = 0; batch_maxc = 0;
    return r;
  };
  auto add_batch = [&](const double* sum, const double* sq, const BnIdx& bn, int c_lo, int c_hi, float* out, int cfull, bool count_batch) -> int {
    BnBatchEntry& e = batch.e[nbatch++];
    e.sum = sum + c_lo; e.sq = sq + c_lo; e.gamma = F(T, bn.w) + c_lo; e.beta = F(T, bn.b) + c_lo; e.rm = F(T, bn.rm) + c_lo; e.rv = F(T, bn.rv) + c_lo;
    e.nbt = count_batch ? static_cast<long long*>(T[bn.nbt]) : nullptr; e.out = out + c_lo; e.count = (double)g.M; e.C = c_hi - c_lo; e.Cout = cfull;
    batch_maxc = std::max(batch_maxc, c_hi - c_lo);
    return nbatch == BN_BATCH ? flush_batch() : 0;
  };
  if (pipelined) RDM_HIP_OK(hipEventRecord(n.ev_fs[1], s));        // statistics of the block's input channels are final ("layer -1")
  for (int i = 0; i < layers; ++i) {
    const LayerIdx& L = reg().layers[b][i];
    const LayerWs& W = n.lws[b][i];
    const int cin = kBlocks[b].cin + i * GROWTH;
    float* Y = at<float>(ws, W.Y);
    float* bn2 = at<float>(ws, W.bn2);
    double* sty = at<double>(ws, W.statY);
    // ---- side stream: part A of layer i+1 (channels [0, cin), final once layer i-1 has published its statistics) ----
    if (raw && i > 0 && (rc = add_batch(bst, bst + g.ctot, L.bn1, cin - GROWTH, cin, at<float>(ws, W.bn1), cin, false))) return rc;   // layer i's newest 48 channels (sums final since layer i-1)
    if (pipelined && i + 1 < layers) {
      RDM_HIP_OK(hipStreamWaitEvent(side, n.ev_fs[(i + 1) & 1], 0));                 // recorded at the end of layer i-1
      if ((rc = launch_zero_rows(at<float>(ws, n.lws[b][i + 1].Y), 1, (long)g.M * g.cb, (long)g.M * g.cb, side))) return rc;
      if ((rc = finalize_norm1(n, b, i + 1, 0, cin, true, ws, T, training, side))) return rc;
      if ((rc = conv1_range(n, b, i + 1, 0, cin, true, false, ws, T, side))) return rc;
      RDM_HIP_OK(hipEventRecord(n.ev_fa[(i + 1) & 1], side));
    }
    // ---- main: conv1 of layer i ----
    const bool fuse = training && fuse_stats(g.M, g.cb);
    bool stats_done = false;
    if (pipelined && i > 0) {
      if (!raw && (rc = finalize_norm1(n, b, i, cin - GROWTH, cin, false, ws, T, training, s))) return rc;
      RDM_HIP_OK(hipStreamWaitEvent(s, n.ev_fa[i & 1], 0));
      if (training && g_variant != 36) {
        // part B is three K-slabs and runs unsplit: its epilogue adds part A's finished sum, stores the final value and takes
        // the channel statistics of it - no separate reduction pass over Y on the critical path (it took 56 us beside part A)
        if ((rc = conv1_range(n, b, i, cin - GROWTH, cin, false, true, ws, T, s, true, raw))) return rc;
        stats_done = true;
      } else if ((rc = conv1_range(n, b, i, cin - GROWTH, cin, true, false, ws, T, s))) return rc;
    } else {
      if ((rc = finalize_norm1(n, b, i, 0, cin, true, ws, T, training, s))) return rc;
      if ((rc = conv1_range(n, b, i, 0, cin, false, fuse, ws, T, s))) return rc;
    }
    if (training && !fuse && !stats_done && (rc = launch_colstats(Y, g.cb, g.M, g.cb, sty, sty + g.cb, s))) return rc;
    if (raw) {                                                          // norm2: the 3x3 conv forms the affine itself; bookkeeping batched (below)
      if (i > 0 && (rc = add_batch(at<double>(ws, n.lws[b][i - 1].statY), at<double>(ws, n.lws[b][i - 1].statY) + g.cb, reg().layers[b][i - 1].bn2, 0, g.cb,
                                   at<float>(ws, n.lws[b][i - 1].bn2), g.cb, true)))
        return rc;                                                      // layer i-1's sums became final one layer ago: no wait for the flush
    } else if ((rc = launch_bn_finalize(sty, sty + g.cb, (double)g.M, F(T, L.bn2.w), F(T, L.bn2.b), F(T, L.bn2.rm), F(T, L.bn2.rv),
                                        static_cast<long long*>(T[L.bn2.nbt]), bn2, bn2 + g.cb, bn2 + 2 * g.cb, bn2 + 3 * g.cb, g.cb, training, s)))
      return rc;
    const float* w2p = n.opt_packed3x3 ? F(T, L.conv2) : at<float>(ws, W.w2p);     // else packed on the side stream at the start of forward
    if (b == 0 && i == 0) RDM_HIP_OK(hipStreamWaitEvent(s, n.ev_side, 0));
    FwdArgs c{};
    c.g = geom3x3(n.B, g.H, g.W, 1);
    c.A = Y; c.lda = g.cb; c.C = g.cb; c.a_scale = bn2; c.a_shift = bn2 + g.cb;
    c.Wt = w2p; c.wtap = (long)GROWTH * g.cb; c.ldw = g.cb;
    c.out = blk + cin; c.ldc = g.ctot; c.M = g.M; c.N = GROWTH;
    c.stat0 = bst + cin; c.stat1 = bst + g.ctot + cin;
    if (raw) {
      c.a_scale = nullptr; c.a_shift = nullptr;
      c.a_sum = sty; c.a_sq = sty + g.cb; c.a_gamma = F(T, L.bn2.w); c.a_beta = F(T, L.bn2.b); c.a_count = (double)g.M;
      c.accumulate = 1;                                                 // the slice was zeroed with the whole block at its start
      if (n.opt_fuse_stats3 && cdiv(g.M, 128) <= 128) c.tickets = at<unsigned>(ws, n.tickets);      // the last K split of a pixel tile takes the tile's statistics
    }
    if (n.wino_fwd[b] && !n.opt_no_wino) {
      // Winograd F(2x2, 3x3): the ordered reduction of the split partials also takes the channel statistics (no zero fill, no separate pass)
      WinoConv wv{};
      wv.A = Y; wv.lda = g.cb; wv.C = g.cb; wv.a_scale = bn2; wv.a_shift = bn2 + g.cb; wv.U = at<float>(ws, n.winoU[b][i]);
      wv.out = blk + cin; wv.ldc = g.ctot; wv.N = GROWTH; wv.B = n.B; wv.H = g.H; wv.W = g.W;
      wv.x6 = n.wino_x6(b);
      wv.split = wv.x6 ? n.wino_split_x6[b] : n.wino_split[b]; wv.partial = at<float>(ws, n.winoPartial); wv.partial_floats = n.winoPartialFloats;
      if (training) { wv.stat0 = bst + cin; wv.stat1 = bst + g.ctot + cin; }
      if ((rc = launch_conv3x3_wino_fwd(wv, s))) return rc;
    } else {
      const bool fuse2 = training && fuse_stats(g.M, GROWTH);
      if ((rc = launch_conv_fwd(c, false, fuse2 ? EPI_STORE_STATS : EPI_STORE, s)) < 0) return rc;
      if (training && !fuse2 && c.tickets == nullptr && (rc = launch_colstats(blk + cin, g.ctot, g.M, GROWTH, bst + cin, bst + g.ctot + cin, s))) return rc;
    }
    if (pipelined) RDM_HIP_OK(hipEventRecord(n.ev_fs[i & 1], s));     // layer i's output channels + their statistics are final
  }
  if (raw) {                                                            // the last layer's norm2, then whatever is left in the batch
    const int i = layers - 1;
    if ((rc = add_batch(at<double>(ws, n.lws[b][i].statY), at<double>(ws, n.lws[b][i].statY) + g.cb, reg().layers[b][i].bn2, 0, g.cb, at<float>(ws, n.lws[b][i].bn2), g.cb, true)))
      return rc;
    if ((rc = flush_batch())) return rc;
  }
  return 0;
}

int forward_transition(NetImpl& n, int t, void* ws, void* const* T, int training, hipStream_t s) {
  const BlockGeom& g = n.bg[t];
  const BlockGeom& gn = n.bg[t + 1];
  const int C = kTrans[t].cin, Co = kTrans[t].cout;
  float* blk = at<float>(ws, n.blk[t]);
  double* bst = at<double>(ws, n.blkstat[t]);
  float* bn = at<float>(ws, n.transBn[t]);
  const BnIdx& b = reg().trans_bn[t];
  // statistics over the zero-PADDED tensor (pad_br precedes the BatchNorm, RDM_Net.py:532)
  const double count = (double)n.B * (g.H + 1) * (g.W + 1);
  int rc = launch_bn_finalize(bst, bst + g.ctot, count, F(T, b.w), F(T, b.b), F(T, b.rm), F(T, b.rv), static_cast<long long*>(T[b.nbt]), bn,
                              bn + C, bn + 2 * C, bn + 3 * C, C, training, s);
  if (rc) return rc;
  float* P = at<float>(ws, n.transP[t]);
  if ((rc = launch_trans_pool(blk, g.ctot, bn, bn + C, P, n.B, g.H, g.W, C, s))) return rc;
  float* nblk = at<float>(ws, n.blk[t + 1]);
  double* nst = at<double>(ws, n.blkstat[t + 1]);
  FwdArgs a{};
  a.g = geom1x1(n.B, gn.H, gn.W);
  a.A = P; a.lda = C; a.C = C;
  a.Wt = F(T, reg().trans_conv[t]); a.wtap = 0; a.ldw = C;
  a.out = nblk; a.ldc = gn.ctot; a.M = gn.M; a.N = Co;
  a.stat0 = nst; a.stat1 = nst + gn.ctot;
  const bool fuse = training && fuse_stats(gn.M, Co);
  if ((rc = launch_conv_fwd(a, false, fuse ? EPI_STORE_STATS : EPI_STORE, s)) < 0) return rc;
  if (training && !fuse && (rc = launch_colstats(nblk, gn.ctot, gn.M, Co, nst, nst + gn.ctot, s))) return rc;
  return 0;
}

int zero_f32(float* p, size_t n, hipStream_t s) {
  if (int rc = launch_zero_rows(p, 1, (long)n, (long)n, s)) return rc;
  return 0;
}

// every layer of block b can run the split 1x1 input gradient (whose epilogue carries the deferred norm1 backward)?
static bool block_defers_norm1(const NetImpl& n, int b) {
  if (!n.xs_block(b)) return false;
  const BlockGeom& g = n.bg[b];
  for (int i = 0; i < kBlocks[b].layers; ++i) {
    FwdArgs e{};
    e.g = geom1x1(n.B, g.H, g.W);
    e.C = g.cb; e.N = kBlocks[b].cin + i * GROWTH; e.M = g.M;
    if (!xs_dgrad1x1_supported(e)) return false;
  }
  return true;
}

int backward_block(NetImpl& n, int b, int i_hi, int i_lo, bool join, void* ws, void* const* T, void* const* Gr, hipStream_t s) {
  const BlockGeom& g = n.bg[b];
  const int training = n.training_saved;
  float* blk = at<float>(ws, n.blk[b]);
  float* G = at<float>(ws, n.G[b]);
  float* dZ1 = at<float>(ws, n.dZ1);
  int rc;
  if ((rc = n.ensure_side())) return rc;
  hipStream_t side = n.side;
  for (int i = i_hi; i >= i_lo; --i) {
    const LayerIdx& L = reg().layers[b][i];
    const LayerWs& W = n.lws[b][i];
    const int cin = kBlocks[b].cin + i * GROWTH, cb = g.cb;
    const int par = i & 1;
    float* dZ = at<float>(ws, n.dZ[par]);
    float* Y = at<float>(ws, W.Y);
    float* bn1 = at<float>(ws, W.bn1);
    float* bn2 = at<float>(ws, W.bn2);
    const float* w2p = n.opt_packed3x3 ? F(T, L.conv2) : at<float>(ws, W.w2p);
    const float* go = G + cin;
    // the layer's output gradient `go` is final here (all later layers have accumulated into it)
    RDM_HIP_OK(hipEventRecord(n.ev_go, s));
    // ---- side stream: conv2 (3x3) wgrad ----
    if (Gr[L.conv2]) {
      // packed gradients go straight to the caller's tensor; otherwise into a scratch that is unpacked to OIHW afterwards
      float* dW3 = n.opt_packed3x3 ? F(Gr, L.conv2) : at<float>(ws, W.dw3);
      RDM_HIP_OK(hipStreamWaitEvent(side, n.ev_go, 0));
      WgradArgs xw{};
      xw.g = geom3x3(n.B, g.H, g.W, 1);
      xw.G = go; xw.ldg = g.ctot; xw.N = GROWTH; xw.Xs = Y; xw.ldx = cb; xw.C = cb; xw.x_scale = bn2; xw.x_shift = bn2 + cb;
      xw.dW = dW3; xw.wtap = (long)GROWTH * cb; xw.ldw = cb; xw.xsplit = n.xs_np();
      if (n.xs_block_wgrad3(b) && xs_wgrad3x3_supported(xw)) {
        // split-precision direct kernel: accumulates with f32 atomics into the zeroed gradient
        if (!(n.opt_packed3x3 && n.opt_prezeroed) && (rc = zero_f32(dW3, 9 * (size_t)GROWTH * cb, side))) return rc;
        if (n.opt_split_rows && n.xs_np() == 3) {
          // the 48-channel gradient as a frame image of split rows, written once: the kernel's 43 column blocks x K splits then stage their
          // gradient slabs verbatim instead of each deriving every row's pixel and splitting it
          void* gf = at<unsigned char>(ws, n.xsGf);
          if ((rc = launch_frame_split_rows(go, g.ctot, GROWTH, n.B, g.H, g.W, gf, side))) return rc;
          xw.G = static_cast<const float*>(gf); xw.ldg = 48; xw.g_frame = 1;
        }
        if ((rc = launch_xs_wgrad3x3(xw, side))) return rc;
      } else if (n.wino_wg[b] && !n.opt_no_wino) {
        // Winograd F(3x3, 2x2): writes the gradient (ordered split reduction, no atomics, no zero fill)
        WinoWgrad wv{};
        wv.G = go; wv.ldg = g.ctot; wv.N = GROWTH; wv.A = Y; wv.lda = cb; wv.C = cb; wv.a_scale = bn2; wv.a_shift = bn2 + cb;
        wv.dW = dW3; wv.wtap = (long)GROWTH * cb; wv.ldw = cb;
        wv.Vy = at<float>(ws, n.winoVy); wv.vy_floats = n.winoVyFloats; wv.part = at<float>(ws, n.winoQ); wv.part_floats = n.winoQFloats;
        wv.B = n.B; wv.H = g.H; wv.W = g.W;
        if ((rc = launch_conv3x3_wino_wgrad(wv, side))) return rc;
      } else {
        if (!(n.opt_packed3x3 && n.opt_prezeroed) && (rc = zero_f32(dW3, 9 * (size_t)GROWTH * cb, side))) return rc;
        WgradArgs w{};
        w.g = geom3x3(n.B, g.H, g.W, 1);
        w.G = go; w.ldg = g.ctot; w.N = GROWTH;
        w.Xs = Y; w.ldx = cb; w.C = cb; w.x_scale = bn2; w.x_shift = bn2 + cb;
        w.dW = dW3; w.wtap = (long)GROWTH * cb; w.ldw = cb;
        if ((rc = launch_conv_wgrad(w, side))) return rc;
      }
      if (!n.opt_packed3x3 && (rc = launch_unpack_w(dW3, F(Gr, L.conv2), GROWTH, cb, 9, GROWTH, side))) return rc;
    }
    // ---- main: conv2 dgrad -> dZ[par], gated by relu2, with the norm2 backward reductions ----
    if (n.dz_busy[par]) { RDM_HIP_OK(hipStreamWaitEvent(s, n.ev_dz[par], 0)); n.dz_busy[par] = false; }   // side wgrad still reading this buffer?
    double* s0 = at<double>(ws, W.bs2);
    double* s1 = s0 + cb;
    FwdArgs d{};
    d.g = geom3x3(n.B, g.H, g.W, -1);
    d.A = go; d.lda = g.ctot; d.C = GROWTH;
    d.Wt = w2p; d.wtap = (long)GROWTH * cb; d.ldw = cb;
    d.out = dZ; d.ldc = cb; d.M = g.M; d.N = cb;
    d.stat0 = s0; d.stat1 = s1; d.X = Y; d.ldx = cb; d.x_scale = bn2; d.x_shift = bn2 + cb;
    // mixed-precision arithmetic on the gradient GEMMs (RDM_NET_OPT_GEMM_BF16): when all three consumers / producers of the layer's dZ -> dY
    // tensor are the one-product kernels, it LIVES as bf16 (they round it to bf16 at staging anyway): its four passes move half the bytes
    bool dz_bf16 = false;
    {
      FwdArgs e1{};
      e1.g = geom1x1(n.B, g.H, g.W); e1.C = cb; e1.N = cin; e1.M = g.M;
      WgradArgs w1{};
      w1.g = geom1x1(n.B, g.H, g.W); w1.N = cb; w1.C = cin;
      dz_bf16 = n.xs_np() == 1 && n.xs_block(b) && xs_dgrad3x3_supported(d) && xs_dgrad1x1_supported(e1) && xs_wgrad1x1_supported(w1) && g.M >= 1024 && Gr[L.conv1] != nullptr;
    }
    d.out_bf16 = dz_bf16;
    // float32 mode (three products): dY leaves the norm2 backward as SPLIT ROWS when both of its consumers are the split kernels - they stage it
    // verbatim instead of each converting and splitting every element (RDM_NET_OPT_SPLIT_ROWS)
    bool dy_split = false, xh_split = false;
    {
      FwdArgs e1{};
      e1.g = geom1x1(n.B, g.H, g.W); e1.C = cb; e1.N = cin; e1.M = g.M;
      WgradArgs w1{};
      w1.g = geom1x1(n.B, g.H, g.W); w1.N = cb; w1.C = cin;
      const bool wg_ok = xs_wgrad1x1_supported(w1);
      dy_split = n.opt_split_rows && n.xs_np() == 3 && n.xs_block(b) && g.M >= 1024 && xs_dgrad1x1_supported(e1) && (Gr[L.conv1] == nullptr || wg_ok);
      xh_split = n.opt_split_rows && n.xs_np() == 3 && n.xs_block(b) && g.M >= 1024 && Gr[L.conv1] != nullptr && wg_ok;
    }
    if (n.xs_block(b) && xs_dgrad3x3_supported(d)) {
      const bool pk = n.pk_bwd_valid && !n.xsP3[b].empty();
      if ((rc = launch_xs_dgrad3x3(d, EPI_MASK_STATS, at<unsigned char>(ws, pk ? n.xsP3[b][i] : n.xsW), pk ? xs_dgrad3x3_workspace_bytes(cb) : n.xsWBytes, s, n.xs_np(), pk))) return rc;
    } else if ((rc = launch_conv_fwd(d, true, EPI_MASK_STATS, s)) < 0) return rc;      // split-K layers gate + reduce atomically
    // one elementwise pass dZ := dY (BN-backward coefficients computed in the same kernel).  Forming dY inside the conv1
    // dgrad / wgrad loaders instead was measured slower (heavier loaders cost the MFMA kernels more: 155 vs 164 img/s)
    if ((rc = launch_bn_bwd_apply(dZ, cb, dZ, cb, Y, cb, s0, s1, (double)g.M, F(T, L.bn2.w), bn2 + 2 * cb, bn2 + 3 * cb,
                                  Gr[L.bn2.w] ? F(Gr, L.bn2.w) : nullptr, Gr[L.bn2.b] ? F(Gr, L.bn2.b) : nullptr, g.M, cb, false, training, s, dz_bf16, dy_split)))
      return rc;
    // ---- side stream: conv1 (1x1) wgrad straight into the PyTorch-layout gradient ([cb][cin][1][1]) ----
    if (Gr[L.conv1]) {
      // (round 3: moving this launch to the caller's stream for the last 1-4 layers of dense_e2 / e3 - where the side stream has become the
      // longer one, 25.9 vs 22.3 ms of kernel time - changed nothing: 72.0-72.7 vs 72.2 ms; the step is throughput-bound, not balance-bound)
      RDM_HIP_OK(hipEventRecord(n.ev_dy, s));
      RDM_HIP_OK(hipStreamWaitEvent(side, n.ev_dy, 0));
      if (!n.opt_prezeroed && (rc = zero_f32(F(Gr, L.conv1), (size_t)cb * cin, side))) return rc;
      WgradArgs w{};
      w.g = geom1x1(n.B, g.H, g.W);
      w.G = dZ; w.ldg = cb; w.N = cb;
      w.Xs = blk; w.ldx = g.ctot; w.C = cin; w.x_scale = bn1; w.x_shift = bn1 + cin;
      w.dW = F(Gr, L.conv1); w.wtap = 0; w.ldw = cin;
      w.xsplit = n.xs_block(b) ? n.xs_np() : 0;
      w.g_bf16 = dz_bf16;
      w.g_split = dy_split;
      if (xh_split) {
        // relu1(norm1(x)) of this layer, activated and split ONCE (the GEMM re-stages its activation tile for each of its cb / 128 gradient tiles)
        float* xh = at<float>(ws, n.xsXh);
        if ((rc = launch_split_rows(blk, g.ctot, bn1, bn1 + cin, xh, cin, g.M, cin, side))) return rc;
        w.Xs = xh; w.ldx = cin; w.x_scale = nullptr; w.x_shift = nullptr; w.x_split = 1;
      }
      if ((rc = launch_conv_wgrad(w, side))) return rc;
      RDM_HIP_OK(hipEventRecord(n.ev_dz[par], side));
      n.dz_busy[par] = true;
    }
    // ---- main: conv1 dgrad -> dZ1, gated by relu1, with the norm1 backward reductions ----
    s0 = at<double>(ws, W.bs1);
    s1 = s0 + cin;
    FwdArgs e{};
    e.g = geom1x1(n.B, g.H, g.W);
    e.A = dZ; e.lda = cb; e.C = cb;
    e.Wt = F(T, L.conv1); e.wtap = 0; e.ldw = cin;
    e.out = dZ1; e.ldc = cin; e.M = g.M; e.N = cin;
    e.stat0 = s0; e.stat1 = s1; e.X = blk; e.ldx = g.ctot; e.x_scale = bn1; e.x_shift = bn1 + cin;
    e.a_bf16 = dz_bf16;
    e.a_split = dy_split;
    const bool xs_d1 = n.xs_block(b) && xs_dgrad1x1_supported(e);
    // deferred norm1 backward: decided once per BLOCK - a layer that did not take it would neither consume nor forward the running (b, c) sums
    const bool defer = xs_d1 && n.opt_defer_norm1 && block_defers_norm1(n, b);
    if (defer) { e.out = G; e.ldc = g.ctot; e.acc_scaled = 1; }          // the epilogue adds (gamma * rstd) * dz into the block gradient itself
    if (xs_d1) {
      const bool pk = n.pk_bwd_valid && !n.xsP1[b].empty();
      if ((rc = launch_xs_dgrad1x1(e, EPI_MASK_STATS, at<unsigned char>(ws, pk ? n.xsP1[b][i] : n.xsW), pk ? xs_dgrad1x1_workspace_bytes(cb, cin) : n.xsWBytes, s, n.xs_np(), pk))) return rc;
    } else if ((rc = launch_conv_fwd(e, true, EPI_MASK_STATS, s)) < 0) return rc;
    if (defer) {
      // ... the b * x + c terms of all layers are summed per channel and applied to the channels whose gradient is read next: the 48 the layer
      // below produced, or the block's input channels after its first layer
      float* bc = at<float>(ws, n.deferBC);
      float* b_in = bc + (size_t)((i + 1) & 1) * 2 * n.deferLd;
      float* b_out = bc + (size_t)(i & 1) * 2 * n.deferLd;
      if (i == kBlocks[b].layers - 1 && (rc = zero_f32(b_in, (size_t)2 * n.deferLd, s))) return rc;
      const int sc0 = i > 0 ? cin - GROWTH : 0, sn = i > 0 ? GROWTH : cin;
      if ((rc = launch_bn_bwd_defer(G, g.ctot, blk, g.ctot, s0, s1, (double)g.M, F(T, L.bn1.w), bn1 + 2 * cin, bn1 + 3 * cin,
                                    Gr[L.bn1.w] ? F(Gr, L.bn1.w) : nullptr, Gr[L.bn1.b] ? F(Gr, L.bn1.b) : nullptr, b_in, b_in + n.deferLd, b_out,
                                    b_out + n.deferLd, g.M, cin, sc0, sn, training, s)))
        return rc;
    } else if ((rc = launch_bn_bwd_apply(G, g.ctot, dZ1, cin, blk, g.ctot, s0, s1, (double)g.M, F(T, L.bn1.w), bn1 + 2 * cin, bn1 + 3 * cin,
                                         Gr[L.bn1.w] ? F(Gr, L.bn1.w) : nullptr, Gr[L.bn1.b] ? F(Gr, L.bn1.b) : nullptr, g.M, cin, true, training, s)))
      return rc;
  }
  // join: everything the side stream produced (weight gradients) is ordered before what the caller enqueues next.  A caller that
  // consumes gradients stage by stage (the data-parallel exchange) needs it after every stage; without one it is needed once per
  // segment only (RDM_NET_OPT_JOIN_PER_SEGMENT) and the dgrad chain of the next stage starts without draining the wgrad stream -
  // the dZ double buffers stay fenced by their own events (dz_busy) across stages
  if (join) {
    RDM_HIP_OK(hipEventRecord(n.ev_side, side));
    RDM_HIP_OK(hipStreamWaitEvent(s, n.ev_side, 0));
    n.dz_busy[0] = n.dz_busy[1] = false;
  }
  return 0;
}

int backward_transition(NetImpl& n, int t, void* ws, void* const* T, void* const* Gr, hipStream_t s) {
  const BlockGeom& g = n.bg[t];
  const BlockGeom& gn = n.bg[t + 1];
  const int C = kTrans[t].cin, Co = kTrans[t].cout;
  const int training = n.training_saved;
  float* blk = at<float>(ws, n.blk[t]);
  float* Gn = at<float>(ws, n.G[t + 1]);      // gradient wrt the transition output = first Co channels
  float* G = at<float>(ws, n.G[t]);
  float* P = at<float>(ws, n.transP[t]);
  float* bn = at<float>(ws, n.transBn[t]);
  float* dP = at<float>(ws, n.dP);
  double* s0 = at<double>(ws, n.transBs[t]);
  double* s1 = s0 + C;
  float* cA = at<float>(ws, n.cA); float* cB = at<float>(ws, n.cB); float* cC = at<float>(ws, n.cC);
  const BnIdx& b = reg().trans_bn[t];
  const int wi = reg().trans_conv[t];
  int rc;
  if (Gr[wi]) {
    if (!n.opt_prezeroed && (rc = zero_f32(F(Gr, wi), (size_t)Co * C, s))) return rc;
    WgradArgs w{};
    w.g = geom1x1(n.B, gn.H, gn.W);
    w.G = Gn; w.ldg = gn.ctot; w.N = Co;
    w.Xs = P; w.ldx = C; w.C = C;
    w.dW = F(Gr, wi); w.wtap = 0; w.ldw = C;
    if ((rc = launch_conv_wgrad(w, s))) return rc;
  }
  FwdArgs d{};
  d.g = geom1x1(n.B, gn.H, gn.W);
  d.A = Gn; d.lda = gn.ctot; d.C = Co;
  d.Wt = F(T, wi); d.wtap = 0; d.ldw = C;
  d.out = dP; d.ldc = C; d.M = gn.M; d.N = C;
  if ((rc = launch_conv_fwd(d, true, EPI_STORE, s)) < 0) return rc;
  if ((rc = launch_trans_pool_bwd_reduce(dP, blk, g.ctot, bn, bn + C, n.B, g.H, g.W, C, s0, s1, s))) return rc;
  const double count = (double)n.B * (g.H + 1) * (g.W + 1);
  if ((rc = launch_bn_bwd_coeffs(s0, s1, count, F(T, b.w), bn + 2 * C, bn + 3 * C, cA, cB, cC, Gr[b.w] ? F(Gr, b.w) : nullptr,
                                 Gr[b.b] ? F(Gr, b.b) : nullptr, C, training, s)))
    return rc;
  return launch_trans_pool_bwd_apply(dP, blk, g.ctot, bn, bn + C, cA, cB, cC, G, g.ctot, n.B, g.H, g.W, C, s);
}

}  // namespace
}  // namespace rdm

using namespace rdm;

extern "C" {

int rdm_net_num_tensors(void) { return (int)reg().t.size(); }
const char* rdm_net_tensor_name(int32_t i) { return (i >= 0 && i < (int)reg().t.size()) ? reg().t[i].name.c_str() : nullptr; }
int64_t rdm_net_tensor_numel(int32_t i) { return (i >= 0 && i < (int)reg().t.size()) ? reg().t[i].numel : -1; }
int rdm_net_tensor_is_param(int32_t i) { return (i >= 0 && i < (int)reg().t.size()) ? reg().t[i].is_param : -1; }

int rdm_net_segment_range(int32_t seg, int32_t* first, int32_t* last) {
  RDM_CHECK_ARG(seg >= 0 && seg < 4 && first && last, "segment must be 0..3");
  *first = reg().seg_first[seg]; *last = reg().seg_last[seg];
  return RDM_OK;
}

int rdm_net_create(int32_t batch, int32_t height, int32_t width, rdm_net** out) {
  RDM_CHECK_ARG(out != nullptr, "out is NULL");
  RDM_CHECK_ARG(batch >= 1 && height >= 33 && width >= 33, "need batch >= 1 and an image of at least 33x33 (got %d, %dx%d)", batch, height, width);
  RDM_CHECK_ARG((long)batch * height * width < (1L << 28), "geometry too large for 32-bit pixel indices");
  NetImpl* n = new NetImpl();
  n->B = batch; n->H0 = height; n->W0 = width;
  n->plan();
  *out = reinterpret_cast<rdm_net*>(n);
  return RDM_OK;
}

void rdm_net_destroy(rdm_net* net) { delete reinterpret_cast<NetImpl*>(net); }
size_t rdm_net_workspace_bytes(const rdm_net* net) { return net ? reinterpret_cast<const NetImpl*>(net)->total : 0; }

int rdm_net_set_option(rdm_net* net, int32_t option, int32_t value) {
  RDM_CHECK_ARG(net != nullptr, "net is NULL");
  NetImpl* n = reinterpret_cast<NetImpl*>(net);
  if (option == RDM_NET_OPT_PACKED_3X3) n->opt_packed3x3 = value != 0;
  else if (option == RDM_NET_OPT_GRADS_PREZEROED) n->opt_prezeroed = value != 0;
  else if (option == RDM_NET_OPT_DIRECT_3X3) n->opt_no_wino = value != 0;
  else if (option == RDM_NET_OPT_DETERMINISTIC) n->opt_det = value != 0;
  else if (option == RDM_NET_OPT_JOIN_PER_SEGMENT) n->opt_join_seg = value != 0;
  // (environment overrides exist in DEVELOPMENT builds only - RDM_DEV_VARIANTS=1, in-process A/B runs of bench.py; the shipped library reads no
  // environment: an option is what the caller set.  Thresholds are clamped to what layout() sized the pack buffers for.)
  else if (option == RDM_NET_OPT_SPLIT_BWD) {
    n->opt_split_bwd = value != 0;
#ifdef RDM_DEV_VARIANTS
    if (getenv("RDM_XS_MIN_PIXELS")) n->xs_min_pixels = std::max(XS_LAYOUT_MIN_PIXELS, atoi(getenv("RDM_XS_MIN_PIXELS")));
    if (getenv("RDM_XS_WG3_MIN")) n->xs_wg3_min_pixels = std::max(XS_LAYOUT_MIN_PIXELS, atoi(getenv("RDM_XS_WG3_MIN")));
#endif
  }
  else if (option == RDM_NET_OPT_PREPACK) {
    n->opt_prepack = value != 0;
#ifdef RDM_DEV_VARIANTS
    if (getenv("RDM_PREPACK")) n->opt_prepack = atoi(getenv("RDM_PREPACK")) != 0;
#endif
  }
  else if (option == RDM_NET_OPT_DEFER_NORM1) {
    n->opt_defer_norm1 = value != 0;
#ifdef RDM_DEV_VARIANTS
    if (getenv("RDM_DEFER_NORM1")) n->opt_defer_norm1 = atoi(getenv("RDM_DEFER_NORM1")) != 0;
#endif
  }
  else if (option == RDM_NET_OPT_GEMM_BF16) n->opt_gemm_bf16 = value == 1 ? 3 : value == 2 ? 1 : value == 3 ? 2 : 0;      // 1 = both, 2 = forward GEMMs only, 3 = gradient GEMMs only
  else if (option == RDM_NET_OPT_SPLIT_ROWS) n->opt_split_rows = value != 0;
  else if (option == RDM_NET_OPT_WINO_X6) n->opt_wino_x6 = value != 0;
  else if (option == RDM_NET_OPT_FUSE_STATS3) n->opt_fuse_stats3 = value != 0;
  else if (option == RDM_NET_OPT_SPLIT_FWD) {
    n->opt_split_fwd = value != 0;
#ifdef RDM_DEV_VARIANTS
    if (getenv("RDM_XF_MIN_PIXELS")) n->xf_min_pixels = std::max(XF_LAYOUT_MIN_PIXELS, atoi(getenv("RDM_XF_MIN_PIXELS")));
#endif
  }
  else { set_error("rdm_net_set_option: unknown option %d", option); return RDM_ERR_BAD_ARGUMENT; }
  return RDM_OK;
}

int rdm_net_output_hw(const rdm_net* net, int32_t* h, int32_t* w) {
  RDM_CHECK_ARG(net && h && w, "NULL argument");
  const NetImpl* n = reinterpret_cast<const NetImpl*>(net);
  *h = n->bg[3].H; *w = n->bg[3].W;
  return RDM_OK;
}

/* debug / test access to the plan's workspace layout: byte offset and float count of a named buffer
 * ("blk0".."blk3", "G0".."G3", "logits", "e1", "dZ", "dZ1", "Y<b>_<i>", "bn1_<b>_<i>", "bn2_<b>_<i>", "P<t>") */
int rdm_net_buffer(const rdm_net* net, const char* name, int64_t* offset_bytes, int64_t* numel) {
  RDM_CHECK_ARG(net && name && offset_bytes && numel, "NULL argument");
  const NetImpl& n = *reinterpret_cast<const NetImpl*>(net);
  int b = -1, i = -1;
  if (sscanf(name, "blk%d", &b) == 1 && b >= 0 && b < 4) { *offset_bytes = n.blk[b]; *numel = (int64_t)n.bg[b].M * n.bg[b].ctot; return RDM_OK; }
  if (sscanf(name, "G%d", &b) == 1 && b >= 0 && b < 4) { *offset_bytes = n.G[b]; *numel = (int64_t)n.bg[b].M * n.bg[b].ctot; return RDM_OK; }
  if (sscanf(name, "Y%d_%d", &b, &i) == 2 && b >= 0 && b < 4 && i >= 0 && i < kBlocks[b].layers) { *offset_bytes = n.lws[b][i].Y; *numel = (int64_t)n.bg[b].M * n.bg[b].cb; return RDM_OK; }
  if (sscanf(name, "bn1_%d_%d", &b, &i) == 2 && b >= 0 && b < 4 && i >= 0 && i < kBlocks[b].layers) { *offset_bytes = n.lws[b][i].bn1; *numel = 4 * (int64_t)(kBlocks[b].cin + i * GROWTH); return RDM_OK; }
  if (sscanf(name, "bn2_%d_%d", &b, &i) == 2 && b >= 0 && b < 4 && i >= 0 && i < kBlocks[b].layers) { *offset_bytes = n.lws[b][i].bn2; *numel = 4 * (int64_t)n.bg[b].cb; return RDM_OK; }
  if (sscanf(name, "P%d", &b) == 1 && b >= 0 && b < 3) { *offset_bytes = n.transP[b]; *numel = (int64_t)n.bg[b + 1].M * kTrans[b].cin; return RDM_OK; }
  if (!strcmp(name, "logits")) { *offset_bytes = n.logits; *numel = (int64_t)n.bg[3].M * 192; return RDM_OK; }
  if (!strcmp(name, "e1")) { *offset_bytes = n.e1; *numel = (int64_t)n.M1 * 96; return RDM_OK; }
  if (!strcmp(name, "dZ")) { *offset_bytes = n.dZ[0]; *numel = 0; for (int k = 0; k < 4; ++k) *numel = std::max<int64_t>(*numel, (int64_t)n.bg[k].M * n.bg[k].cb); return RDM_OK; }
  if (!strcmp(name, "dZ1")) { *offset_bytes = n.dZ1; *numel = 0; for (int k = 0; k < 4; ++k) *numel = std::max<int64_t>(*numel, (int64_t)n.bg[k].M * n.bg[k].ctot); return RDM_OK; }
  set_error("rdm_net_buffer: unknown buffer '%s'", name);
  return RDM_ERR_BAD_ARGUMENT;
}

/* the encoder's output (trans_e4, RDM_Net.py:94) of the last rdm_net_forward on this workspace, as (B,1056,h,w) float32 NCHW: the tensor
 * every decoder of the reference consumes (:103-125).  It lives in the first 1056 channels of the decoder block's NHWC buffer. */
int rdm_net_encoder_output(const rdm_net* net, const void* ws, size_t ws_bytes, float* out_nchw, rdm_stream_t stream) {
  RDM_CHECK_ARG(net && ws && out_nchw, "NULL argument");
  const NetImpl& n = *reinterpret_cast<const NetImpl*>(net);
  if (ws_bytes < n.total) { set_error("workspace too small: %zu < %zu", ws_bytes, n.total); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  const BlockGeom& g = n.bg[3];
  return launch_nhwc_to_nchw(reinterpret_cast<const float*>(static_cast<const char*>(ws) + n.blk[3]), g.ctot, out_nchw, n.B, kBlocks[3].cin, g.H * g.W, stream);
}

double rdm_net_forward_flops(const rdm_net* net) {
  if (!net) return 0;
  const NetImpl* n = reinterpret_cast<const NetImpl*>(net);
  double mac = (double)n->M1 * 96 * 147;
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < kBlocks[b].layers; ++i)
      mac += (double)n->bg[b].M * n->bg[b].cb * (kBlocks[b].cin + i * GROWTH) + (double)n->bg[b].M * GROWTH * n->bg[b].cb * 9;
  for (int t = 0; t < 3; ++t)   // algorithmic count: the reference convolves at the padded full resolution
    mac += (double)n->B * (n->bg[t].H + 1) * (n->bg[t].W + 1) * kTrans[t].cin * kTrans[t].cout;
  mac += (double)n->bg[3].M * 180 * 2208;
  return 2.0 * mac;
}

double rdm_net_backward_flops(const rdm_net* net) {
  if (!net) return 0;
  const NetImpl* n = reinterpret_cast<const NetImpl*>(net);
  return 2.0 * rdm_net_forward_flops(net) - 2.0 * (double)n->M1 * 96 * 147;   // no dgrad into the image
}

int rdm_net_forward(rdm_net* net, const float* x, void* const* T, void* ws, size_t ws_bytes, float* logits_nchw, int32_t training,
                    rdm_stream_t stream) {
  RDM_CHECK_ARG(net && x && T && ws && logits_nchw, "NULL argument");
  NetImpl& n = *reinterpret_cast<NetImpl*>(net);
  if (ws_bytes < n.total) { set_error("workspace too small: %zu < %zu", ws_bytes, n.total); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  RDM_CHECK_ARG(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
  for (int i = 0; i < (int)reg().t.size(); ++i)
    RDM_CHECK_ARG(T[i] != nullptr || reg().t[i].numel == 0, "tensor %d (%s) is NULL", i, reg().t[i].name.c_str());
  hipStream_t s = stream;
  n.training_saved = training;
  DetScope det(n.opt_det != 0);
  int rc;
  if (training) RDM_HIP_OK(hipMemsetAsync(at<char>(ws, n.stats_begin), 0, n.stats_end - n.stats_begin, s));
  // the 78 3x3 weights are re-packed to [tap][out][in] once per forward, off the critical path: on the
  // library's side stream, fenced against the caller's stream on both ends
  // (with RDM_NET_OPT_PACKED_3X3 the caller keeps them packed and these 78 launches do not exist)
  if ((rc = n.ensure_side())) return rc;
  RDM_HIP_OK(hipEventRecord(n.ev_go, s));
  RDM_HIP_OK(hipStreamWaitEvent(n.side, n.ev_go, 0));
  // split kernels: the three-way-split images of the conv1 weights the forward reads first (dense_e2's first layer starts ~0.3 ms from here)
  n.pk_fwd_valid = n.pk_bwd_valid = false;
  if (n.opt_prepack && n.opt_split_fwd && !n.opt_det) {
    for (int b = 0; b < 4; ++b)
      if (n.xf_block(b) && !n.xfP[b].empty())
        for (int i = 0; i < kBlocks[b].layers; ++i) {
          const int cin = kBlocks[b].cin + i * GROWTH;
          if ((rc = launch_xs_pack_w1_fwd(F(T, reg().layers[b][i].conv1), cin, n.bg[b].cb, cin, at<unsigned char>(ws, n.xfP[b][i]), n.side))) return rc;
        }
    RDM_HIP_OK(hipEventRecord(n.ev_pkf, n.side));
    n.pk_fwd_valid = true;
  }
  if (!n.opt_packed3x3)
    for (int b = 0; b < 4; ++b)
      for (int i = 0; i < kBlocks[b].layers; ++i)
        if ((rc = launch_pack_w(F(T, reg().layers[b][i].conv2), at<float>(ws, n.lws[b][i].w2p), GROWTH, n.bg[b].cb, 9, GROWTH, n.side))) return rc;
  for (int b = 0; b < 4; ++b)                                 // Winograd weight transforms U = G g G^T, off the critical path like the packing
    if (n.wino_fwd[b] && !n.opt_no_wino)
      for (int i = 0; i < kBlocks[b].layers; ++i) {
        const float* w2p = n.opt_packed3x3 ? F(T, reg().layers[b][i].conv2) : at<float>(ws, n.lws[b][i].w2p);
        if ((rc = launch_wino_weight(w2p, (long)GROWTH * n.bg[b].cb, n.bg[b].cb, GROWTH, n.bg[b].cb, at<float>(ws, n.winoU[b][i]), n.side, n.wino_x6(b)))) return rc;
      }
  RDM_HIP_OK(hipEventRecord(n.ev_side, n.side));
  // ... and, behind everything the forward waits for, the split / fragment-order images the BACKWARD's input-gradient kernels read (the weights
  // do not change between this forward and its backward): 108 launches that used to sit on the backward's dependent chain
  if (training && n.opt_prepack && n.opt_split_bwd && !n.opt_det) {
    for (int b = 3; b >= 0; --b)
      if (n.xs_block(b) && !n.xsP3[b].empty())
        for (int i = kBlocks[b].layers - 1; i >= 0; --i) {
          const int cin = kBlocks[b].cin + i * GROWTH, cb = n.bg[b].cb;
          const float* w2p = n.opt_packed3x3 ? F(T, reg().layers[b][i].conv2) : at<float>(ws, n.lws[b][i].w2p);
          if ((rc = launch_xs_pack_w3_dgrad(w2p, (long)GROWTH * cb, cb, cb, at<unsigned char>(ws, n.xsP3[b][i]), n.side))) return rc;
          if ((rc = launch_xs_pack_w1_dgrad(F(T, reg().layers[b][i].conv1), cin, cb, cin, at<unsigned char>(ws, n.xsP1[b][i]), n.side))) return rc;
        }
    RDM_HIP_OK(hipEventRecord(n.ev_pkb, n.side));
    n.pk_bwd_valid = true;
  }
  // stem: 7x7/s2 conv as im2col + GEMM (K = 147 padded to 160), bias, then 3x3/s2 max-pool
  if ((rc = launch_im2col_stem(x, at<float>(ws, n.patches), n.B, n.H0, n.W0, s))) return rc;
  RDM_HIP_OK(hipMemsetAsync(at<float>(ws, n.stem_wp), 0, 96 * 160 * sizeof(float), s));
  RDM_HIP_OK(hipMemcpy2DAsync(at<float>(ws, n.stem_wp), 160 * sizeof(float), T[reg().stem_w], 147 * sizeof(float), 147 * sizeof(float), 96,
                              hipMemcpyDeviceToDevice, s));
  {
    FwdArgs a{};
    a.g = ConvGeom{n.B, n.H1, n.W1, n.H1, n.W1, 1, 1, 1, 1, 0, 0, 1};
    a.A = at<float>(ws, n.patches); a.lda = 160; a.C = 160;
    a.Wt = at<float>(ws, n.stem_wp); a.wtap = 0; a.ldw = 160;
    a.out = at<float>(ws, n.e1); a.ldc = 96; a.M = n.M1; a.N = 96; a.bias = F(T, reg().stem_b);
    if ((rc = launch_conv_fwd(a, false, EPI_STORE, s)) < 0) return rc;
  }
  if ((rc = launch_maxpool3s2(at<float>(ws, n.e1), at<float>(ws, n.blk[0]), n.bg[0].ctot, at<unsigned char>(ws, n.argmax), n.B, n.H1, n.W1, 96, s)))
    return rc;
  if (training) {
    double* st = at<double>(ws, n.blkstat[0]);
    if ((rc = launch_colstats(at<float>(ws, n.blk[0]), n.bg[0].ctot, n.bg[0].M, 96, st, st + n.bg[0].ctot, s))) return rc;
  }
  for (int b = 0; b < 4; ++b) {
    if ((rc = forward_block(n, b, ws, T, training, s))) return rc;
    if (b < 3 && (rc = forward_transition(n, b, ws, T, training, s))) return rc;
  }
  // d_1.conv2: 1x1 2208 -> 180 + bias (RDM_Net.py:147,159)
  {
    const BlockGeom& g = n.bg[3];
    FwdArgs a{};
    a.g = geom1x1(n.B, g.H, g.W);
    a.A = at<float>(ws, n.blk[3]); a.lda = g.ctot; a.C = g.ctot;
    a.Wt = F(T, reg().conv2_w); a.wtap = 0; a.ldw = g.ctot;
    a.out = at<float>(ws, n.logits); a.ldc = 192; a.M = g.M; a.N = 180; a.bias = F(T, reg().conv2_b);
    if ((rc = launch_conv_fwd(a, false, EPI_STORE, s)) < 0) return rc;
    if ((rc = launch_nhwc_to_nchw(at<float>(ws, n.logits), 192, logits_nchw, n.B, 180, g.H * g.W, s))) return rc;
  }
  return RDM_OK;
}

/* ---------------------------------------------------------------------------------------------
 * reduced-precision forward (BASELINE config 2): bf16 weights / activations, f32 accumulation, eval-mode BatchNorm
 * --------------------------------------------------------------------------------------------- */
size_t rdm_net_bf16_weight_bytes(const rdm_net* net) { return net ? reinterpret_cast<const NetImpl*>(net)->bf_wtotal : 0; }
size_t rdm_net_bf16_workspace_bytes(const rdm_net* net) { return net ? reinterpret_cast<const NetImpl*>(net)->bf_total : 0; }
double rdm_net_bf16_forward_bytes(const rdm_net* net) { return net ? reinterpret_cast<const NetImpl*>(net)->bf_bytes : 0; }

int rdm_net_bf16_prepare(rdm_net* net, void* const* T, void* wbuf, size_t wbuf_bytes, rdm_stream_t stream) {
  RDM_CHECK_ARG(net && T && wbuf, "NULL argument");
  NetImpl& n = *reinterpret_cast<NetImpl*>(net);
  if (wbuf_bytes < n.bf_wtotal) { set_error("bf16 weight buffer too small: %zu < %zu", wbuf_bytes, n.bf_wtotal); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  RDM_CHECK_ARG(((uintptr_t)wbuf & 255) == 0, "bf16 weight buffer must be 256-byte aligned");
  for (int i = 0; i < (int)reg().t.size(); ++i)
    RDM_CHECK_ARG(T[i] != nullptr || reg().t[i].numel == 0, "tensor %d (%s) is NULL", i, reg().t[i].name.c_str());
  hipStream_t s = stream;
  int rc;
  auto affine = [&](const BnIdx& b, size_t off, int C, int Cp) {      // eval-mode BatchNorm folded to (scale, shift): running statistics; arrays Cp long
    float* d = at<float>(wbuf, off);
    return launch_bn_finalize(nullptr, nullptr, 1.0, F(T, b.w), F(T, b.b), F(T, b.rm), F(T, b.rv), nullptr, d, d + Cp, d + 2 * Cp, d + 3 * Cp, C, 0, s);
  };
  RDM_HIP_OK(hipMemsetAsync(wbuf, 0, n.bf_wtotal, s));        // the padded rows / columns / affines of the bottleneck widths are zeros
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < kBlocks[b].layers; ++i) {
      const LayerIdx& L = reg().layers[b][i];
      const NetImpl::Bf16Layer& W = n.bfl[b][i];
      const int cin = kBlocks[b].cin + i * GROWTH, cb = n.bg[b].cb, cbp = n.bf_cbp[b];
      if ((rc = launch_f32_to_bf16_rows(F(T, L.conv1), cin, at<char>(wbuf, W.w1), cin, cb, cin, cin, s))) return rc;
      if (n.bf_act3[b]) { if ((rc = launch_pack_w3_frag_bf16(F(T, L.conv2), at<char>(wbuf, W.w3), cb, cbp, n.opt_packed3x3, s))) return rc; }
      else if (n.opt_packed3x3) { if ((rc = launch_f32_to_bf16_rows(F(T, L.conv2), cb, at<char>(wbuf, W.w3), cbp, 9L * GROWTH, cb, cb, s))) return rc; }
      else if ((rc = launch_pack_w_bf16(F(T, L.conv2), at<char>(wbuf, W.w3), GROWTH, cb, cbp, 9, s))) return rc;
      if ((rc = affine(L.bn1, W.bn1, cin, cin))) return rc;
      if ((rc = affine(L.bn2, W.bn2, cb, cbp))) return rc;
    }
  for (int t = 0; t < 3; ++t) {
    if ((rc = launch_f32_to_bf16_rows(F(T, reg().trans_conv[t]), kTrans[t].cin, at<char>(wbuf, n.bf_wt[t]), kTrans[t].cin, kTrans[t].cout, kTrans[t].cin, kTrans[t].cin, s))) return rc;
    if ((rc = affine(reg().trans_bn[t], n.bf_tbn[t], kTrans[t].cin, kTrans[t].cin))) return rc;
  }
  if ((rc = launch_f32_to_bf16_rows(F(T, reg().stem_w), 147, at<char>(wbuf, n.bf_stem_w), 160, 96, 147, 160, s))) return rc;   // K 147 -> 160, zero padded
  if ((rc = launch_f32_to_bf16_rows(F(T, reg().conv2_w), 2208, at<char>(wbuf, n.bf_head_w), 2208, 180, 2208, 2208, s))) return rc;
  return RDM_OK;
}

int rdm_net_forward_bf16(rdm_net* net, const float* x, void* const* T, const void* wbuf, size_t wbuf_bytes, void* ws, size_t ws_bytes,
                         float* logits_nchw, rdm_stream_t stream) {
  RDM_CHECK_ARG(net && x && T && wbuf && ws && logits_nchw, "NULL argument");
  NetImpl& n = *reinterpret_cast<NetImpl*>(net);
  if (wbuf_bytes < n.bf_wtotal) { set_error("bf16 weight buffer too small: %zu < %zu", wbuf_bytes, n.bf_wtotal); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  if (ws_bytes < n.bf_total) { set_error("workspace too small: %zu < %zu", ws_bytes, n.bf_total); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  RDM_CHECK_ARG((((uintptr_t)ws | (uintptr_t)wbuf) & 255) == 0, "workspace and weight buffer must be 256-byte aligned");
  hipStream_t s = stream;
  void* wb = const_cast<void*>(wbuf);
  int rc;
  RDM_HIP_OK(hipMemsetAsync(at<char>(ws, n.bf_counters), 0, NetImpl::kBfCounters * sizeof(unsigned), s));     // tile tickets (self-resetting; a fresh workspace is not)
  // stem: 7x7/s2 as im2col + GEMM (K = 147 padded to 160) + bias, 3x3/s2 max-pool into the first 96 channels of block 0
  if ((rc = launch_im2col_stem_bf16(x, at<char>(ws, n.bf_patches), n.B, n.H0, n.W0, s))) return rc;
  {
    GemmBf16Args a{};
    a.X = at<char>(ws, n.bf_patches); a.ldx = 160; a.K = 160;
    a.W = at<char>(wb, n.bf_stem_w); a.ldw = 160;
    a.out = at<char>(ws, n.bf_e1); a.ldc = 96; a.M = n.M1; a.N = 96; a.bias = F(T, reg().stem_b);
    if ((rc = launch_gemm_bf16(a, false, s))) return rc;
  }
  if ((rc = launch_maxpool3s2_bf16(at<char>(ws, n.bf_e1), at<char>(ws, n.bf_blk[0]), n.bg[0].ctot, n.B, n.H1, n.W1, 96, s))) return rc;
  for (int b = 0; b < 4; ++b) {
    const BlockGeom& g = n.bg[b];
    unsigned short* blk = at<unsigned short>(ws, n.bf_blk[b]);
    for (int i = 0; i < kBlocks[b].layers; ++i) {
      const NetImpl::Bf16Layer& W = n.bfl[b][i];
      const int cin = kBlocks[b].cin + i * GROWTH;
      const float* bn1 = at<float>(wb, W.bn1);
      const float* bn2 = at<float>(wb, W.bn2);
      const int cbp = n.bf_cbp[b];
      GemmBf16Args a{};                                      // BN-ReLU -> 1x1 (cin -> cb) -> the 3x3's BN-ReLU in the epilogue (eval mode: known ahead)
      a.X = blk; a.ldx = g.ctot; a.K = cin; a.scale = bn1; a.shift = bn1 + cin;
      a.W = at<char>(wb, W.w1); a.ldw = cin;
      a.out = at<char>(ws, n.bf_Y); a.ldc = cbp; a.M = g.M; a.N = cbp; a.oscale = bn2; a.oshift = bn2 + cbp;
      a.partial = at<float>(ws, n.bf_partial); a.partial_floats = n.bf_partial_floats;
      if ((rc = launch_gemm_bf16(a, false, s))) return rc;
      if (n.bf_act3[b]) {                                    // 3x3 (cb -> 48) on the activated tensor, written in place behind the block's channels
        Conv3ActArgs c{};
        c.Y = at<char>(ws, n.bf_Y); c.ldy = cbp; c.C = cbp; c.Wimg = at<char>(wb, W.w3);
        c.out = blk + cin; c.ldc = g.ctot; c.B = n.B; c.H = g.H; c.W = g.W;
        c.partial = at<float>(ws, n.bf_partial); c.partial_floats = n.bf_partial_floats;
        c.counters = at<unsigned>(ws, n.bf_counters); c.n_counters = NetImpl::kBfCounters;
        if ((rc = launch_conv3x3_act_bf16(c, s))) return rc;
      } else {
        Conv3Bf16Args c{};
        c.Y = at<char>(ws, n.bf_Y); c.ldy = cbp; c.C = cbp;
        c.Wt = at<char>(wb, W.w3); c.wtap = (long)GROWTH * cbp; c.ldw = cbp;
        c.out = blk + cin; c.ldc = g.ctot; c.B = n.B; c.H = g.H; c.W = g.W; c.M = g.M;
        c.partial = at<float>(ws, n.bf_partial); c.partial_floats = n.bf_partial_floats;
        if ((rc = launch_conv3x3_bf16(c, s))) return rc;
      }
    }
    if (b < 3) {
      const BlockGeom& gn = n.bg[b + 1];
      const int C = kTrans[b].cin, Co = kTrans[b].cout;
      const float* bn = at<float>(wb, n.bf_tbn[b]);
      if ((rc = launch_trans_pool_bf16(blk, g.ctot, bn, bn + C, at<char>(ws, n.bf_P[b]), n.B, g.H, g.W, C, s))) return rc;
      GemmBf16Args a{};
      a.X = at<char>(ws, n.bf_P[b]); a.ldx = C; a.K = C;
      a.W = at<char>(wb, n.bf_wt[b]); a.ldw = C;
      a.out = at<char>(ws, n.bf_blk[b + 1]); a.ldc = gn.ctot; a.M = gn.M; a.N = Co;
      if ((rc = launch_gemm_bf16(a, false, s))) return rc;
    }
  }
  {
    const BlockGeom& g = n.bg[3];                            // d_1.conv2: 1x1 2208 -> 180 + bias, f32 logits
    GemmBf16Args a{};
    a.X = at<char>(ws, n.bf_blk[3]); a.ldx = g.ctot; a.K = g.ctot;
    a.W = at<char>(wb, n.bf_head_w); a.ldw = g.ctot;
    a.out = at<char>(ws, n.bf_logits); a.ldc = 192; a.M = g.M; a.N = 180; a.bias = F(T, reg().conv2_b);
    if ((rc = launch_gemm_bf16(a, true, s))) return rc;
    if ((rc = launch_nhwc_to_nchw(at<float>(ws, n.bf_logits), 192, logits_nchw, n.B, 180, g.H * g.W, s))) return rc;
  }
  return RDM_OK;
}

static int backward_head(NetImpl& n, const float* dlogits, void* ws, void* const* T, void* const* Gr, hipStream_t s) {
  int rc;
  RDM_HIP_OK(hipMemsetAsync(at<char>(ws, n.bwd_stats_begin), 0, n.bwd_stats_end - n.bwd_stats_begin, s));
  const BlockGeom& g = n.bg[3];
  float* dL = at<float>(ws, n.dL);
  if ((rc = launch_nchw_to_nhwc(dlogits, dL, 192, n.B, 180, g.H * g.W, s))) return rc;
  const int wi = reg().conv2_w, bi = reg().conv2_b;
  if (Gr[bi]) {
    double* t64 = at<double>(ws, n.tmp64);
    RDM_HIP_OK(hipMemsetAsync(t64, 0, 512 * sizeof(double), s));
    if ((rc = launch_colstats(dL, 192, g.M, 180, t64, nullptr, s))) return rc;
    if ((rc = launch_f64_to_f32(t64, F(Gr, bi), 180, s))) return rc;
  }
  if (Gr[wi]) {
    if (!n.opt_prezeroed && (rc = zero_f32(F(Gr, wi), (size_t)180 * 2208, s))) return rc;
    WgradArgs w{};
    w.g = geom1x1(n.B, g.H, g.W);
    w.G = dL; w.ldg = 192; w.N = 180;
    w.Xs = at<float>(ws, n.blk[3]); w.ldx = g.ctot; w.C = g.ctot;
    w.dW = F(Gr, wi); w.wtap = 0; w.ldw = g.ctot;
    if ((rc = launch_conv_wgrad(w, s))) return rc;
  }
  // dgrad needs a contracted extent that is a multiple of 16: 180 -> 192 zero rows
  if ((rc = launch_pack_w(F(T, wi), at<float>(ws, n.w2pad), 180, 2208, 1, 192, s))) return rc;
  FwdArgs d{};
  d.g = geom1x1(n.B, g.H, g.W);
  d.A = dL; d.lda = 192; d.C = 192;
  d.Wt = at<float>(ws, n.w2pad); d.wtap = 0; d.ldw = 2208;
  d.out = at<float>(ws, n.G[3]); d.ldc = g.ctot; d.M = g.M; d.N = g.ctot;
  if ((rc = launch_conv_fwd(d, true, EPI_STORE, s)) < 0) return rc;
  // d_1.conv1 is constructed but unused for id 1 (RDM_Net.py:156-157): no gradient
  return 0;
}

static int backward_stem(NetImpl& n, void* ws, void* const* T, void* const* Gr, hipStream_t s) {
  // stem: max-pool backward, bias gradient, weight gradient (no gradient into the image)
  int rc;
  float* gE1 = at<float>(ws, n.gE1);
  if ((rc = launch_maxpool3s2_bwd(at<float>(ws, n.G[0]), n.bg[0].ctot, at<unsigned char>(ws, n.argmax), gE1, n.B, n.H1, n.W1, 96, s))) return rc;
  if (Gr[reg().stem_b]) {
    double* t64 = at<double>(ws, n.tmp64);
    RDM_HIP_OK(hipMemsetAsync(t64, 0, 512 * sizeof(double), s));
    if ((rc = launch_colstats(gE1, 96, n.M1, 96, t64, nullptr, s))) return rc;
    if ((rc = launch_f64_to_f32(t64, F(Gr, reg().stem_b), 96, s))) return rc;
  }
  if (Gr[reg().stem_w]) {
    float* dWs = at<float>(ws, n.dWstem);
    if ((rc = zero_f32(dWs, 96 * 160, s))) return rc;
    WgradArgs w{};
    w.g = ConvGeom{n.B, n.H1, n.W1, n.H1, n.W1, 1, 1, 1, 1, 0, 0, 1};
    w.G = gE1; w.ldg = 96; w.N = 96;
    w.Xs = at<float>(ws, n.patches); w.ldx = 160; w.C = 160;
    w.dW = dWs; w.wtap = 0; w.ldw = 160;
    if ((rc = launch_conv_wgrad(w, s))) return rc;
    RDM_HIP_OK(hipMemcpy2DAsync(Gr[reg().stem_w], 147 * sizeof(float), dWs, 160 * sizeof(float), 147 * sizeof(float), 96, hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

static int backward_stage(NetImpl& n, int stage, const float* dlogits, void* ws, void* const* T, void* const* Gr, hipStream_t s) {
  const Registry::Stage& st = reg().stages[stage];
  int rc;
  if (stage == 0 && n.pk_bwd_valid) RDM_HIP_OK(hipStreamWaitEvent(s, n.ev_pkb, 0));      // the packed weights of the input-gradient kernels (long done)
  if (st.first_of_seg) {
    if (st.seg == 0) { if ((rc = backward_head(n, dlogits, ws, T, Gr, s))) return rc; }
    else if ((rc = backward_transition(n, st.block, ws, T, Gr, s))) return rc;          // seg 1 -> trans_e4 (t = 2) feeding dense_e4 (b = 2), ...
  }
  if ((rc = backward_block(n, st.block, st.i_hi, st.i_lo, st.last_of_seg || !n.opt_join_seg, ws, T, Gr, s))) return rc;
  if (st.last_of_seg && st.seg == 3 && (rc = backward_stem(n, ws, T, Gr, s))) return rc;
  return 0;
}

int rdm_net_num_backward_stages(void) { return (int)reg().stages.size(); }

int rdm_net_backward_stage_range(int32_t stage, int32_t* first, int32_t* last) {
  RDM_CHECK_ARG(stage >= 0 && stage < (int)reg().stages.size() && first && last, "stage must be 0..%d", (int)reg().stages.size() - 1);
  *first = reg().stages[stage].first_t; *last = reg().stages[stage].last_t;
  return RDM_OK;
}

int rdm_net_backward_stage(rdm_net* net, const float* dlogits, void* const* T, void* const* Gr, void* ws, size_t ws_bytes, int32_t stage,
                           rdm_stream_t stream) {
  RDM_CHECK_ARG(net && T && Gr && ws, "NULL argument");
  RDM_CHECK_ARG(stage >= 0 && stage < (int)reg().stages.size(), "stage must be 0..%d", (int)reg().stages.size() - 1);
  RDM_CHECK_ARG(stage != 0 || dlogits != nullptr, "dlogits is NULL");
  NetImpl& n = *reinterpret_cast<NetImpl*>(net);
  if (ws_bytes < n.total) { set_error("workspace too small: %zu < %zu", ws_bytes, n.total); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  DetScope det(n.opt_det != 0);
  return backward_stage(n, stage, dlogits, ws, T, Gr, stream);
}

int rdm_net_backward(rdm_net* net, const float* dlogits, void* const* T, void* const* Gr, void* ws, size_t ws_bytes, int32_t first_seg,
                     int32_t last_seg, rdm_stream_t stream) {
  RDM_CHECK_ARG(net && T && Gr && ws, "NULL argument");
  RDM_CHECK_ARG(first_seg >= 0 && last_seg <= 3 && first_seg <= last_seg, "segments must satisfy 0 <= first <= last <= 3");
  RDM_CHECK_ARG(first_seg != 0 || dlogits != nullptr, "dlogits is NULL");
  NetImpl& n = *reinterpret_cast<NetImpl*>(net);
  if (ws_bytes < n.total) { set_error("workspace too small: %zu < %zu", ws_bytes, n.total); return RDM_ERR_WORKSPACE_TOO_SMALL; }
  DetScope det(n.opt_det != 0);
  for (int st = 0; st < (int)reg().stages.size(); ++st) {
    const int seg = reg().stages[st].seg;
    if (seg < first_seg || seg > last_seg) continue;
    if (int rc = backward_stage(n, st, dlogits, ws, T, Gr, stream)) return rc;
  }
  return RDM_OK;
}

}  // extern "C"
