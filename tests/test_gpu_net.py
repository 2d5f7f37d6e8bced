"""-m gpu: DepthEstimationNet on the native plan vs (a) the fixtures produced by RUNNING the
reference and (b) the oracle on the same seeded inputs.

Forward tolerance: the north star's 1e-4 relative, ELEMENT-WISE: the HIP logits against the reference's logits at
atol = 1e-4 * max|ref|, the ordinal probabilities at 1e-4, every block tap on a strided (pixel, channel) lattice
(filler.tap_subsample, ~3 K values per tap stored by make_golden.py) at 1e-4 * max|tap|; ordinal indices BIT-EXACT,
asserted outright: the fixture inputs (filler.MARGIN_SEEDS) were chosen so that every ordinal pair decision of the
reference has a margin of 25x the f32 conv noise.
Backward: the loss surface is piecewise linear (ReLU, clamp); at B=2 the reference's own float32
gradients deviate from a float64 evaluation by up to ~10 % on some tensors because single ReLU
decisions flip (tests/test_oracle_net.py::test_f32_gradient_noise_floor documents it on CPU).  The
criterion is PER TENSOR: each tensor's error w.r.t. the float64 oracle is bounded by a multiple of THAT
tensor's own float32-oracle error (no pooled maximum), and each tensor's gradient norm is held to the
norm the reference itself produced (fixture train228_grad_norm, all 491 tensors)."""
import numpy as np
import pytest
import torch

from md_rdm_amd import filler
from oracle import rdm_net_cpu as onet

pytestmark = pytest.mark.gpu
TAPS = {"max_e1": ("blk0", 96), "dense_e2": ("blk0", 384), "dense_e3": ("blk1", 768), "dense_e4": ("blk2", 2112), "d1_dense": ("blk3", 2208)}
# element-wise taps: the block outputs above + the transition outputs (= the first channels of the next block's buffer)
SUBTAPS = dict(TAPS, trans_e2=("blk1", 192), trans_e3=("blk2", 384), trans_e4=("blk3", 1056))
CTOT = {"blk0": 384, "blk1": 768, "blk2": 2112, "blk3": 2208}
REL = 1e-4                                                     # BASELINE.json north_star: "1e-4 relative float tolerance"


def check_taps(m, gold, prefix):
    """Every tap, element-wise on the fixture's lattice, at 1e-4 of the tap's maximum (+ the three summary statistics as before)."""
    for tap, (buf, c) in SUBTAPS.items():
        v = m.debug_buffer(buf).view(-1, CTOT[buf])[:, :c]
        ref = gold[f"{prefix}_tapsub_{tap}"]
        got = filler.tap_subsample(v)
        assert got.shape == ref.shape, (tap, got.shape, ref.shape)
        np.testing.assert_allclose(got, ref, rtol=0, atol=REL * np.abs(ref).max(), err_msg=tap)
        if tap in TAPS:
            np.testing.assert_allclose(stats3(v), gold[f"{prefix}_tap_{tap}"], rtol=1e-4, atol=1e-6)


def check_taps_vs_oracle(m, taps):
    """The same element-wise lattice against the ORACLE's taps of the same input (oracle/rdm_net_cpu.py fills taps["sub_<tap>"])."""
    for tap, (buf, c) in SUBTAPS.items():
        v = m.debug_buffer(buf).view(-1, CTOT[buf])[:, :c]
        ref = taps["sub_" + tap]
        np.testing.assert_allclose(filler.tap_subsample(v), ref, rtol=0, atol=REL * np.abs(ref).max(), err_msg=tap)
        if tap in TAPS:
            np.testing.assert_allclose(stats3(v), taps[tap], rtol=1e-4, atol=1e-6, err_msg=tap)


def check_logits_vs_oracle(m, ref_nchw):
    ref = np.transpose(np.asarray(ref_nchw), (0, 2, 3, 1)).reshape(-1, ref_nchw.shape[1])
    np.testing.assert_allclose(hip_logits(m), ref, rtol=0, atol=REL * np.abs(ref).max())


def hip_logits(m):
    """(pixels, 180) logits of the last forward (the d_1.conv2 output the DORN head consumes; rows of 192 in the plan)."""
    return m.debug_buffer("logits").view(-1, 192)[:, :180].cpu().numpy()


def stats3(t):
    t = t.double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.abs().max().item()])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.set_num_threads(16)
    return torch.device("cuda:0")


def make_model(dev, train=True, deterministic=False):
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    m = DepthEstimationNet()
    m.deterministic = deterministic                    # RDM_NET_OPT_DETERMINISTIC: ordered reductions (set before the first forward)
    filler.fill_state_dict(m.state_dict())
    m = m.to(dev)
    return m.train() if train else m.eval()


SEED = filler.MARGIN_SEEDS


def check_head(dec, P, gold_dec, gold_P, gold_logits, logits=None):
    assert filler.dorn_unsafe_pairs(gold_logits) == 0          # the fixture was built with margins (make_golden.py refuses otherwise)
    np.testing.assert_array_equal(dec, gold_dec)               # ordinal indices: bit-exact, everywhere
    np.testing.assert_allclose(P, gold_P, rtol=0, atol=REL)
    if logits is not None:                                     # the conv stack's output itself, element by element (reference: NCHW)
        ref = np.transpose(gold_logits, (0, 2, 3, 1)).reshape(-1, gold_logits.shape[1])
        np.testing.assert_allclose(logits, ref, rtol=0, atol=REL * np.abs(ref).max())


def test_train_step_vs_reference_goldens(dev, net_gold):
    from md_rdm_amd import harness
    m = make_model(dev)
    x, y = filler.synthetic_batch(2, 228, 228, seed=SEED["train228"])
    loss, parts = harness.training_step(m, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    check_taps(m, net_gold, "train228")
    dec, P = parts["ord_depth_pred"].cpu().numpy(), parts["ord_label_pred"].detach().cpu().numpy()
    check_head(dec, P, net_gold["train228_decode_c"], net_gold["train228_ord_labels"], net_gold["train228_logits"], hip_logits(m))
    np.testing.assert_array_equal(parts["ord_y"].cpu().numpy(), net_gold["train228_ord_y"])
    for i in range(4):   # atol: log-domain values near log(1)=0 inherit the reference's own f32 geometric-mean rounding
        np.testing.assert_allclose(parts["fine_details"][i].detach().cpu().numpy(), net_gold[f"train228_yhat{i}"], rtol=1e-4, atol=5e-6)
    got = np.array([parts["mse"].item(), parts["fine_detail_loss"].item(), parts["ord_loss"].item(), loss.item()])
    np.testing.assert_allclose(got, net_gold["train228_losses"], rtol=1e-4)
    np.testing.assert_allclose(parts["final_depth"].detach()[:, :, :4, :4].cpu().numpy(), net_gold["train228_final_depth_corner"], rtol=1e-4, atol=1e-6)
    loss.backward()
    sd = m.state_dict()
    for k in net_gold.files:
        if k.startswith("train228_rm__"):
            n = k[len("train228_rm__"):]
            np.testing.assert_allclose(sd[n + ".running_mean"].cpu().numpy(), net_gold[k], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(sd[n + ".running_var"].cpu().numpy(), net_gold["train228_rv__" + n], rtol=1e-4)
            assert int(sd[n + ".num_batches_tracked"]) == 1
    params = dict(m.named_parameters())
    for n in ["d_1.conv1.weight", "d_1.conv1.bias", "weight_layer.f4"]:
        assert params[n].grad is None
    for n in ["weight_layer.d0", "weight_layer.f1", "weight_layer.f2", "weight_layer.f3", "d_1.conv2.bias"]:
        ref = net_gold["train228_grad__" + n]
        np.testing.assert_allclose(params[n].grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())


def test_eval_forward_226_vs_reference(dev, net_gold):
    m = make_model(dev, train=False)
    x, _ = filler.synthetic_batch(1, 226, 226, seed=SEED["eval226"])
    with torch.no_grad():
        yh, dec, P = m(torch.from_numpy(x).to(dev))
    check_head(dec.cpu().numpy(), P.cpu().numpy(), net_gold["eval226_decode_c"], net_gold["eval226_ord_labels"], net_gold["eval226_logits"], hip_logits(m))
    check_taps(m, net_gold, "eval226")
    for i in range(4):
        np.testing.assert_allclose(yh[i].cpu().numpy(), net_gold[f"eval226_yhat{i}"], rtol=1e-4, atol=5e-6)


def test_rectangular_228x304_head_vs_reference(dev, net_gold):
    m = make_model(dev)
    x, _ = filler.synthetic_batch(2, 228, 304, seed=SEED["train228x304"])
    with torch.no_grad():
        yh, dec, P = m(torch.from_numpy(x).to(dev))
    assert dec.shape == (2, 1, 8, 10)
    check_head(dec.cpu().numpy(), P.cpu().numpy(), net_gold["train228x304_decode_c"], net_gold["train228x304_ord_labels"], net_gold["train228x304_logits"], hip_logits(m))
    check_taps(m, net_gold, "train228x304")
    assert [tuple(t.shape) for t in yh] == [(2, 1, 1, 1), (2, 1, 2, 2), (2, 1, 4, 4), (2, 1, 8, 8)]      # documented generalisation


def test_gradients_per_tensor_vs_float64_oracle_and_reference_norms(dev, net_gold):
    """Backward parity, tensor by tensor (all 485 tensors that receive a gradient; nothing is pooled into one maximum):
    (a) relative L2 error w.r.t. the float64 oracle:  ||g_hip - g_f64|| / ||g_f64||  <=  2 x (the float32 ORACLE's own error on
        THAT tensor) + 1.5e-2;
    (b) gradient NORM against the norm the REFERENCE itself produced for that tensor (fixture train228_grad_norm, written by
        tests/golden/make_golden.py from the reference's autograd):  | ||g_hip|| - N_ref | / N_ref  <=  3 x (that tensor's own
        float32-oracle norm error | ||g_f32|| - ||g_f64|| | / ||g_f64||) + 1.5e-3   (the CPU oracle is held to 2e-3 flat).
    Why a floor next to the tensor's own error: the loss is piecewise linear and two float32 evaluations differ by discrete ReLU
    flips (tests/test_oracle_net.py::test_f32_gradient_noise_floor).  Measured on MI355X: every tensor sits at a relative L2
    error of 0.6-1.5 % in BOTH float32 evaluations (medians 7.8e-3 HIP / 7.3e-3 oracle, ratio median 0.99, p90 1.14) - except
    the handful the float32 oracle happens to evaluate without a flip (d_1 denselayer24: 2e-5), where its own error says nothing
    about the noise level; the floors are 2x the typical L2 level and 4x the p90 norm error.
    (b) cannot be tighter than geometry allows: | ||g'|| - ||g|| | / ||g|| = e cos(theta) + O(e^2) for an error of relative length e at
    angle theta to the gradient, and an error that is NOT systematically aligned with the gradient has |cos(theta)| ~ 1 / sqrt(numel):
    a tensor may also pass (b) with  e x 4 / sqrt(numel) + e^2  (4 sigma of a random direction) - 6.7e-3 for a 288-element BatchNorm bias
    at e = 2.6 %, 5e-5 (irrelevant next to the floor) for a 1 M-element weight, where only a SYSTEMATIC error could move the norm.
    Round 3 built the deterministic reduction mode the round-2 review asked for (RDM_NET_OPT_DETERMINISTIC: no K split, ordered
    statistics; test_deterministic_mode_is_bit_reproducible... below) and ran this criterion in it: the verdict is then the same on
    every run, but the floors cannot come down - a fixed, purely sequential float32 summation order has MORE rounding error than the
    split partial sums (which add pairwise), so it flips more ReLU decisions: seven BatchNorm tensors of 96-720 elements sit at
    e = 2.2-2.6 % with 1.7e-3 .. 6.2e-3 norm deviation (inside the geometric bound, outside any flat floor that would still mean
    something for the large tensors), d_1.denselayer24.conv1.weight at e = 1.55 %.  The noise is the discrete flip noise of ANY float32
    evaluation (the oracle's own float32 run shows it), not an ordering artefact.  This test therefore keeps measuring the SHIPPED mode
    with the floors of round 2; what the deterministic mode adds is bit-reproducibility, asserted separately."""
    from md_rdm_amd import harness
    K_L2, FLOOR_L2, K_NORM, FLOOR_NORM = 2.0, 1.5e-2, 3.0, 1.5e-3
    B = 2
    x, y = filler.synthetic_batch(B, 228, 228, seed=SEED["train228"])
    m = make_model(dev)
    loss, _ = harness.training_step(m, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    loss.backward()
    r32 = onet.training_step(onet.new_state_dict(filler.state_value), torch.from_numpy(x), y)
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in onet.new_state_dict(filler.state_value).items()}
    r64 = onet.training_step(sd64, torch.from_numpy(x).double(), y)
    assert abs(loss.item() - r64["loss_all"]) < 1e-4 * abs(r64["loss_all"])
    gold_norm = dict(zip([str(n) for n in net_gold["train228_grad_names"]], net_gold["train228_grad_norm"]))
    report = []
    for n, p in m.named_parameters():
        g64 = r64["grads"].get(n)
        if g64 is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n
            assert gold_norm[n] < 0, n                                        # the reference has no gradient there either
            continue
        g = p.grad.cpu().double()
        g32 = r32["grads"][n].double()
        n64 = g64.norm().item() + 1e-300
        report.append((n, (g - g64).norm().item() / n64, (g32 - g64).norm().item() / n64,
                       abs(g.norm().item() - gold_norm[n]) / gold_norm[n], abs(g32.norm().item() - n64) / n64, p.numel()))
    assert len(report) == 485, len(report)                                     # 491 parameter tensors minus d_1.conv1.{weight,bias} and the empty f4..f7
    import os
    if os.environ.get("RDM_GRAD_REPORT"):                                       # development aid: dump the per-tensor table
        import json
        with open(os.environ["RDM_GRAD_REPORT"], "w") as fh:
            json.dump(report, fh)
    bad = [t for t in report if t[1] > K_L2 * t[2] + FLOOR_L2 or t[3] > max(K_NORM * t[4] + FLOOR_NORM, t[1] * 4.0 / t[5] ** 0.5 + t[1] ** 2)]
    assert not bad, "per-tensor gradient parity failed for %d tensors (name, l2 hip, l2 f32, norm hip, norm f32, numel): %r" % (
        len(bad), sorted(bad, key=lambda t: -t[1])[:5])


def test_deterministic_mode_is_bit_reproducible_and_agrees_with_the_default_mode(dev):
    """RDM_NET_OPT_DETERMINISTIC: two train steps on the same inputs from the same weights - in two separate model instances - give
    BIT-IDENTICAL logits, gradients (all 485 tensors) and post-AdamW weights; against the default mode (split-K atomics, fused statistics,
    pipelined small blocks) the forward agrees to float32 rounding and every gradient tensor to the ReLU-flip noise level."""
    from md_rdm_amd import harness
    x, y = filler.synthetic_batch(2, 228, 228, seed=9)
    xa, ya = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    runs = []
    for det in (True, True, False):
        m = make_model(dev, deterministic=det)
        m.flatten_parameters()
        m.direct_grads = True
        opt = harness.FusedAdamW(m, lr=1e-4)
        opt.zero_grad()
        loss, parts = harness.training_step(m, xa, ya)
        loss.backward()
        grads = m._flat[1].clone()
        opt.step()
        torch.cuda.synchronize()
        runs.append((parts["ord_label_pred"].detach().clone(), grads, m._flat[0].clone()))
    (P0, g0, w0), (P1, g1, w1), (P2, g2, w2) = runs
    assert torch.equal(P0, P1) and torch.equal(g0, g1) and torch.equal(w0, w1)
    assert g0.abs().max() > 0
    assert (P0 - P2).abs().max().item() < 5e-4          # two float32 summation orders (measured 1.8e-4: the probabilities amplify the 1e-5 logit noise)
    assert ((g0 - g2).norm() / g0.norm()).item() < 5e-2    # ... and their ReLU-flip noise (measured 2.6e-2)


def test_direct_gradient_mode_and_fused_adamw(dev):
    from md_rdm_amd import harness
    x, y = filler.synthetic_batch(2, 228, 228, seed=9)
    xa, ya = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    a, b = make_model(dev), make_model(dev)
    b.flatten_parameters()
    b.direct_grads = True
    la, _ = harness.training_step(a, xa, ya)
    la.backward()
    lb, _ = harness.training_step(b, xa, ya)
    lb.backward()
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for n in ["encoder.conv_e1.weight", "encoder.dense_e3.denselayer5.norm2.weight", "d_1.conv2.weight"]:
        assert pb[n].grad.data_ptr() != 0 and (pa[n].grad - pb[n].grad).abs().max().item() <= 2e-2 * pa[n].grad.abs().max().item()
    flat0 = b._flat[0].clone()
    opt = harness.FusedAdamW(b, lr=1e-4)
    opt.step()
    ref = torch.optim.AdamW([p for p in a.parameters() if p.requires_grad], lr=1e-4)
    ref.step()
    for n in ["encoder.conv_e1.weight", "encoder.dense_e2.denselayer1.norm1.bias", "d_1.conv2.bias", "weight_layer.d0"]:
        np.testing.assert_allclose(pb[n].detach().cpu().numpy(), pa[n].detach().cpu().numpy(), rtol=0, atol=2.1e-4)   # first Adam step moves every weight by ~lr
    assert (b._flat[0] - flat0).abs().max().item() > 5e-5


def test_fused_adamw_three_steps_vs_torch_and_gradless_params_untouched(dev):
    """module.py:41 torch.optim.AdamW skips parameters whose .grad is None: d_1.conv1.* (RDM_Net.py:146,156-157) must stay
    bit-unchanged (no weight decay, no moments); every other tensor must follow torch's AdamW fed the SAME gradients."""
    from md_rdm_amd import harness
    x, y = filler.synthetic_batch(2, 228, 228, seed=9)
    xa, ya = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    a, b = make_model(dev), make_model(dev)
    b.flatten_parameters()
    b.direct_grads = True
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    init = {n: p.detach().clone() for n, p in pb.items()}
    opt = harness.FusedAdamW(b, lr=1e-4)
    assert len(opt.trainable_ranges()) == 2                      # [conv_e1 .. last dense layer], [d_1.conv2.*]; d_1.conv1.* lies between
    ref = torch.optim.AdamW([p for p in a.parameters() if p.requires_grad], lr=1e-4)
    for step in range(3):
        opt.zero_grad()
        loss, _ = harness.training_step(b, xa, ya)
        loss.backward()
        for n, p in pb.items():                                   # the reference optimiser sees exactly the gradients ours sees
            pa[n].grad = None if p.grad is None else p.grad.detach().clone()
        assert pb["d_1.conv1.weight"].grad is None and pb["d_1.conv1.bias"].grad is None
        opt.step()
        ref.step()
    worst = 0.0
    for n, p in pb.items():
        if n.startswith("d_1.conv1.") or p.numel() == 0:
            continue
        d = (p.detach() - pa[n].detach()).abs().max().item()
        worst = max(worst, d)
        assert d <= 2e-8 + 1e-6 * pa[n].detach().abs().max().item(), (n, d)
        assert (p.detach() - init[n]).abs().max().item() > 1e-5, n     # and it did move
    for n in ["d_1.conv1.weight", "d_1.conv1.bias"]:
        assert torch.equal(pb[n].detach(), init[n]), n                # bit-unchanged, as under torch.optim.AdamW
        assert torch.equal(pa[n].detach(), init[n]), n
    flat, _, entries = b._flat
    for k, p, o, n, g in entries:
        if k.startswith("d_1.conv1."):
            assert float(opt.m[o:o + n].abs().max()) == 0 and float(opt.v[o:o + n].abs().max()) == 0


def test_fused_adamw_leaves_frozen_encoder_bit_identical(dev):
    """freeze_encoder() (RDM_Net.py:65-67, module.py:38-41): frozen tensors get no gradient, so torch's AdamW never touches them."""
    from md_rdm_amd import harness
    x, y = filler.synthetic_batch(2, 228, 228, seed=9)
    xa, ya = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    m = make_model(dev)
    m.freeze_encoder()
    m.flatten_parameters()
    m.direct_grads = True
    init = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = harness.FusedAdamW(m, lr=1e-4)
    for _ in range(2):
        opt.zero_grad()
        loss, _ = harness.training_step(m, xa, ya)
        loss.backward()
        opt.step()
    moved = 0
    for n, p in m.named_parameters():
        if n.startswith("encoder.") or n.startswith("d_1.conv1."):
            assert p.grad is None and torch.equal(p.detach(), init[n]), n
        elif p.numel():
            moved += int((p.detach() - init[n]).abs().max().item() > 1e-5)
    assert moved > 90                                              # the decoder (24 layers x 6 tensors + conv2) and the 4 scalars train


def test_train_step_b16_228x304_vs_oracle(dev):
    """The BENCH geometry itself (BASELINE configs[2]: B=16, 228x304, full train step) against the CPU oracle on the same seeded
    batch - the step that selects the split-precision gradient kernels of dense_e2 / e3, the 256-pixel halo tiles and conv1x1_dma256_kernel.  Forward: block taps
    rtol 1e-4 + element-wise lattice at 1e-4 of the tap's max, logits element-wise at 1e-4 of their max, probabilities 1e-4, losses 1e-4, ordinal indices equal wherever the oracle's decision has
    the +-2.5e-4 margin (this input was not margin-searched; the unsafe pairs are counted and bounded).  Backward: the per-tensor
    criterion of the B=2 test - relative L2 error against the float64 oracle <= 2 x that tensor's own float32-oracle error + 1.5e-2,
    gradient norm against the float64 oracle's <= 3 x the float32 oracle's own norm error + 1.5e-3 (or the geometric bound)."""
    import time
    from md_rdm_amd import harness
    B, H, W = 16, 228, 304
    x, y = filler.synthetic_batch(B, H, W, seed=1234)
    m = make_model(dev)
    loss, parts = harness.training_step(m, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    t0 = time.time()
    taps = {}
    r32 = onet.training_step(onet.new_state_dict(filler.state_value), torch.from_numpy(x), y, taps=taps)
    t1 = time.time()
    check_taps_vs_oracle(m, taps)
    check_logits_vs_oracle(m, r32["logits"])
    P = parts["ord_label_pred"].detach().cpu().numpy()
    dec = parts["ord_depth_pred"].cpu().numpy()
    np.testing.assert_allclose(P, r32["P"], rtol=0, atol=REL)
    safe = filler.dorn_safe_mask(r32["logits"])                                # pair decisions that survive +-2.5e-4 on both logits
    assert safe.mean() > 0.99
    np.testing.assert_array_equal(((P > 0.5) & safe).sum(1, keepdims=True), ((r32["P"] > 0.5) & safe).sum(1, keepdims=True))
    assert np.abs(dec - r32["decode"]).max() <= int((~safe).sum(1).max())      # a count can only differ by its unsafe pairs
    np.testing.assert_array_equal(parts["ord_y"].cpu().numpy(), r32["ord_y"])
    got = np.array([parts["mse"].item(), parts["fine_detail_loss"].item(), parts["ord_loss"].item(), loss.item()])
    same_counts = np.array_equal(dec, r32["decode"])
    unsafe_px = int(((~safe).sum(1) > 0).sum())
    diff_px = int((dec != r32["decode"]).sum())
    print(f"[b16 oracle] ordinal count maps {'IDENTICAL' if same_counts else 'differ at %d pixels' % diff_px} ({unsafe_px} pixels hold an unsafe pair): "
          f"the mse / fine-detail / weight_layer comparisons run at {'1e-4' if same_counts else 'the count-perturbation bound'}")
    assert diff_px <= unsafe_px, (diff_px, unsafe_px)                            # a count may only move where one of its pairs is unsafe
    if same_counts:                                                             # the mse / fine-detail terms are functions of the integer counts
        np.testing.assert_allclose(got, [r32["mse"], r32["fine_detail_loss"], r32["ord_loss"], r32["loss_all"]], rtol=1e-4)
    else:
        # a count that moved by one at `diff_px` of the B*h*w pixels moves the two count-driven terms by at most that fraction of their scale
        frac = diff_px / dec.size
        np.testing.assert_allclose(got[:2], [r32["mse"], r32["fine_detail_loss"]], rtol=max(5.0 * frac, 1e-4))
    np.testing.assert_allclose(got[2], r32["ord_loss"], rtol=1e-4)
    sd64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in onet.new_state_dict(filler.state_value).items()}
    r64 = onet.training_step(sd64, torch.from_numpy(x).double(), y)
    t2 = time.time()
    print(f"[b16 oracle] f32 {t1 - t0:.1f} s, f64 {t2 - t1:.1f} s")
    K_L2, FLOOR_L2, K_NORM, FLOOR_NORM = 2.0, 1.5e-2, 3.0, 1.5e-3
    report = []
    for n, p in m.named_parameters():
        g64 = r64["grads"].get(n)
        if g64 is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0, n
            continue
        if n.startswith("weight_layer.") and not same_counts:
            print(f"[b16 oracle] {n}: gradient comparison skipped (count maps differ, see above)")
            continue                                                            # the 4 scalars see the counts, not the logits
        g, g32 = p.grad.cpu().double(), r32["grads"][n].double()
        n64 = g64.norm().item() + 1e-300
        report.append((n, (g - g64).norm().item() / n64, (g32 - g64).norm().item() / n64, abs(g.norm().item() - n64) / n64,
                       abs(g32.norm().item() - n64) / n64, p.numel()))
    assert len(report) >= 481
    import os
    if os.environ.get("RDM_GRAD_REPORT"):
        import json
        with open(os.environ["RDM_GRAD_REPORT"] + ".b16", "w") as fh:
            json.dump(report, fh)
    bad = [t for t in report if t[1] > K_L2 * t[2] + FLOOR_L2 or t[3] > max(K_NORM * t[4] + FLOOR_NORM, t[1] * 4.0 / t[5] ** 0.5 + t[1] ** 2)]
    assert not bad, "per-tensor gradient parity failed for %d tensors (name, l2 hip, l2 f32, norm hip, norm f32, numel): %r" % (
        len(bad), sorted(bad, key=lambda t: -t[1])[:5])


def test_full_size_properties_b16_228x304(dev):
    """BASELINE geometry (too slow for the CPU oracle): per-sample independence in eval mode,
    P in [0,1], decode == #{P > 0.5}, finite loss/gradients for a full train step."""
    from md_rdm_amd import harness
    B, H, W = 16, 228, 304
    x, y = filler.synthetic_batch(B, H, W, seed=1234)
    xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    m = make_model(dev, train=False)
    with torch.no_grad():
        _, dec, P = m(xg)
        _, dec1, P1 = m(xg[3:4].contiguous())
    assert P.min() >= 0 and P.max() <= 1 and torch.equal(dec, (P > 0.5).sum(1, keepdim=True))
    np.testing.assert_allclose(P[3:4].cpu().numpy(), P1.cpu().numpy(), atol=5e-5)       # eval BN: samples do not interact
    assert (dec[3:4] != dec1).sum().item() <= 2
    m.train()
    loss, parts = harness.training_step(m, xg, yg)
    loss.backward()
    assert torch.isfinite(loss)
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    assert dict(m.named_parameters())["encoder.conv_e1.weight"].grad.abs().max() > 0


def test_kitti_train_step_b2_352x1216_vs_oracle(dev):
    """BASELINE configs[4]'s geometry (352x1216: 88x304 / 44x152 / 22x76 / 11x38 maps; the Winograd kernels at 53 504 and 13 376 pixels, the
    direct ones below) as a train step against the CPU oracle at B=2: block taps and logits element-wise at 1e-4 of their maximum, probabilities 1e-4, ordinal indices equal wherever the
    oracle's pair decision has the margin, the ordinal loss 1e-4, every gradient tensor to the ReLU-flip noise level of a float32 evaluation
    (relative L2 <= 4e-2 against the oracle's float32 gradients, their norms within 2e-2)."""
    from md_rdm_amd import harness
    B, H, W = 2, 352, 1216
    x, y = filler.synthetic_batch(B, H, W, seed=4321)
    m = make_model(dev)
    loss, parts = harness.training_step(m, torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    loss.backward()
    torch.cuda.synchronize()
    taps = {}
    r32 = onet.training_step(onet.new_state_dict(filler.state_value), torch.from_numpy(x), y, taps=taps)
    check_taps_vs_oracle(m, taps)
    check_logits_vs_oracle(m, r32["logits"])
    P = parts["ord_label_pred"].detach().cpu().numpy()
    assert P.shape == (B, 90, 11, 38)
    np.testing.assert_allclose(P, r32["P"], rtol=0, atol=REL)
    safe = filler.dorn_safe_mask(r32["logits"])
    np.testing.assert_array_equal(((P > 0.5) & safe).sum(1, keepdims=True), ((r32["P"] > 0.5) & safe).sum(1, keepdims=True))
    np.testing.assert_allclose(parts["ord_loss"].item(), r32["ord_loss"], rtol=1e-4)
    worst = (0.0, "")
    for n, p in m.named_parameters():
        g32 = r32["grads"].get(n)
        if g32 is None or n.startswith("weight_layer."):
            continue
        g = p.grad.cpu().double()
        g32 = g32.double()
        e = ((g - g32).norm() / (g32.norm() + 1e-300)).item()
        worst = max(worst, (e, n))
        assert e < 4e-2, (n, e)
        assert abs(g.norm().item() - g32.norm().item()) <= 2e-2 * g32.norm().item() + 1e-12, n
    print("[kitti b2] worst gradient L2 vs the float32 oracle: %.3e (%s)" % worst)


def test_kitti_geometry_b8_352x1216(dev):
    """BASELINE config 5 (wide aspect, 11x38 head; the reference's own forward raises past the DORN head there): same
    size-independent properties, plus the 38-wide maps exercise the 256-pixel halo tiles with a 6-load halo run."""
    B, H, W = 8, 352, 1216
    x, y = filler.synthetic_batch(B, H, W, seed=77)
    xg = torch.from_numpy(x).to(dev)
    m = make_model(dev, train=False)
    with torch.no_grad():
        _, dec, P = m(xg)
        _, dec1, P1 = m(xg[5:6].contiguous())
    assert tuple(P.shape) == (B, 90, 11, 38) and tuple(dec.shape) == (B, 1, 11, 38)
    assert P.min() >= 0 and P.max() <= 1 and torch.equal(dec, (P > 0.5).sum(1, keepdim=True))
    np.testing.assert_allclose(P[5:6].cpu().numpy(), P1.cpu().numpy(), atol=5e-5)
    m.train()
    y_hat, x_d1, ord_labels = m(xg)
    (ord_labels.float() ** 2).mean().backward()                     # gradient of the conv stack through the DORN head
    g = dict(m.named_parameters())
    for n in ["encoder.conv_e1.weight", "encoder.dense_e2.denselayer3.conv2.weight", "encoder.dense_e4.denselayer30.norm1.weight", "d_1.conv2.weight"]:
        assert g[n].grad is not None and torch.isfinite(g[n].grad).all() and g[n].grad.abs().max() > 0, n


def test_state_dict_roundtrip_and_reflatten(dev):
    """Checkpoints interchange (968 reference keys); .to()/load_state_dict keep the native plan coherent."""
    a = make_model(dev, train=False)
    x, _ = filler.synthetic_batch(1, 228, 228, seed=3)
    xg = torch.from_numpy(x).to(dev)
    with torch.no_grad():
        _, da, Pa = a(xg)
    sd = {k: v.detach().cpu().clone() for k, v in a.state_dict().items()}
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    b = DepthEstimationNet()                     # random init, CPU
    b.load_state_dict(sd)
    b = b.to(dev).eval()
    with torch.no_grad():
        _, db, Pb = b(xg)
        b.load_state_dict(sd)                    # in-place copy into the flattened parameters
        b = b.to(dev)                            # no-op move must not detach the flat views
        _, dc, Pc = b(xg)
    # split-K f32 atomics make the summation order run-dependent: equal to float32 rounding, not bitwise
    assert (da != db).sum().item() <= 1 and (da != dc).sum().item() <= 1
    assert torch.allclose(Pa, Pb, atol=5e-6) and torch.allclose(Pa, Pc, atol=5e-6)


def test_several_steps_train_and_guard_against_stale_activations(dev):
    from md_rdm_amd import _lib, harness
    m = make_model(dev)
    m.flatten_parameters()
    m.direct_grads = True
    opt = harness.FusedAdamW(m, lr=1e-4)
    x, y = filler.synthetic_batch(2, 228, 228, seed=21)
    xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss, parts = harness.training_step(m, xg, yg)
        loss.backward()
        opt.step()
        losses.append(parts["ord_loss"].item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]          # the ordinal loss is what trains the convs
    # a second forward before backward would overwrite the saved activations: must fail loudly, not silently
    loss, _ = harness.training_step(m, xg, yg)
    with torch.no_grad():
        m(xg)
    with pytest.raises(_lib.RdmError):
        loss.backward()


def test_sixth_plan_of_a_process_steps_as_fast_as_the_first(dev):
    """Every plan used to own its weight-gradient stream; the runtime maps streams onto few hardware queues round-robin, and the fourth
    plan's stream shared the caller's queue: its weight gradients ran in series with the dgrad chain (71 ms instead of 52 ms at B=16
    228x304).  The stream is now one per device.  Six models in a row (each its own plan), same geometry: the last steps no slower than the
    first (10 % margin for clocks)."""
    import gc
    import time
    from md_rdm_amd import filler, harness
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    x, y = filler.synthetic_batch(8, 228, 304, seed=3)
    xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    ms = []
    for i in range(6):
        m = DepthEstimationNet()
        filler.fill_state_dict(m.state_dict())
        m = m.to(dev).train()
        m.flatten_parameters()

        def step():
            loss, _ = harness.training_step(m, xg, yg)
            loss.backward()
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / 4 * 1e3)
        del m, step
        gc.collect()
        torch.cuda.empty_cache()
    print("[plans in a row] ms per forward+backward:", [round(v, 2) for v in ms])
    assert max(ms[3:]) <= 1.10 * min(ms[:3]), ms


def test_deferred_norm1_backward_gives_the_same_gradients(dev):
    """RDM_NET_OPT_DEFER_NORM1 (default on): a * dz added by the 1x1 dgrad's epilogue, the b * x + c terms of all layers summed per channel and applied
    when a channel's gradient is read next - against the per-layer elementwise pass (defer_norm1 = False) on the same step.  Two runs of the SAME mode
    already differ (split-K atomics reorder the forward's float32 sums; ReLU flips carry that into the gradients - the deterministic-mode test
    measures 2.6e-2 of the gradient norm), so the criterion is relative to that floor, measured here: the distance between the two modes is no
    larger than three times the distance between two runs of one mode (+ 2e-3 of the norm), per block of the network - dense_e4 and d_1, whose
    backward the option does not touch, show what that ratio is by chance (measured 1.6-3.1 on all five).  B=4 228x304: dense_e2 and dense_e3
    are on the split kernels.  (The absolute check of the deferred path is the oracle comparison of the tests above, which run with it on.)"""
    from md_rdm_amd import filler, harness
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    x, y = filler.synthetic_batch(4, 228, 304, seed=filler.MARGIN_SEEDS["train228x304"])
    xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    runs = []
    for defer in (False, False, True):
        m = DepthEstimationNet()
        filler.fill_state_dict(m.state_dict())
        m = m.to(dev).train()
        m.defer_norm1 = defer
        m.flatten_parameters()
        loss, _ = harness.training_step(m, xg, yg)
        loss.backward()
        torch.cuda.synchronize()
        runs.append({n: p.grad.detach().double().clone() for n, p in m.named_parameters() if p.grad is not None})
    a, b, c = runs
    for prefix in ("encoder.conv_e1", "encoder.dense_e2", "encoder.dense_e3", "encoder.dense_e4", "d_1"):
        names = [n for n in a if n.startswith(prefix)]
        assert names, prefix
        norm = sum(float(a[n].pow(2).sum()) for n in names) ** 0.5
        floor = sum(float((a[n] - b[n]).pow(2).sum()) for n in names) ** 0.5 / norm
        dist = sum(float((a[n] - c[n]).pow(2).sum()) for n in names) ** 0.5 / norm
        print(f"[deferred norm1] {prefix}: two runs of one mode {floor:.2e} of the gradient norm, the two modes {dist:.2e}")
        assert dist <= 3.0 * floor + 2e-3, (prefix, floor, dist)


def test_in_launch_statistics_of_the_few_pixel_3x3_match_the_separate_pass(dev):
    """RDM_NET_OPT_FUSE_STATS3 (default on): in the training forward of dense_e4 / d_1 the K-split 3x3 convolution of a layer (torchvision
    _DenseLayer.conv2, RDM_Net.py:144,530) takes the channel statistics of its 48 outputs itself - the last split of a pixel tile to arrive reduces the
    finished tile - instead of a column-reduction launch.  Those sums feed norm1 of every later layer of the block: after one training forward from the
    same state the running statistics of every BatchNorm of the two blocks agree with the separate pass to 1e-4 of their maximum (a tile counted
    twice or missed would move a mean by 1 / 36 at dense_e4), the logits to 1e-3, and the census shows which form ran."""
    from md_rdm_amd import filler, _lib
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    x, _ = filler.synthetic_batch(4, 228, 304, seed=filler.MARGIN_SEEDS["train228x304"])
    xg = torch.from_numpy(x).to(dev)
    got = []
    for fuse in (False, True):
        m = DepthEstimationNet()
        filler.fill_state_dict(m.state_dict())
        m = m.to(dev).train()
        m.fuse_stats3 = fuse
        L = _lib.lib()
        L.rdm_census_enable(1)
        before = _lib.census()
        with torch.no_grad():
            m(xg)
        torch.cuda.synchronize()
        after = _lib.census()
        L.rdm_census_enable(0)
        names = [k for k in after if after[k] > before.get(k, 0) and k.startswith("conv3x3_halo_kernel/fwd/px128") and "rawbn" in k]
        assert names and all(k.endswith("/stats") == fuse for k in names), (fuse, names)
        bufs = {n: b.detach().double().cpu().clone() for n, b in m.named_buffers() if ("dense_e4" in n or n.startswith("d_1.")) and ("running_mean" in n or "running_var" in n)}
        got.append((bufs, m.debug_buffer("logits").double().cpu().clone()))
    (a, la), (b, lb) = got
    assert len(a) >= 2 * (36 + 24) * 2 - 8
    for n in a:
        assert float((a[n] - b[n]).abs().max()) <= 1e-4 * float(a[n].abs().max()) + 1e-7, n
    assert float((la - lb).abs().max()) <= 1e-3 * float(la.abs().max())
