"""-m gpu: the reduced-precision forward (BASELINE config 2, "NYU-v2 batch=8 forward-only bf16"; reference: train.py:11,57-58
mixed precision, module.py:49-56 -> RDM_Net.py:70-103).

Tolerance, stated here because bf16 cannot meet the f32 path's 1e-4: 8 significant bits per stored activation
and weight, f32 accumulation, through 78 dense layers.  Against the float32 oracle / the reference-generated fixture:
  * logits:         max |d| <= 4 % of max |logits| and RMS(d) <= 1 % of RMS(logits)
  * probabilities:  mean |dP| <= 5e-3, max |dP| <= 0.15 (a 2-way softmax over a pair of logits a few percent apart)
  * ordinal counts: every pixel within +-3 of the reference count, mean |d count| <= 0.5
and at the BASELINE size (B=8, 228x304) and the KITTI geometry the same bounds against the ORACLE's eval-mode float32 forward on the
same batch (the CPU restatement pinned by the reference's fixtures - not the product's own f32 path, which is only a second witness
there) + per-sample independence.
Op level (no accumulation through depth): the bf16 kernels reproduce a torch f32 reference fed the SAME bf16-rounded operands to
2e-3 of the output's max (f32 accumulation, one rounding of the result)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from md_rdm_amd import filler
from oracle import rdm_net_cpu as onet

pytestmark = pytest.mark.gpu
U = filler.uniform


def oracle_eval_forward(x):
    """(decode, P, logits) of the oracle's eval-mode float32 forward (oracle/rdm_net_cpu.py, pinned by tests/test_oracle_net.py)."""
    with torch.no_grad():
        _, dec, P, lg = onet.forward(onet.new_state_dict(filler.state_value), torch.from_numpy(x), training=False)
    return np.asarray(dec), np.asarray(P), np.asarray(lg)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.set_num_threads(16)
    return torch.device("cuda:0")


def make_model(dev):
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    m = DepthEstimationNet()
    filler.fill_state_dict(m.state_dict())
    return m.to(dev).eval()


def head_bounds(dec, P, lg, ref_dec, ref_P, ref_lg):
    d = lg - ref_lg
    assert np.abs(d).max() <= 0.04 * np.abs(ref_lg).max(), (np.abs(d).max(), np.abs(ref_lg).max())
    assert np.sqrt((d ** 2).mean()) <= 0.01 * np.sqrt((ref_lg ** 2).mean()), (np.sqrt((d ** 2).mean()), np.sqrt((ref_lg ** 2).mean()))
    assert np.abs(P - ref_P).mean() <= 5e-3 and np.abs(P - ref_P).max() <= 0.15, (np.abs(P - ref_P).mean(), np.abs(P - ref_P).max())
    dc = np.abs(dec.astype(np.int64) - ref_dec.astype(np.int64))
    assert dc.max() <= 3 and dc.mean() <= 0.5, (dc.max(), dc.mean())


def test_bf16_eval_forward_226_vs_reference_fixture(dev, net_gold):
    m = make_model(dev).set_precision("bf16")
    x, _ = filler.synthetic_batch(1, 226, 226, seed=filler.MARGIN_SEEDS["eval226"])
    with torch.no_grad():
        yh, dec, P = m(torch.from_numpy(x).to(dev))
    assert dec.dtype == torch.int64 and P.dtype == torch.float64 and [tuple(t.shape) for t in yh] == [(1, 1, 1, 1), (1, 1, 2, 2), (1, 1, 4, 4), (1, 1, 8, 8)]
    lg = m._native_forward_bf16(torch.from_numpy(x).to(dev)).cpu().numpy()
    head_bounds(dec.cpu().numpy(), P.cpu().numpy(), lg, net_gold["eval226_decode_c"], net_gold["eval226_ord_labels"], net_gold["eval226_logits"])
    assert any("librdm_hip.so" in l for l in open("/proc/self/maps"))


def test_bf16_b8_228x304_vs_f32_path_and_sample_independence(dev):
    """BASELINE config 2 geometry: batch 8, 228x304."""
    B, H, W = 8, 228, 304
    x, _ = filler.synthetic_batch(B, H, W, seed=1234)
    xg = torch.from_numpy(x).to(dev)
    m = make_model(dev)
    with torch.no_grad():
        _, dec32, P32 = m(xg)
        lg32 = m._native_forward(xg).cpu().numpy()
        m.set_precision("bf16")
        _, dec, P = m(xg)
        lg = m._native_forward_bf16(xg).cpu().numpy()
        _, dec1, P1 = m(xg[5:6].contiguous())
    assert tuple(P.shape) == (B, 90, 8, 10) and torch.isfinite(P).all()
    odec, oP, olg = oracle_eval_forward(x)                      # BASELINE configs[1] against the ORACLE, whole batch (~10 s of CPU)
    head_bounds(dec.cpu().numpy(), P.cpu().numpy(), lg, odec, oP, olg)
    head_bounds(dec.cpu().numpy(), P.cpu().numpy(), lg, dec32.cpu().numpy(), P32.cpu().numpy(), lg32)      # second witness: the f32 HIP path
    assert torch.equal(dec, (P > 0.5).sum(1, keepdim=True))
    # eval BatchNorm: samples do not interact.  A batch of 1 takes other tiles / K-splits (another summation order, other bf16
    # roundings), so a sample alone agrees with itself in the batch to bf16 noise, not bit for bit ...
    assert (P[5:6] - P1).abs().max().item() <= 0.05 and (dec[5:6] - dec1).abs().max().item() <= 2
    # ... while the SAME launch configuration is deterministic (no atomics anywhere on the bf16 path): same bits on a second run
    with torch.no_grad():
        _, dec_again, P_again = m(xg)
    assert torch.equal(P, P_again) and torch.equal(dec, dec_again)


def test_bf16_kitti_geometry_352x1216_vs_f32_path(dev):
    """BASELINE configs[4]'s geometry on the bf16 path (what `--precision 16` validates on): dense_e2 has rows of 304 pixels, whose 3x3
    takes the 64-column rectangular tiles of conv3x3_act_bf16_kernel."""
    B, H, W = 2, 352, 1216
    x, _ = filler.synthetic_batch(B, H, W, seed=77)
    xg = torch.from_numpy(x).to(dev)
    m = make_model(dev)
    with torch.no_grad():
        _, dec32, P32 = m(xg)
        lg32 = m._native_forward(xg).cpu().numpy()
        m.set_precision("bf16")
        _, dec, P = m(xg)
        lg = m._native_forward_bf16(xg).cpu().numpy()
    odec, oP, olg = oracle_eval_forward(x)
    head_bounds(dec.cpu().numpy(), P.cpu().numpy(), lg, odec, oP, olg)
    head_bounds(dec.cpu().numpy(), P.cpu().numpy(), lg, dec32.cpu().numpy(), P32.cpu().numpy(), lg32)


def test_bf16_is_inference_only_and_tracks_weight_updates(dev):
    from md_rdm_amd import _lib
    m = make_model(dev).set_precision("bf16")
    x, _ = filler.synthetic_batch(1, 226, 226, seed=3)
    xg = torch.from_numpy(x).to(dev)
    m.train()
    with pytest.raises(_lib.RdmError):
        m(xg)
    m.eval()
    with torch.no_grad():
        a = m._native_forward_bf16(xg).clone()
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["d_1.conv2.bias"] += 1.0
        m.load_state_dict(sd)                       # invalidates the prepared copies
        b = m._native_forward_bf16(xg)
    np.testing.assert_allclose((b - a).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.parametrize("M,K,N,prologue,bias,ld_pad", [(1000, 96, 2736, True, False, 48), (34656, 336, 96, True, False, 0), (640, 2208, 180, False, True, 0),
                                                        (2280, 1056, 720, True, False, 96), (77, 160, 96, False, True, 0), (640, 1632, 384, True, False, 576),
                                                        (80, 2160, 384, True, False, 48), (34656, 96, 1056, True, False, 48), (33000, 208, 1060, False, False, 0)])
def test_gemm_bf16_operator(dev, M, K, N, prologue, bias, ld_pad):
    """rdm_gemm_bf16: out = bias + relu(X*scale+shift) @ W^T on bf16 operands vs torch f32 on the same bf16-rounded operands."""
    from md_rdm_amd import _lib
    L, st, P = _lib.lib(), _lib.stream(), _lib.ptr
    ldx = K + ld_pad
    X = torch.from_numpy(U(f"g.x{M}", (M, ldx), -2.0, 2.0)).to(dev).bfloat16()
    Wt = torch.from_numpy(U(f"g.w{N}", (N, K), -0.1, 0.1)).to(dev).bfloat16()
    sc = torch.from_numpy(U("g.sc", (K,), 0.5, 1.5)).to(dev) if prologue else None
    sh = torch.from_numpy(U("g.sh", (K,), -0.3, 0.3)).to(dev) if prologue else None
    bs = torch.from_numpy(U("g.b", (N,), -0.5, 0.5)).to(dev) if bias else None
    ws = torch.empty(8 * M * N * 4 if M <= 1024 else 256, dtype=torch.uint8, device=dev)      # few-row shapes also take the K-split path
    for out_f32, use_ws in ((0, 0), (1, 0), (0, 1)):
        ldc = N + (12 if out_f32 == 0 else 0)
        out = torch.full((M, ldc), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        _lib.check(L.rdm_gemm_bf16(P(X), ldx, K, P(sc), P(sh), P(Wt), K, P(bs), P(out), ldc, M, N, out_f32, P(ws) if use_ws else None,
                                   ws.numel() if use_ws else 0, st))
        A = X[:, :K].float()
        if prologue:
            A = torch.relu(A * sc + sh).bfloat16().float()          # the kernel rounds the normalised activation to bf16 once
        want = A @ Wt.float().t()
        if bias:
            want = want + bs
        got = out[:, :N].float()
        err = (got - want).abs().max().item()
        assert err <= (2e-3 if out_f32 else 6e-3) * want.abs().max().item(), (out_f32, err, want.abs().max().item())
        if ldc > N:
            assert torch.isnan(out[:, N:].float()).all()             # columns past N untouched


@pytest.mark.parametrize("B,H,W,Cc", [(2, 57, 76, 2736), (8, 29, 38, 1392), (3, 15, 19, 720), (2, 8, 10, 384), (1, 11, 38, 384)])
def test_conv3x3_bf16_operator(dev, B, H, W, Cc):
    """rdm_conv3x3_bf16 (BN-ReLU prologue, zero padding, 48 outputs written into a wider NHWC buffer) vs torch conv2d in f32 on the
    same bf16-rounded operands; channel counts 48*odd exercise the 16-channel tail slab, H/W the border masks and batch wrap."""
    from md_rdm_amd import _lib
    L, st, P = _lib.lib(), _lib.stream(), _lib.ptr
    M = B * H * W
    Y = torch.from_numpy(U(f"c3.y{H}", (B, H, W, Cc), -2.0, 2.0)).to(dev).bfloat16()
    w = torch.from_numpy(U(f"c3.w{Cc}", (48, Cc, 3, 3), -0.05, 0.05)).to(dev)
    wp = w.permute(2, 3, 0, 1).reshape(9, 48, Cc).contiguous().bfloat16()
    sc = torch.from_numpy(U("c3.sc", (Cc,), 0.5, 1.5)).to(dev)
    sh = torch.from_numpy(U("c3.sh", (Cc,), -0.3, 0.3)).to(dev)
    ldc = 96
    A = torch.relu(Y.float() * sc + sh).bfloat16().float().permute(0, 3, 1, 2)
    want = F.conv2d(A, wp.float().reshape(3, 3, 48, Cc).permute(2, 3, 0, 1), padding=1).permute(0, 2, 3, 1).reshape(M, 48)
    wsb = int(L.rdm_conv3x3_bf16_workspace_bytes(Cc, B, H, W))
    outs = []
    for ws_bytes in (0, wsb, 3 * M * 48 * 4):             # unsplit, the heuristic's K-split, a K-split squeezed into 3 slabs
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        out = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device=dev)
        _lib.check(L.rdm_conv3x3_bf16(P(Y), Cc, Cc, P(sc), P(sh), P(wp), C.c_void_p(out.data_ptr() + 2 * 16), ldc, B, H, W,
                                      P(ws) if ws_bytes else None, ws_bytes, st))
        got = out[:, 16:64].float()
        err = (got - want).abs().max().item()
        assert err <= 6e-3 * want.abs().max().item(), (ws_bytes, err, want.abs().max().item())
        assert torch.isnan(out[:, :16].float()).all() and torch.isnan(out[:, 64:].float()).all()
        outs.append(out[:, 16:64].clone())
    if wsb:                                                  # the split path is deterministic: same bits on a second run
        out2 = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device=dev)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.check(L.rdm_conv3x3_bf16(P(Y), Cc, Cc, P(sc), P(sh), P(wp), C.c_void_p(out2.data_ptr() + 2 * 16), ldc, B, H, W, P(ws), wsb, st))
        assert torch.equal(out2[:, 16:64].view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("B,H,W,Cc", [(2, 57, 76, 2736), (8, 29, 38, 1392), (3, 15, 19, 720), (2, 8, 10, 384), (1, 11, 38, 384), (1, 23, 90, 96),
                                      (5, 9, 13, 64), (1, 23, 304, 96), (2, 41, 260, 64), (1, 88, 304, 480)])
def test_conv3x3_act_bf16_operator(dev, op_census, B, H, W, Cc):
    """rdm_conv3x3_act_bf16 (input already activated, both operands by LDS-DMA, zero padding by out-of-range source offsets, K-split
    combined inside the launch by the tile's last workgroup) vs torch conv2d in f32 on the same bf16-rounded operands.  Channel counts
    48*odd are padded to a multiple of 32 (finite garbage behind the real channels: their weights are zero); the shapes take tiles of
    2..8 waves, ragged last tiles of an image, rows that wrap inside a 16-pixel fragment, with and without the split; rows of 260 / 304
    pixels (the dense_e2 of a 352x1216 input) take the 64-column rectangular tiles."""
    from md_rdm_amd import _lib
    L, st, P = _lib.lib(), _lib.stream(), _lib.ptr
    M = B * H * W
    Cp = (Cc + 31) // 32 * 32
    ldy = Cp + 8
    Yf = torch.full((B, H, W, ldy), 3.0, device=dev)
    Yf[..., :Cc] = torch.relu(torch.from_numpy(U(f"c3a.y{H}", (B, H, W, Cc), -2.0, 2.0)).to(dev))
    Y = Yf.bfloat16()
    w = torch.from_numpy(U(f"c3a.w{Cc}", (48, Cc, 3, 3), -0.05, 0.05)).to(dev)
    wimg = torch.empty(int(L.rdm_conv3x3_act_bf16_weight_bytes(Cc)), dtype=torch.uint8, device=dev)
    _lib.check(L.rdm_conv3x3_act_bf16_pack(P(w), Cc, P(wimg), st))
    want = F.conv2d(Y[..., :Cc].float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=1).permute(0, 2, 3, 1).reshape(M, 48)
    ldc = 96
    wsb = int(L.rdm_conv3x3_act_bf16_workspace_bytes(Cp, B, H, W))
    outs = []
    for ws_bytes in (0, wsb, 16384 + 2 * 512 * 48 * 4 * B * ((H * W + 127) // 128)):     # unsplit, the heuristic's split, a tight scratch
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        out = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device=dev)
        _lib.check(L.rdm_conv3x3_act_bf16(P(Y), ldy, Cp, P(wimg), C.c_void_p(out.data_ptr() + 2 * 16), ldc, B, H, W, P(ws) if ws_bytes else None, ws_bytes, st))
        got = out[:, 16:64].float()
        err = (got - want).abs().max().item()
        assert err <= 6e-3 * want.abs().max().item(), (ws_bytes, err, want.abs().max().item())
        assert torch.isnan(out[:, :16].float()).all() and torch.isnan(out[:, 64:].float()).all()
        outs.append(out[:, 16:64].clone())
        if ws_bytes:                                             # the tile counters are left at zero
            assert int(ws[:16384].view(torch.int32).abs().sum().item()) == 0
    out2 = torch.full((M, ldc), float("nan"), dtype=torch.bfloat16, device=dev)  # same bits on a second run of the split path
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    _lib.check(L.rdm_conv3x3_act_bf16(P(Y), ldy, Cp, P(wimg), C.c_void_p(out2.data_ptr() + 2 * 16), ldc, B, H, W, P(ws), wsb, st))
    assert torch.equal(out2[:, 16:64].view(torch.int16), outs[1].view(torch.int16))
    assert any(k.startswith("conv3x3_act_bf16_kernel/") for k in _lib.census())
    if W > 250:
        assert any(k.endswith("/rect") for k in _lib.census())


@pytest.mark.parametrize("M,K,N,ld_pad", [(34656, 144, 2752, 48), (2280, 1056, 736, 96), (285, 2064, 736, 48), (80, 2160, 384, 48),
                                          (34656, 96, 2752, 0), (20001, 336, 2736, 96), (8200, 240, 1040, 48), (24000, 352, 1100, 0), (34656, 288, 2752, 0),
                                          (8816, 456, 1392, 48)])
def test_gemm_bf16_act_operator(dev, op_census, M, K, N, ld_pad):
    """rdm_gemm_bf16_act: bf16(relu((relu(X*scale+shift) @ W^T) * out_scale + out_shift)) - the consumer's BatchNorm + ReLU applied to the
    f32 accumulator before the single bf16 rounding; direct epilogue and (few rows) the K-split + reduction path."""
    from md_rdm_amd import _lib
    L, st, P = _lib.lib(), _lib.stream(), _lib.ptr
    ldx = K + ld_pad
    X = torch.from_numpy(U(f"ga.x{M}", (M, ldx), -2.0, 2.0)).to(dev).bfloat16()
    Wt = torch.from_numpy(U(f"ga.w{N}", (N, K), -0.1, 0.1)).to(dev).bfloat16()
    sc = torch.from_numpy(U("ga.sc", (K,), 0.5, 1.5)).to(dev)
    sh = torch.from_numpy(U("ga.sh", (K,), -0.3, 0.3)).to(dev)
    osc = torch.from_numpy(U("ga.osc", (N,), -1.5, 1.5)).to(dev)
    osh = torch.from_numpy(U("ga.osh", (N,), -0.5, 0.5)).to(dev)
    A = torch.relu(X[:, :K].float() * sc + sh).bfloat16().float()
    want = torch.relu((A @ Wt.float().t()) * osc + osh)
    ws = torch.empty(8 * M * N * 4 if M <= 1024 else 256, dtype=torch.uint8, device=dev)
    for use_ws in (0, 1):
        out = torch.full((M, N + 8), float("nan"), dtype=torch.bfloat16, device=dev)
        _lib.check(L.rdm_gemm_bf16_act(P(X), ldx, K, P(sc), P(sh), P(Wt), K, P(osc), P(osh), P(out), N + 8, M, N, P(ws) if use_ws else None, ws.numel() if use_ws else 0, st))
        err = (out[:, :N].float() - want).abs().max().item()
        assert err <= 6e-3 * want.abs().max().item(), (use_ws, err, want.abs().max().item())
        assert torch.isnan(out[:, N:].float()).all()
    if (M, K) == (8816, 456):                                    # dense_e3: the 128 x 112 tile (two rounds of workgroups instead of three)
        assert "gemm_bf16_kernel/128x112" in _lib.census()
    if ((M + 255) // 256) * ((N + 95) // 96) >= 1024:            # the persistent panel kernel (dense_e2 shapes) ran
        assert any(k.startswith("gemm_panel_bf16_kernel/") for k in _lib.census())


@pytest.mark.parametrize("B,H,W,Cc", [(1, 29, 38, 1408), (8, 15, 19, 736), (8, 8, 10, 384)])
def test_conv3x3_bf16_operator_without_prologue(dev, B, H, W, Cc):
    """rdm_conv3x3_bf16 with scale = shift = NULL: the input is already activated (few-pixel blocks of rdm_net_forward_bf16)."""
    from md_rdm_amd import _lib
    L, st, P = _lib.lib(), _lib.stream(), _lib.ptr
    M = B * H * W
    Y = torch.relu(torch.from_numpy(U(f"c3n.y{H}", (B, H, W, Cc), -2.0, 2.0)).to(dev)).bfloat16()
    w = torch.from_numpy(U(f"c3n.w{Cc}", (48, Cc, 3, 3), -0.05, 0.05)).to(dev)
    wp = w.permute(2, 3, 0, 1).reshape(9, 48, Cc).contiguous().bfloat16()
    want = F.conv2d(Y.float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=1).permute(0, 2, 3, 1).reshape(M, 48)
    wsb = int(L.rdm_conv3x3_bf16_workspace_bytes(Cc, B, H, W))
    for ws_bytes in (0, wsb):
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
        out = torch.full((M, 48), float("nan"), dtype=torch.bfloat16, device=dev)
        _lib.check(L.rdm_conv3x3_bf16(P(Y), Cc, Cc, None, None, P(wp), P(out), 48, B, H, W, P(ws) if ws_bytes else None, ws_bytes, st))
        err = (out.float() - want).abs().max().item()
        assert err <= 6e-3 * want.abs().max().item(), (ws_bytes, err, want.abs().max().item())
