"""-m gpu: the split-precision (bf16x3) gradient kernels of csrc/xsplit.hip through the C ABI vs float64 evaluations of the same operator.
Same tolerance as the exact-f32 MFMA kernels they replace in the backward pass of dense_e2 / dense_e3 (network/RDM_Net.py:526,528):
2e-5 of the tensor's maximum."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5
import conftest

_RAN = conftest.RAN.setdefault(__name__, set())


@pytest.fixture(scope="module", autouse=True)
def _census(op_census):
    yield


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


WGRAD1_CASES = [
    # B, H, W, C (in), ld, N (out), bn
    (16, 29, 38, 192, 768, 1392, True),      # dense_e3 conv1, first layer (RDM_Net.py:528): one 192-channel tile
    (4, 57, 76, 336, 384, 2736, True),       # dense_e2 conv1, last layer (RDM_Net.py:526): tiles 192 + 144, ragged row tile (2736 = 21 x 128 + 48)
    (3, 57, 57, 144, 144, 200, False),       # one 144-channel tile, N < two row tiles, no prologue, 9 747 pixels (ragged last slab: not a multiple of 32)
    (2, 75, 73, 240, 272, 136, True),        # tiles 144 + 96, ld > C
    (5, 41, 43, 720, 768, 96, True),         # dense_e3's last layer width: 4 tiles (192 x 3 + 144), one row tile
    (16, 15, 19, 2064, 2112, 720, True),     # dense_e4 conv1, last layer (RDM_Net.py:530): 11 column tiles, 4 560 pixels (142.5 slabs), deep K split
]


@pytest.mark.parametrize("case", WGRAD1_CASES, ids=[f"w1x1_{i}" for i in range(len(WGRAD1_CASES))])
def test_xs_wgrad1x1_vs_float64(case):
    """xs_wgrad1x1_kernel: every column-tile width (96 / 144 / 192), ragged row tiles, a ragged pixel tail, the BatchNorm + ReLU prologue, NaN
    behind the contracted channel prefix (nothing beyond C may reach a product), the launcher's K split and split_k = 1 and 5."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cc, ld, N, bn = case
    M = B * H * W
    g = torch.Generator().manual_seed(4000 + Cc)
    x = torch.randn(M, ld, generator=g)
    x[:, Cc:] = float("nan")
    gy = torch.randn(M, N, generator=g)
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.3
    a = (torch.relu(x[:, :Cc] * sc + sh) if bn else x[:, :Cc]).double()
    want = gy.double().t() @ a
    d = ConvDesc(B, H, W, Cc, ld, N, N, 1, 1, 1, 1, 0, 0)
    xg, gyg, scg, shg = x.to(dev), gy.to(dev), sc.to(dev), sh.to(dev)
    before = sum(v for k, v in _lib.census().items() if k.startswith("xs_wgrad1x1_kernel/"))
    for split in (0, 1, 5):
        dw = torch.zeros(N, Cc, device=dev)
        check(L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(gyg), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), split, 0, stream()))
        err = rel(dw.cpu().double(), want)
        assert err < TOL, (split, err)
    assert sum(v for k, v in _lib.census().items() if k.startswith("xs_wgrad1x1_kernel/")) == before + 3
    _RAN.add(("xs_wgrad1x1", case))


def test_xs_wgrad1x1_rejects_unsupported_shapes():
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    t = torch.zeros(64, 64, device=dev)
    d = ConvDesc(1, 8, 8, 64, 64, 64, 64, 1, 1, 1, 1, 0, 0)            # C = 64: not a multiple of 48
    assert L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(t), ptr(t), None, None, ptr(t), 0, 0, stream()) < 0


def _ref3x3_dgrad(gy, w9):
    """gy (B,H,W,N) f64, w9 (9,N,C) -> dx (B*H*W, C): the definition (shifted-slice matmuls, zero padding 1)"""
    import torch.nn.functional as F
    B, H, W, N = gy.shape
    gp = F.pad(gy, (0, 0, 1, 1, 1, 1))
    dx = torch.zeros(B * H * W, w9.shape[2], dtype=torch.float64)
    for r in range(3):
        for q in range(3):
            dx += gp[:, 2 - r:2 - r + H, 2 - q:2 - q + W, :].reshape(-1, N) @ w9[r * 3 + q]
    return dx


DGRAD3_CASES = [
    # B, H, W, Cb, ldg (the 48 gradient channels sit in a wider block-gradient buffer)
    (2, 57, 76, 336, 384),       # dense_e2's map (RDM_Net.py:526): 3 full column tiles + a 48-wide one, ragged last pixel tile
    (6, 29, 38, 1392, 96),       # dense_e3 at its real width (RDM_Net.py:528): 14 x 96 + 48
    (3, 9, 7, 96, 48),           # tiny rows: the halo spans several image rows and image boundaries inside one tile
    (1, 66, 127, 192, 48),       # wide rows (LDS image 384 slots)
    (16, 15, 19, 720, 2112),     # dense_e4 at the bench batch (RDM_Net.py:530): 144 work items for 512 persistent workgroups, 6 images in a tile
]


@pytest.mark.parametrize("case", DGRAD3_CASES, ids=[f"d3x3_{i}" for i in range(len(DGRAD3_CASES))])
def test_xs_dgrad3x3_vs_float64(case):
    """xs_dgrad3x3_kernel, plain and with the ReLU gate + BatchNorm-backward sums: image borders, batch wrap-around inside a tile, the
    ragged last pixel tile and the 48-wide last column tile."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cb, ldg = case
    N, M = 48, B * H * W
    g = torch.Generator().manual_seed(5000 + Cb)
    gyb = torch.randn(B, H, W, ldg, generator=g)
    gy = gyb[..., ldg - N:].contiguous()                      # the layer's slice: the LAST 48 channels of the buffer row
    w9 = torch.randn(9, N, Cb, generator=g) / (9 * N) ** 0.5
    y = torch.randn(M, Cb, generator=g)
    sc = torch.rand(Cb, generator=g) + 0.5
    sh = torch.randn(Cb, generator=g) * 0.3
    want = _ref3x3_dgrad(gy.double(), w9.double())
    d = ConvDesc(B, H, W, Cb, Cb, N, ldg, 3, 3, 1, 1, 1, 1)
    gyg, wg, yg, scg, shg = gyb.to(dev), w9.to(dev), y.to(dev), sc.to(dev), sh.to(dev)
    gptr = C.c_void_p(gyg.data_ptr() + (ldg - N) * 4)
    wsb = L.rdm_conv3x3_dgrad_x3_workspace_bytes(Cb)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dx = torch.full((M, Cb), float("nan"), device=dev)
    check(L.rdm_conv3x3_dgrad_x3(C.byref(d), gptr, ptr(wg), ptr(dx), Cb, None, 0, None, None, None, None, ptr(ws), wsb, 0, stream()))
    err = rel(dx.cpu().double(), want)
    assert err < TOL, err
    # gated: dz = dx * (y * sc + sh > 0); sums of dz and dz * y per channel
    s0 = torch.zeros(Cb, dtype=torch.float64, device=dev)
    s1 = torch.zeros_like(s0)
    dz = torch.full((M, Cb), float("nan"), device=dev)
    check(L.rdm_conv3x3_dgrad_x3(C.byref(d), gptr, ptr(wg), ptr(dz), Cb, ptr(yg), Cb, ptr(scg), ptr(shg), ptr(s0), ptr(s1), ptr(ws), wsb, 0, stream()))
    gate = (y * sc + sh) > 0
    wantz = want * gate
    assert rel(dz.cpu().double(), wantz) < TOL
    assert rel(s0.cpu(), wantz.sum(0)) < 1e-5 and rel(s1.cpu(), (wantz * y.double()).sum(0)) < 1e-5
    _RAN.add(("xs_dgrad3x3", case))


DGRAD1_CASES = [
    # B, H, W, K (= Cb, contracted), N (= cin, outputs), ldx (mask buffer row stride)
    (4, 57, 76, 2736, 336, 384),     # dense_e2 conv1, last layer (RDM_Net.py:526): two column tiles (11 + 10 sixteen-channel tiles), K = 85.5 steps (ragged)
    (16, 29, 38, 1392, 192, 768),    # dense_e3 conv1, first layer (RDM_Net.py:528): one column tile, 80-pixel tiles (an idle wave row)
    (3, 41, 43, 144, 96, 96),        # short K, 6 sixteen-channel tiles (3 + 3), 5 289 pixels: ragged last pixel tile
    (2, 33, 35, 1392, 720, 768),     # dense_e3's last layer width: 4 column tiles
    (16, 15, 19, 720, 2064, 2112),   # dense_e4 conv1, last layer (RDM_Net.py:530): 11 column tiles, K = 22.5 steps
]


@pytest.mark.parametrize("case", DGRAD1_CASES, ids=[f"d1x1_{i}" for i in range(len(DGRAD1_CASES))])
def test_xs_dgrad1x1_vs_float64(case):
    """xs_dgrad1x1_kernel, plain and with the ReLU gate + BatchNorm-backward sums, against a float64 product."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, K, N, ldx = case
    M = B * H * W
    g = torch.Generator().manual_seed(6000 + N)
    gy = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    x = torch.randn(M, ldx, generator=g)
    sc = torch.rand(N, generator=g) + 0.5
    sh = torch.randn(N, generator=g) * 0.3
    want = gy.double() @ w.double()
    d = ConvDesc(B, H, W, N, N, K, K, 1, 1, 1, 1, 0, 0)
    gyg, wg, xg, scg, shg = gy.to(dev), w.to(dev), x.to(dev), sc.to(dev), sh.to(dev)
    wsb = L.rdm_conv1x1_dgrad_x3_workspace_bytes(K, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dx = torch.full((M, N), float("nan"), device=dev)
    check(L.rdm_conv1x1_dgrad_x3(C.byref(d), ptr(gyg), ptr(wg), ptr(dx), N, None, 0, None, None, None, None, ptr(ws), wsb, 0, stream()))
    err = rel(dx.cpu().double(), want)
    assert err < TOL, err
    s0 = torch.zeros(N, dtype=torch.float64, device=dev)
    s1 = torch.zeros_like(s0)
    dz = torch.full((M, N), float("nan"), device=dev)
    check(L.rdm_conv1x1_dgrad_x3(C.byref(d), ptr(gyg), ptr(wg), ptr(dz), N, ptr(xg), ldx, ptr(scg), ptr(shg), ptr(s0), ptr(s1), ptr(ws), wsb, 0, stream()))
    xin = x[:, :N]
    gate = torch.addcmul(sh, xin, sc) > 0
    wantz = want * gate
    assert rel(dz.cpu().double(), wantz) < TOL
    assert rel(s0.cpu(), wantz.sum(0)) < 1e-5 and rel(s1.cpu(), (wantz * xin.double()).sum(0)) < 1e-5
    _RAN.add(("xs_dgrad1x1", case))


def F_pad_frame(gy, B, H, W):
    """(B, H, W, N) -> (B (H + 2) (W + 2), 48): the padded-frame ordering xs_wgrad3x3_kernel contracts over, zeros on the border and in channels >= N"""
    import torch.nn.functional as F
    N = gy.shape[-1]
    return F.pad(gy, (0, 48 - N, 1, 1, 1, 1)).reshape(-1, 48).contiguous()


def _ref3x3_wgrad(gy, a):
    import torch.nn.functional as F
    B, H, W, N = gy.shape
    Cc = a.shape[3]
    ap = F.pad(a, (0, 0, 1, 1, 1, 1))
    g2 = gy.reshape(-1, N).t().contiguous()
    return torch.stack([g2 @ ap[:, r:r + H, q:q + W, :].reshape(-1, Cc) for r in range(3) for q in range(3)])


WGRAD3_CASES = [
    # B, H, W, C, ld, N, ldg, bn
    (4, 57, 76, 336, 336, 48, 384, True),        # dense_e2's map (RDM_Net.py:526): ring runs 3 slabs ahead; 5 full column blocks + a 16-wide one
    (16, 29, 38, 1392, 1392, 48, 96, True),      # dense_e3 conv2 at the bench batch (RDM_Net.py:528): 21 full blocks + 48 channels; ring 2 slabs ahead
    (3, 20, 93, 80, 96, 40, 40, False),          # widest supported rows, no prologue, N < 48, ld > C
    (2, 9, 7, 64, 64, 48, 48, True),             # tiny frame: many border positions per slab
]


@pytest.mark.parametrize("case", WGRAD3_CASES, ids=[f"w3x3_{i}" for i in range(len(WGRAD3_CASES))])
def test_xs_wgrad3x3_vs_float64(case):
    """xs_wgrad3x3_kernel (padded-frame contraction, LDS ring, transposed reads at shifted rows) against shifted-slice float64 products: image
    borders, batch boundaries, ragged channel blocks, the launcher's K split and split_k = 1 / 3."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cc, ld, N, ldg, bn = case
    g = torch.Generator().manual_seed(7000 + Cc)
    x = torch.randn(B, H, W, ld, generator=g)
    x[..., Cc:] = float("nan")
    gyb = torch.randn(B, H, W, ldg, generator=g)
    gy = gyb[..., :N].contiguous()
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.3
    a = (torch.relu(x[..., :Cc] * sc + sh) if bn else x[..., :Cc]).double()
    want = _ref3x3_wgrad(gy.double(), a)
    d = ConvDesc(B, H, W, Cc, ld, N, ldg, 3, 3, 1, 1, 1, 1)
    xg, gyg, scg, shg = x.to(dev), gyb.to(dev), sc.to(dev), sh.to(dev)
    ref1 = None
    for split in (0, 1, 3):
        dw = torch.zeros(9, N, Cc, device=dev)
        check(L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(gyg), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), split, 0, stream()))
        err = rel(dw.cpu().double(), want)
        assert err < TOL, (split, err)
        if split == 1:
            ref1 = dw
    # the gradient operand as a FRAME IMAGE of split rows (round 5: what the plan hands the kernel): zeros on the frames' borders, the same two bf16
    # values per element - same tolerance, and with one workgroup per output tile (split 1: no race between atomic adds) the very same bits
    fb = int(L.rdm_frame_split_rows_bytes(B, H, W))
    frame = torch.full((fb // 4,), float("nan"), device=dev)
    check(L.rdm_frame_split_rows_f32(ptr(gyg), ldg, N, B, H, W, ptr(frame), stream()))
    fi = frame.view(torch.int16).view(-1, 12, 8).cpu()
    gp = F_pad_frame(gyb[..., :N], B, H, W)                                      # (U, 48) float32 with zero borders / zero channels >= N
    assert torch.equal(fi[:gp.shape[0]], _split_rows_ref(gp)) and int(fi[gp.shape[0]:].abs().max() if fi.shape[0] > gp.shape[0] else 0) == 0
    for split in (0, 1):
        dw = torch.zeros(9, N, Cc, device=dev)
        check(L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(frame), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), split, 0x40, stream()))
        assert rel(dw.cpu().double(), want) < TOL, ("frame", split)
        if split == 1:
            assert torch.equal(dw, ref1)
    _RAN.add(("xs_wgrad3x3", case))


FWD1_CASES = [
    # B, H, W, K (= Cin), ldx, N (= Cb), bn
    (4, 57, 76, 336, 384, 2736, True),       # dense_e2 conv1, last layer (RDM_Net.py:526): 15 column tiles (14 x 192 + 48), K = 10.5 steps, NaN behind the prefix
    (16, 29, 38, 192, 768, 1392, True),      # dense_e3 conv1, first layer (RDM_Net.py:528)
    (3, 41, 43, 96, 96, 208, False),         # no prologue, N = 13 sixteen-channel tiles (one full column tile + a 16-wide one), ragged pixel tile
    (2, 33, 35, 720, 768, 96, True),         # long K (22.5 steps), one narrow column tile
]


@pytest.mark.parametrize("case", FWD1_CASES, ids=[f"f1x1_{i}" for i in range(len(FWD1_CASES))])
def test_xs_fwd1x1_x6_vs_float64(case):
    """xs_fwd1x1_kernel (three-way split, six bf16 MFMAs per product) against a float64 product at the tolerance of the f32 MFMA kernel it can
    replace in the FORWARD pass (2e-5 of the maximum; measured ~1e-6: float32-equivalent), with and without the statistics epilogue."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, K, ld, N, bn = case
    M = B * H * W
    g = torch.Generator().manual_seed(8000 + K)
    x = torch.randn(M, ld, generator=g)
    x[:, K:] = float("nan")
    w = torch.randn(N, K, generator=g) / K ** 0.5
    sc = torch.rand(K, generator=g) + 0.5
    sh = torch.randn(K, generator=g) * 0.3
    a = (torch.relu(x[:, :K] * sc + sh) if bn else x[:, :K]).double()
    want = a @ w.double().t()
    d = ConvDesc(B, H, W, K, ld, N, N, 1, 1, 1, 1, 0, 0)
    xg, wg, scg, shg = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    wsb = L.rdm_conv1x1_fwd_x6_workspace_bytes(K, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    for stats in (True, False):
        y = torch.full((M, N), float("nan"), device=dev)
        ssum = torch.zeros(N, dtype=torch.float64, device=dev)
        ssq = torch.zeros_like(ssum)
        check(L.rdm_conv1x1_fwd_x6(C.byref(d), ptr(xg), ptr(wg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(y),
                                   ptr(ssum) if stats else None, ptr(ssq) if stats else None, ptr(ws), wsb, 0, stream()))
        err = rel(y.cpu().double(), want)
        assert err < 2e-6, err                       # float32-equivalent: an order of magnitude inside the f32 kernels' 2e-5 gate
        if stats:
            assert rel(ssum.cpu(), want.sum(0)) < 1e-5 and rel(ssq.cpu(), (want ** 2).sum(0)) < 1e-5
    _RAN.add(("xs_fwd1x1", case))


# ---- SPLIT ROWS (round 5): operands converted + split once by their producer, staged verbatim by the 1x1 gradient kernels ----------------------
def _split_rows_ref(v):
    """numpy / torch restatement of xsplit_dev.h: per four values [hi x4 | lo x4] bf16, hi = bf16(x) (round to nearest even), lo = bf16(x - hi);
    returned as the int16 view (M, C / 4, 8) of the row bytes."""
    hi = v.to(torch.bfloat16)
    lo = (v - hi.float()).to(torch.bfloat16)
    M, Cc = v.shape
    grp = torch.cat([hi.view(M, Cc // 4, 4), lo.view(M, Cc // 4, 4)], dim=2)      # (M, C/4, 8) bf16
    return grp.view(torch.int16)


def test_split_rows_producers_are_bit_exact():
    """rdm_split_rows_f32 (with and without the BatchNorm + ReLU prologue, ld > C on both sides) and rdm_bn_bwd(accumulate = 2) write exactly the
    two bf16 values per element the definition gives - the consumers then see the same operand bits as when they split the float32 value themselves."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    M, Cc, ld, ldd = 1037, 144, 160, 148
    x = torch.randn(M, ld, generator=g) * torch.logspace(-3, 3, ld)                 # a wide dynamic range: the lo parts matter
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    xg, scg, shg = x.to(dev), sc.to(dev), sh.to(dev)
    for bn in (False, True):
        dst = torch.full((M, ldd), float("nan"), device=dev)
        check(L.rdm_split_rows_f32(ptr(xg), ld, ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dst), ldd, M, Cc, stream()))
        # the kernel's prologue is ONE fused multiply-add per element (fmaf): product exact in float64, one rounding of the sum
        act = torch.relu((x[:, :Cc].double() * sc.double() + sh.double()).float())
        want = _split_rows_ref(act if bn else x[:, :Cc].contiguous())
        got = dst[:, :Cc].contiguous().cpu().view(torch.int16).view(M, Cc // 4, 8)
        assert torch.equal(got, want), bn
        assert torch.isnan(dst[:, Cc:]).all()                                       # nothing beyond the row's channels is touched
    # the norm2 backward writing dY as split rows == the float32 dY, split
    dz = torch.randn(M, Cc, generator=g).to(dev)
    s0 = dz.double().sum(0)
    s1 = (dz.double() * xg[:, :Cc].double()).sum(0)
    gamma, mean, rstd = (torch.rand(Cc, generator=g) + 0.5).to(dev), xg[:, :Cc].mean(0), 1.0 / (xg[:, :Cc].var(0, unbiased=False) + 1e-5).sqrt()
    outs = []
    for mode in (0, 2):
        dx = torch.full((M, Cc), float("nan"), device=dev)
        check(L.rdm_bn_bwd(ptr(dx), Cc, ptr(dz), Cc, ptr(xg), ld, ptr(s0), ptr(s1), float(M), ptr(gamma), ptr(mean), ptr(rstd), None, None, M, Cc, mode, 1, stream()))
        outs.append(dx.cpu())
    assert torch.equal(outs[1].view(torch.int16).view(M, Cc // 4, 8), _split_rows_ref(outs[0]))


@pytest.mark.parametrize("case", [WGRAD1_CASES[1], WGRAD1_CASES[3], WGRAD1_CASES[5]], ids=["rows_w1x1_e2", "rows_w1x1_ragged", "rows_w1x1_e4"])
def test_xs_wgrad1x1_on_split_rows(case):
    """The 1x1 weight gradient with its operands handed over as split rows (dY alone; dY and the activated input): same 2e-5 against float64,
    and the same result as the float32-operand launch up to the order of the K split's atomic adds (the MFMA operands are the same bits)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cc, ld, N, bn = case
    M = B * H * W
    g = torch.Generator().manual_seed(4100 + Cc)
    x = torch.randn(M, ld, generator=g)
    x[:, Cc:] = float("nan")
    gy = torch.randn(M, N, generator=g)
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    a = torch.relu(x[:, :Cc] * sc + sh).double()
    want = gy.double().t() @ a
    xg, gyg, scg, shg = x.to(dev), gy.to(dev), sc.to(dev), sh.to(dev)
    gy_rows = torch.empty(M, N, device=dev)
    check(L.rdm_split_rows_f32(ptr(gyg), N, None, None, ptr(gy_rows), N, M, N, stream()))
    x_rows = torch.empty(M, Cc, device=dev)                                          # contiguous [M][C]: the plan's layout for this operand
    check(L.rdm_split_rows_f32(ptr(xg), ld, ptr(scg), ptr(shg), ptr(x_rows), Cc, M, Cc, stream()))
    d_f32 = ConvDesc(B, H, W, Cc, ld, N, N, 1, 1, 1, 1, 0, 0)
    d_rows = ConvDesc(B, H, W, Cc, Cc, N, N, 1, 1, 1, 1, 0, 0)
    ref = torch.zeros(N, Cc, device=dev)
    check(L.rdm_conv2d_wgrad_x3(C.byref(d_f32), ptr(gyg), ptr(xg), ptr(scg), ptr(shg), ptr(ref), 1, 0, stream()))
    for flags, desc, xop, bnp in ((0x10, d_f32, xg, True), (0x30, d_rows, x_rows, False)):
        for split in (0, 1):
            dw = torch.zeros(N, Cc, device=dev)
            check(L.rdm_conv2d_wgrad_x3(C.byref(desc), ptr(gy_rows), ptr(xop), ptr(scg) if bnp else None, ptr(shg) if bnp else None, ptr(dw), split, flags, stream()))
            assert rel(dw.cpu().double(), want) < TOL, (flags, split)
            if split == 1:                                                           # one workgroup per output tile: no atomics race, same operand bits
                assert torch.equal(dw, ref), flags
    assert any(k.startswith("xs_wgrad1x1_kernel/") and k.endswith("/rowsGX") for k in _lib.census())
    assert L.rdm_conv2d_wgrad_x3(C.byref(d_rows), ptr(gy_rows), ptr(x_rows), ptr(scg), ptr(shg), ptr(ref), 0, 0x30, stream()) < 0      # activated rows + a prologue: refused
    assert L.rdm_conv2d_wgrad_x3(C.byref(d_rows), ptr(gy_rows), ptr(x_rows), None, None, ptr(ref), 0, 0x31, stream()) < 0              # rows in the bf16-operand mode: refused
    _RAN.add(("xs_wgrad1x1_rows", case))


def test_xs_dgrad1x1_on_split_rows_is_bit_identical():
    """The 1x1 input gradient fed dY as split rows: the very bits of the float32-operand launch (no atomics on its outputs), gate epilogue included,
    at dense_e2's and dense_e4's tile instantiations."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    for (B, H, W, K, N, ldx) in ((4, 57, 76, 2736, 336, 384), (16, 15, 19, 720, 432, 2112)):
        M = B * H * W
        g = torch.Generator().manual_seed(6100 + N)
        gy = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
        x = torch.randn(M, ldx, generator=g).to(dev)
        sc, sh = (torch.rand(N, generator=g) + 0.5).to(dev), (torch.randn(N, generator=g) * 0.3).to(dev)
        rows = torch.empty(M, K, device=dev)
        check(L.rdm_split_rows_f32(ptr(gy), K, None, None, ptr(rows), K, M, K, stream()))
        d = ConvDesc(B, H, W, N, N, K, K, 1, 1, 1, 1, 0, 0)
        wsb = L.rdm_conv1x1_dgrad_x3_workspace_bytes(K, N)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        out = []
        for flags, op in ((0, gy), (0x10, rows)):
            dz = torch.full((M, N), float("nan"), device=dev)
            s0 = torch.zeros(N, dtype=torch.float64, device=dev)
            s1 = torch.zeros_like(s0)
            check(L.rdm_conv1x1_dgrad_x3(C.byref(d), ptr(op), ptr(w), ptr(dz), N, ptr(x), ldx, ptr(sc), ptr(sh), ptr(s0), ptr(s1), ptr(ws), wsb, flags, stream()))
            out.append(dz)
        assert torch.equal(out[0], out[1]), (K, N)
        assert rel(out[1].cpu().double(), (gy.double() @ w.double()).cpu() * (torch.addcmul(sh, x[:, :N], sc) > 0).cpu()) < TOL
    assert any(k.startswith("xs_dgrad1x1_kernel/") and k.endswith("/rowsG") for k in _lib.census())


@pytest.mark.parametrize("geom", [(2, 9, 11, 96, 3, 96), (4, 15, 19, 384, 4, 720)], ids=["small", "dense_e4_width"])
def test_deferred_norm1_backward_operator_level(geom):
    """The deferred norm1 backward of a dense block (autograd of torchvision _DenseLayer.norm1 reached from network/RDM_Net.py:526-530) at OPERATOR
    level, with fixed inputs and a data dependence that makes the ORDER of application matter: layer i's bottleneck gradient is formed from the
    block gradient of the 48 channels it produced (dY_i = G[:, cin_i : cin_i + 48] @ R_i), so every b x + c term of the layers above must have
    reached that slice by the time layer i reads it.  The deferred sequence (1x1 input gradient with RDM_X3_ACC_SCALED, then rdm_bn_bwd_defer with
    its ping-pong sums, last layer first, a stage boundary in the middle) against the per-layer sequence it replaces (plain gate epilogue, then
    rdm_bn_bwd with accumulate = 1) to 1e-5 of the block gradient's maximum, and both against a float64 evaluation to 2e-5 x layers."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, C0, layers, Cb = geom
    GROWTH, M = 48, B * H * W
    ctot = C0 + layers * GROWTH
    g = torch.Generator().manual_seed(8800 + C0)
    x = torch.randn(M, ctot, generator=g)
    G0 = torch.randn(M, ctot, generator=g)
    Ws = [torch.randn(Cb, C0 + i * GROWTH, generator=g) / Cb ** 0.5 for i in range(layers)]
    Rs = [torch.randn(GROWTH, Cb, generator=g) / GROWTH ** 0.5 for i in range(layers)]
    gam = [torch.rand(C0 + i * GROWTH, generator=g) + 0.5 for i in range(layers)]
    bet = [torch.randn(C0 + i * GROWTH, generator=g) * 0.3 for i in range(layers)]
    mean, rstd, sc, sh = [], [], [], []
    for i in range(layers):
        cin = C0 + i * GROWTH
        xd = x[:, :cin].double()
        mu = xd.mean(0); var = xd.var(0, unbiased=False)
        rs = 1.0 / torch.sqrt(var + 1e-5)
        mean.append(mu.float()); rstd.append(rs.float())
        sc.append((gam[i].double() * rs).float()); sh.append((bet[i].double() - mu * gam[i].double() * rs).float())
    # ---- float64 evaluation (what k_bn_bwd_apply<true> computes, layer by layer, last layer first)
    Gr = G0.double().clone()
    want_dg, want_db = [None] * layers, [None] * layers
    for i in reversed(range(layers)):
        cin = C0 + i * GROWTH
        dY = Gr[:, cin:cin + GROWTH] @ Rs[i].double()
        xd = x[:, :cin].double()
        mask = (torch.addcmul(sh[i].double(), xd, sc[i].double()) > 0).double()
        dz = (dY @ Ws[i].double()) * mask
        xhat = (xd - mean[i].double()) * rstd[i].double()
        want_dg[i] = (dz * xhat).sum(0); want_db[i] = dz.sum(0)
        Gr[:, :cin] += gam[i].double() * rstd[i].double() * (dz - dz.mean(0) - xhat * (dz * xhat).mean(0))
    xg = x.to(dev)
    todev = lambda lst: [t.to(dev) for t in lst]
    Wg, Rg, gamg, meang, rstdg, scg, shg = todev(Ws), todev(Rs), todev(gam), todev(mean), todev(rstd), todev(sc), todev(sh)
    wsb = max(int(L.rdm_conv1x1_dgrad_x3_workspace_bytes(Cb, C0 + i * GROWTH)) for i in range(layers))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    results = {}
    for mode in ("per_layer", "deferred"):
        G = G0.to(dev).clone()
        dgs, dbs = [None] * layers, [None] * layers
        ld = (ctot + 63) // 64 * 64
        bc = torch.zeros(2, 2, ld, device=dev)                            # [parity][b | c][channel]: zeros = the sums above the block's last layer
        for i in reversed(range(layers)):
            cin = C0 + i * GROWTH
            dY = (G[:, cin:cin + GROWTH] @ Rg[i]).contiguous()           # reads the slice: it must be final here
            s0 = torch.zeros(cin, dtype=torch.float64, device=dev); s1 = torch.zeros_like(s0)
            dgs[i] = torch.empty(cin, device=dev); dbs[i] = torch.empty(cin, device=dev)
            d = ConvDesc(B, H, W, cin, cin, Cb, Cb, 1, 1, 1, 1, 0, 0)
            if mode == "per_layer":
                dz = torch.full((M, cin), float("nan"), device=dev)
                check(L.rdm_conv1x1_dgrad_x3(C.byref(d), ptr(dY), ptr(Wg[i]), ptr(dz), cin, ptr(xg), ctot, ptr(scg[i]), ptr(shg[i]), ptr(s0), ptr(s1), ptr(ws), wsb, 0, stream()))
                check(L.rdm_bn_bwd(ptr(G), ctot, ptr(dz), cin, ptr(xg), ctot, ptr(s0), ptr(s1), float(M), ptr(gamg[i]), ptr(meang[i]), ptr(rstdg[i]), ptr(dgs[i]), ptr(dbs[i]), M, cin, 1, 1, stream()))
            else:
                check(L.rdm_conv1x1_dgrad_x3(C.byref(d), ptr(dY), ptr(Wg[i]), ptr(G), ctot, ptr(xg), ctot, ptr(scg[i]), ptr(shg[i]), ptr(s0), ptr(s1), ptr(ws), wsb, 0x80, stream()))
                b_in, b_out = bc[(i + 1) & 1], bc[i & 1]
                c0, n = (cin - GROWTH, GROWTH) if i > 0 else (0, cin)
                check(L.rdm_bn_bwd_defer(ptr(G), ctot, ptr(xg), ctot, ptr(s0), ptr(s1), float(M), ptr(gamg[i]), ptr(meang[i]), ptr(rstdg[i]), ptr(dgs[i]), ptr(dbs[i]),
                                         ptr(b_in[0]), ptr(b_in[1]), ptr(b_out[0]), ptr(b_out[1]), M, cin, c0, n, 1, stream()))
            if i == layers // 2:
                torch.cuda.synchronize()                                  # a stage boundary of the staged backward (i_hi, i_lo): the running sums live across it
        torch.cuda.synchronize()
        results[mode] = (G.cpu().double(), [t.cpu().double() for t in dgs], [t.cpu().double() for t in dbs])
    scale = Gr.abs().max().item()
    for mode, (G, dgs, dbs) in results.items():
        assert torch.isfinite(G).all(), mode
        assert (G - Gr).abs().max().item() / scale < TOL * layers, (mode, (G - Gr).abs().max().item() / scale)
        for i in range(layers):
            assert rel(dgs[i], want_dg[i]) < TOL * layers and rel(dbs[i], want_db[i]) < TOL * layers, (mode, i)
    assert (results["deferred"][0] - results["per_layer"][0]).abs().max().item() / scale < 1e-5
    # the slices the deferred form finalises late differ from "never applied" by far more than the tolerance: the check above has teeth
    assert (Gr - G0.double())[:, :C0].abs().max().item() / scale > 1e-2
