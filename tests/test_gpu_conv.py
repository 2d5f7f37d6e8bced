"""-m gpu: the fp32 MFMA implicit-GEMM family through the C ABI vs PyTorch-CPU conv2d + autograd
(fp32 reference of the same op).  Tolerance 2e-5 relative to the tensor's max (f32 accumulation order)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5
import conftest

_RAN = conftest.RAN.setdefault(__name__, set())          # headline-size parity cases that ran in this process (tests/test_gpu_zz_coverage.py needs all of them)


@pytest.fixture(scope="module", autouse=True)
def _census(op_census):
    """Kernel-variant census for the whole module (conftest.op_census): tests/test_gpu_zz_coverage.py asserts that every variant the
    headline geometry selects was launched by one of the oracle comparisons of the operator-level modules."""
    yield


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


CASES = [
    # B, H, W, Cin, Cout, kh, kw, pad, ld_in, bn, stats, bias, stride
    (2, 8, 10, 48, 48, 1, 1, 0, None, False, False, False, (1, 1)),
    (2, 8, 10, 96, 384, 1, 1, 0, 144, True, False, False, (1, 1)),       # channel prefix of a wider buffer
    (2, 8, 10, 384, 48, 3, 3, 1, None, True, False, False, (1, 1)),      # decoder 3x3 (split-K)
    (4, 29, 38, 192, 1392, 1, 1, 0, 240, True, True, False, (1, 1)),     # e3 bottleneck + stats epilogue
    (4, 29, 38, 1392, 48, 3, 3, 1, None, True, True, False, (1, 1)),
    (2, 57, 57, 96, 2736, 1, 1, 0, None, True, True, False, (1, 1)),     # e2 bottleneck
    (1, 15, 19, 720, 48, 3, 3, 1, None, True, False, False, (1, 1)),     # ragged M
    (3, 7, 5, 2208, 180, 1, 1, 0, None, False, False, True, (1, 1)),     # conv2: N=180 + bias
    (2, 16, 16, 32, 32, 5, 5, 2, None, False, False, True, (1, 1)),      # WSM 5x5
    (2, 18, 16, 32, 32, 3, 16, 0, None, False, False, True, (1, 16)),    # WSM strip conv (3,k)/(1,k) after ZeroPad2d((0,0,1,1))
    (2, 16, 18, 32, 32, 16, 3, 0, None, False, False, True, (16, 1)),    # WSM strip conv (k,3)/(k,1)
    (1, 1, 1, 16, 16, 1, 1, 0, None, False, False, False, (1, 1)),       # single pixel
]


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_conv_family(case):
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cin, Cout, kh, kw, pad, ld_in, bn, stats, bias, stride = case
    ld_in = ld_in or Cin
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, ld_in, H, W, generator=gen)
    w = torch.randn(Cout, Cin, kh, kw, generator=gen) / (Cin * kh * kw) ** 0.5
    sc = torch.rand(Cin, generator=gen) + 0.5
    sh = torch.randn(Cin, generator=gen) * 0.3
    bs = torch.randn(Cout, generator=gen) if bias else None
    xin = x[:, :Cin]
    a = (F.relu(xin * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if bn else xin).detach().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(a, wr, bs, stride=stride, padding=pad)
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)
    Ho, Wo = ref.shape[2], ref.shape[3]

    xg, wg = nhwc(x).to(dev), w.to(dev)
    wp = torch.empty(kh * kw, Cout, Cin, device=dev)
    check(L.rdm_pack_conv_weight(ptr(wg), ptr(wp), Cout, Cin, kh, kw, Cout, stream()))
    y = torch.full((B, Ho, Wo, Cout), float("nan"), device=dev)
    d = ConvDesc(B, H, W, Cin, ld_in, Cout, Cout, kh, kw, stride[0], stride[1], pad, pad)
    scg, shg = sc.to(dev), sh.to(dev)
    ssum = torch.zeros(Cout, dtype=torch.float64, device=dev)
    ssq = torch.zeros_like(ssum)
    check(L.rdm_conv2d_fwd(C.byref(d), ptr(xg), ptr(wp), ptr(bs.to(dev)) if bias else None, ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(y),
                           ptr(ssum) if stats else None, ptr(ssq) if stats else None, stream()))
    got = y.cpu().permute(0, 3, 1, 2)
    assert rel(got, ref.detach()) < TOL
    if stats:
        rs = ref.detach().double()
        assert rel(ssum.cpu(), rs.sum((0, 2, 3))) < 1e-6 and rel(ssq.cpu(), (rs ** 2).sum((0, 2, 3))) < 1e-6
    gyg = nhwc(gy).to(dev)
    dwp = torch.zeros(kh * kw, Cout, Cin, device=dev)
    check(L.rdm_conv2d_wgrad(C.byref(d), ptr(gyg), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dwp), stream()))
    dw = torch.empty(Cout, Cin, kh, kw, device=dev)
    check(L.rdm_unpack_conv_weight(ptr(dwp), ptr(dw), Cout, Cin, kh, kw, Cout, stream()))
    assert rel(dw.cpu(), wr.grad) < TOL
    if Cout % 16 == 0 and stride == (1, 1):
        dx = torch.full((B, H, W, Cin), float("nan"), device=dev)
        check(L.rdm_conv2d_dgrad(C.byref(d), ptr(gyg), ptr(wp), ptr(dx), Cin, None, 0, None, None, None, None, stream()))
        assert rel(dx.cpu().permute(0, 3, 1, 2), a.grad) < TOL
        if bn:
            s0 = torch.zeros(Cin, dtype=torch.float64, device=dev)
            s1 = torch.zeros_like(s0)
            dx2 = torch.full((B, H, W, Cin), float("nan"), device=dev)
            check(L.rdm_conv2d_dgrad(C.byref(d), ptr(gyg), ptr(wp), ptr(dx2), Cin, ptr(xg), ld_in, ptr(scg), ptr(shg), ptr(s0), ptr(s1), stream()))
            z = xin * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
            refm = a.grad * (z > 0)
            assert rel(dx2.cpu().permute(0, 3, 1, 2), refm) < TOL
            assert rel(s0.cpu(), refm.double().sum((0, 2, 3))) < 1e-5
            assert rel(s1.cpu(), (refm.double() * xin.double()).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("cin,cout", [(96, 2736), (144, 1392), (144, 200), (336, 208), (48, 96), (16, 520)])
def test_conv1x1_big_grid(cin, cout):
    """The big-grid 1x1 forward (>= 32 768 pixels, dense_e2's regime) vs an f64 matmul: K a multiple of 32 and with a 16-channel tail,
    ragged M and N tiles, the BN-ReLU prologue, the statistics epilogue - and NaN in the channels BEHIND the contracted prefix of the
    input buffer (nothing beyond K may ever reach an accumulator)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, ld = 11, 57, 57, 384                       # 35 739 pixels: not a multiple of 128
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B * H * W, ld, generator=g)
    x[:, cin:] = float("nan")
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    d = ConvDesc(B, H, W, cin, ld, cout, cout, 1, 1, 1, 1, 0, 0)
    xg, wg, scg, shg = x.to(dev), w.to(dev).view(1, cout, cin).contiguous(), sc.to(dev), sh.to(dev)
    for bn in (True, False):
        a = (torch.relu(x[:, :cin] * sc + sh) if bn else x[:, :cin]).double()
        want = a @ w.double().t()
        for stats in (True, False):
            y = torch.full((B * H * W, cout), float("nan"), device=dev)
            ssum = torch.zeros(cout, dtype=torch.float64, device=dev)
            ssq = torch.zeros_like(ssum)
            check(L.rdm_conv2d_fwd(C.byref(d), ptr(xg), ptr(wg), None, ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(y),
                                   ptr(ssum) if stats else None, ptr(ssq) if stats else None, stream()))
            assert rel(y.cpu().double(), want) < TOL
            if stats:
                assert rel(ssum.cpu(), want.sum(0)) < 1e-5 and rel(ssq.cpu(), (want ** 2).sum(0)) < 1e-5
        # weight gradient of the same operator (long contraction over the 35 739 pixels; ragged pixel tail, ragged channel tiles)
        gy = torch.randn(B * H * W, cout, generator=g)
        dw = torch.zeros(1, cout, cin, device=dev)
        check(L.rdm_conv2d_wgrad(C.byref(d), ptr(gy.to(dev)), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), stream()))
        assert rel(dw[0].cpu().double(), gy.double().t() @ a) < TOL


# ---------------------------------------------------------------------------------------------
# Headline-size variants (B=16, 228x304: RDM_Net.py:526-530 shapes).  The dispatch in csrc/igemm.hip picks other kernels above certain
# sizes - the 256-pixel conv3x3_halo_kernel tiles (> 8 192 pixels), conv1x1_dma256_kernel
# (16 384 <= pixels < 32 768, >= 512 outputs) - and every one of them is compared here with a float64 torch-CPU evaluation of the same
# operator (shifted-slice matmuls: the definition of the convolution, no library conv), through the C ABI, at 2e-5 of the tensor's max.
# ---------------------------------------------------------------------------------------------
def _ref3x3(a, w9):
    """a (B,H,W,C) f64 (already normalised), w9 (9,N,C) f64 -> y (B*H*W, N): zero padding 1, stride 1."""
    B, H, W, Cc = a.shape
    ap = F.pad(a, (0, 0, 1, 1, 1, 1))
    y = torch.zeros(B * H * W, w9.shape[1], dtype=torch.float64)
    for r in range(3):
        for q in range(3):
            y += ap[:, r:r + H, q:q + W, :].reshape(-1, Cc) @ w9[r * 3 + q].t()
    return y


def _ref3x3_dgrad(gy, w9):
    """gy (B,H,W,N) f64, w9 (9,N,C) -> dx (B*H*W, C)"""
    B, H, W, N = gy.shape
    gp = F.pad(gy, (0, 0, 1, 1, 1, 1))
    dx = torch.zeros(B * H * W, w9.shape[2], dtype=torch.float64)
    for r in range(3):
        for q in range(3):
            dx += gp[:, 2 - r:2 - r + H, 2 - q:2 - q + W, :].reshape(-1, N) @ w9[r * 3 + q]
    return dx


def _ref3x3_wgrad(gy, a):
    B, H, W, N = gy.shape
    Cc = a.shape[3]
    ap = F.pad(a, (0, 0, 1, 1, 1, 1))
    g2 = gy.reshape(-1, N).t().contiguous()
    return torch.stack([g2 @ ap[:, r:r + H, q:q + W, :].reshape(-1, Cc) for r in range(3) for q in range(3)])


ROW_WGRAD_CASES = [
    # B, H, W, C, ld, N, bn      (>= 16 369 pixels: the shapes the round-1..3 row kernel served; since round 4 the generic tap kernel, conv_wgrad_kernel<.., TAPS>)
    (4, 57, 76, 2736, 2736, 48, True),       # dense_e2 conv2 (RDM_Net.py:526): C = 10 x 256 + 176 (ragged channel tile)
    (16, 29, 38, 1392, 1392, 48, True),      # dense_e3 conv2 at the bench batch (RDM_Net.py:528): C = 5 x 256 + 112
    (3, 75, 73, 208, 272, 40, False),        # 16 425 pixels (not a multiple of 16: ragged last slab), C < 256, ld > C, N < 48
    (5, 58, 57, 64, 64, 96, True),           # two 48-row tiles of output channels
]


@pytest.mark.parametrize("case", ROW_WGRAD_CASES, ids=[f"row{i}" for i in range(len(ROW_WGRAD_CASES))])
def test_wgrad3_many_pixels_vs_float64(case):
    """The exact-f32 3x3 weight gradient of the operator API / RDM_NET_OPT_DIRECT_3X3 at the many-pixel shapes (generic tap kernel) with its
    BN-ReLU prologue, ragged channel / row tiles, ragged pixel tail, the launcher's own split (f32 atomics) and split_k = 1 (every element
    owned by one workgroup)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cc, ld, N, bn = case
    g = torch.Generator().manual_seed(1000 + Cc)
    x = torch.randn(B, H, W, ld, generator=g)
    x[..., Cc:] = float("nan")                                   # nothing behind the contracted prefix may be read into a product
    gy = torch.randn(B, H, W, N, generator=g)
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.3
    a = (torch.relu(x[..., :Cc] * sc + sh) if bn else x[..., :Cc]).double()
    want = _ref3x3_wgrad(gy.double(), a)
    d = ConvDesc(B, H, W, Cc, ld, N, N, 3, 3, 1, 1, 1, 1)
    xg, gyg, scg, shg = x.to(dev), gy.to(dev), sc.to(dev), sh.to(dev)
    taps_launches = lambda: sum(v for k, v in _lib.census().items() if k.startswith("conv_wgrad_kernel/taps/"))
    before = taps_launches()
    for split in (0, 1, 7):
        dw = torch.zeros(9, N, Cc, device=dev)
        check(L.rdm_conv2d_wgrad_ex(C.byref(d), ptr(gyg), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), split, stream()))
        assert rel(dw.cpu().double(), want) < TOL, split
    assert taps_launches() == before + 3
    _RAN.add(("row", case))


HALO_CASES = [
    # B, H, W, Cb (3x3 input channels), hl     (> 8 192 pixels -> 256-pixel halo tiles; hl = ceil((256 + 2(W+1)) / 64))
    (9, 30, 31, 208, 5),
    (8, 29, 38, 192, 6),                     # dense_e3's map (RDM_Net.py:528)
    (2, 57, 76, 192, 7),                     # dense_e2's map (RDM_Net.py:526)
    (1, 66, 127, 144, 8),
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[f"hl{c[4]}" for c in HALO_CASES])
def test_halo3x3_256px_forward_and_dgrad_vs_float64(case):
    """conv3x3_halo_kernel<.., HL = 5..8, .., MT = 4> (256-pixel tiles), forward and dgrad, every epilogue: STORE, STORE_STATS (train-mode
    statistics), ATOMIC (K-split), MASK_STATS and MASK_STATS_ATOMIC (ReLU gate + BatchNorm-backward sums); ragged last pixel tile,
    image borders and batch wrap-around inside a tile, ragged channel tile of the dgrad (Cb not a multiple of 48)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cb, hl = case
    N, M = 48, B * H * W
    assert M > 8192
    g = torch.Generator().manual_seed(2000 + W)
    ld = Cb + 16
    y = torch.randn(B, H, W, ld, generator=g)
    y[..., Cb:] = float("nan")
    w = torch.randn(9, N, Cb, generator=g) / (9 * Cb) ** 0.5
    sc = torch.rand(Cb, generator=g) + 0.5
    sh = torch.randn(Cb, generator=g) * 0.3
    z = y[..., :Cb] * sc + sh
    a = torch.relu(z).double()
    want = _ref3x3(a, w.double())
    d = ConvDesc(B, H, W, Cb, ld, N, N, 3, 3, 1, 1, 1, 1)
    yg, wg, scg, shg = y.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    for split, stats in ((1, False), (0, True), (0, False), (4, False)):        # STORE | STORE_STATS | launcher's split | ATOMIC
        out = torch.full((M, N), float("nan"), device=dev)
        ssum = torch.zeros(N, dtype=torch.float64, device=dev)
        ssq = torch.zeros_like(ssum)
        check(L.rdm_conv2d_fwd_ex(C.byref(d), ptr(yg), ptr(wg), None, ptr(scg), ptr(shg), ptr(out), ptr(ssum) if stats else None,
                                  ptr(ssq) if stats else None, split, stream()))
        assert rel(out.cpu().double(), want) < TOL, (split, stats)
        if stats:
            assert rel(ssum.cpu(), want.sum(0)) < 1e-5 and rel(ssq.cpu(), (want ** 2).sum(0)) < 1e-5
    # dgrad of the same conv: dY[m][c] = sum_taps dOut[..][n] w[tap][n][c], gated by relu'(z), with sum dY and sum dY*y per channel
    go = torch.randn(B, H, W, N, generator=g)
    gog = go.to(dev)
    dx_ref = _ref3x3_dgrad(go.double(), w.double())
    gate = (z > 0).reshape(M, Cb).double()
    y2 = y[..., :Cb].reshape(M, Cb).double()
    for split, mask in ((1, False), (1, True), (3, False), (3, True)):          # STORE | MASK_STATS | ATOMIC | MASK_STATS_ATOMIC
        dx = torch.full((M, Cb), float("nan"), device=dev)
        s0 = torch.zeros(Cb, dtype=torch.float64, device=dev)
        s1 = torch.zeros_like(s0)
        check(L.rdm_conv2d_dgrad_ex(C.byref(d), ptr(gog), ptr(wg), ptr(dx), Cb, ptr(yg) if mask else None, ld, ptr(scg) if mask else None,
                                    ptr(shg) if mask else None, ptr(s0) if mask else None, ptr(s1) if mask else None, split, stream()))
        ref = dx_ref * gate if mask else dx_ref
        assert rel(dx.cpu().double(), ref) < TOL, (split, mask)
        if mask:
            assert rel(s0.cpu(), ref.sum(0)) < 1e-5 and rel(s1.cpu(), (ref * y2).sum(0)) < 1e-5
    cen = _lib.census()
    for name in ("fwd/px256/hl%d/STORE", "fwd/px256/hl%d/STORE_STATS", "fwd/px256/hl%d/ATOMIC", "dgrad/px256/hl%d/STORE", "dgrad/px256/hl%d/MASK_STATS",
                 "dgrad/px256/hl%d/ATOMIC", "dgrad/px256/hl%d/MASK_STATS_ATOMIC"):
        assert cen.get("conv3x3_halo_kernel/" + name % hl, 0) >= 1, (name % hl, sorted(cen))
    _RAN.add(("halo", case))


@pytest.mark.parametrize("cin,cout", [(192, 1392), (240, 1392), (720, 1392), (912, 1392), (192, 528), (240, 528), (720, 528), (912, 528), (240, 2736)])
def test_conv1x1_dma256_kernel_vs_float64(cin, cout):
    """conv1x1_dma256_kernel (16 384 <= pixels < 32 768, >= 512 outputs: dense_e3's bottleneck, RDM_Net.py:528) vs an f64 matmul:
    K a multiple of 32 (192) and with a 16-channel tail (240, 720, 912), ragged last pixel tile (17 632 = 68 x 256 +
    224) and channel tile (1392 = 10 x 128 + 112, 528 = 4 x 128 + 16), BN-ReLU coefficients by DMA on / off, STORE and STORE_STATS, and NaN
    in the channels BEHIND the contracted prefix of the input buffer.  The launcher takes 192 x 128 tiles where they fill the rounds of 512
    resident workgroups better (1392 / 528 outputs here: 17 632 = 91 x 192 + 160) and 256 x 128 otherwise (2736 outputs)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W = 16, 29, 38
    M, ld = B * H * W, cin + 48
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.randn(M, ld, generator=g)
    x[:, cin:] = float("nan")
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    d = ConvDesc(B, H, W, cin, ld, cout, cout, 1, 1, 1, 1, 0, 0)
    xg, wg, scg, shg = x.to(dev), w.to(dev).view(1, cout, cin).contiguous(), sc.to(dev), sh.to(dev)
    for bn in (True, False):
        a = (torch.relu(x[:, :cin] * sc + sh) if bn else x[:, :cin]).double()
        want = a @ w.double().t()
        for stats in (True, False):
            nt, cd = (cout + 127) // 128, lambda a, b: (a + b - 1) // b
            m192 = cd(nt * cd(M, 192), 512) * 192 < cd(nt * cd(M, 256), 512) * 256          # the launcher's rule
            assert m192 == (cout != 2736)
            key = "conv1x1_dma256_kernel/%s/bn%d%s" % ("STORE_STATS" if stats else "STORE", int(bn), "/m192" if m192 else "")
            before = _lib.census().get(key, 0)
            y = torch.full((M, cout), float("nan"), device=dev)
            ssum = torch.zeros(cout, dtype=torch.float64, device=dev)
            ssq = torch.zeros_like(ssum)
            # split_k = 1: what the statistics epilogue runs with anyway (the plan's 12 launches per step); without it the launcher
            # may split the short-N case (528 outputs) over K, which is conv_fwd_kernel's business
            check(L.rdm_conv2d_fwd_ex(C.byref(d), ptr(xg), ptr(wg), None, ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(y),
                                      ptr(ssum) if stats else None, ptr(ssq) if stats else None, 1, stream()))
            assert _lib.census().get(key, 0) == before + 1, key                 # the kernel under test is the one that ran
            assert rel(y.cpu().double(), want) < TOL, (bn, stats)
            if stats:
                assert rel(ssum.cpu(), want.sum(0)) < 1e-5 and rel(ssq.cpu(), (want ** 2).sum(0)) < 1e-5
    _RAN.add(("dma256", cin, cout))


@pytest.mark.parametrize("cin", [96, 288])
def test_conv1x1_big_grid_dgrad_and_small_grid_split(cin):
    """The 1x1 dgrad of dense_e2's bottleneck (RDM_Net.py:526; > 32 768 pixels, N = cin a multiple of 96 -> 128 x 96 tiles) with the
    ReLU gate + BatchNorm-backward sums, unsplit (MASK_STATS) and K-split (MASK_STATS_ATOMIC, ATOMIC); and the K-split 1x1 forward on
    128 x 48 tiles (dense_e4 / decoder: conv1 'part A' accumulates atomically, net.hip forward_block) - all vs float64 matmuls."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, cb = (16, 57, 76, 2736) if cin == 96 else (11, 57, 57, 2736)    # N = 96 takes the 128 x 96 tiles only at the headline's 69 312 pixels (>= 512 tiles)
    M, ld = B * H * W, cin + 96
    g = torch.Generator().manual_seed(31 + cin)
    dz = torch.randn(M, cb, generator=g)
    w = torch.randn(cb, cin, generator=g) / cb ** 0.5
    x = torch.randn(M, ld, generator=g)
    sc = torch.rand(cin, generator=g) + 0.5
    sh = torch.randn(cin, generator=g) * 0.3
    want = dz.double() @ w.double()
    gate = ((x[:, :cin] * sc + sh) > 0).double()
    d = ConvDesc(B, H, W, cin, ld, cb, cb, 1, 1, 1, 1, 0, 0)
    dzg, wg, xg, scg, shg = dz.to(dev), w.to(dev).view(1, cb, cin).contiguous(), x.to(dev), sc.to(dev), sh.to(dev)
    for split, mask in ((1, True), (3, True), (1, False), (3, False)):
        dx = torch.full((M, cin), float("nan"), device=dev)
        s0 = torch.zeros(cin, dtype=torch.float64, device=dev)
        s1 = torch.zeros_like(s0)
        check(L.rdm_conv2d_dgrad_ex(C.byref(d), ptr(dzg), ptr(wg), ptr(dx), cin, ptr(xg) if mask else None, ld, ptr(scg) if mask else None,
                                    ptr(shg) if mask else None, ptr(s0) if mask else None, ptr(s1) if mask else None, split, stream()))
        ref = want * gate if mask else want
        assert rel(dx.cpu().double(), ref) < TOL, (split, mask)
        if mask:
            assert rel(s0.cpu(), ref.sum(0)) < 1e-5 and rel(s1.cpu(), (ref * x[:, :cin].double()).sum(0)) < 1e-5
    cen = _lib.census()
    for e in ("MASK_STATS", "MASK_STATS_ATOMIC", "STORE", "ATOMIC"):
        assert cen.get("conv_fwd_kernel/dgrad/1x1/tile128x96/" + e, 0) >= 1, (e, sorted(cen))
    # K-split forward at dense_e4's size: 4 560 pixels, K = 1056 + cin, 720 outputs
    Bs, Hs, Ws, K, N = 16, 15, 19, 1056 + cin, 720
    Ms = Bs * Hs * Ws
    xs = torch.randn(Ms, K, generator=g)
    ws = torch.randn(N, K, generator=g) / K ** 0.5
    scs = torch.rand(K, generator=g) + 0.5
    shs = torch.randn(K, generator=g) * 0.3
    wants = torch.relu(xs * scs + shs).double() @ ws.double().t()
    ds = ConvDesc(Bs, Hs, Ws, K, K, N, N, 1, 1, 1, 1, 0, 0)
    xsg, wsg, scsg, shsg = xs.to(dev), ws.to(dev).view(1, N, K).contiguous(), scs.to(dev), shs.to(dev)
    for split in (1, 4, 0):
        y = torch.full((Ms, N), float("nan"), device=dev)
        check(L.rdm_conv2d_fwd_ex(C.byref(ds), ptr(xsg), ptr(wsg), None, ptr(scsg), ptr(shsg), ptr(y), None, None, split, stream()))
        assert rel(y.cpu().double(), wants) < TOL, split
    cen = _lib.census()
    assert cen.get("conv_fwd_kernel/fwd/1x1/tile128x48/ATOMIC", 0) >= 1 and cen.get("conv_fwd_kernel/fwd/1x1/tile128x48/STORE", 0) >= 1
    _RAN.add(("dgrad_big", cin))


@pytest.mark.parametrize("case", [(16, 15, 19, 48, 2112, 720, 1, 1), (16, 15, 19, 720, 720, 48, 3, 3), (16, 8, 10, 384, 384, 48, 3, 3), (16, 8, 10, 48, 2208, 384, 1, 1)],
                         ids=["e4_1x1", "e4_3x3", "d1_3x3", "d1_1x1"])
def test_conv_fwd_with_raw_batchnorm_sums(case):
    """rdm_conv2d_fwd_bnsums: the BatchNorm + ReLU prologue formed INSIDE the conv from the channel sums (what the plan's few-pixel blocks
    run in training, so that no k_bn_finalize launch sits on the dependent chain) vs float64, and bit-identical with rdm_bn_finalize ->
    rdm_conv2d_fwd on the same sums (same arithmetic)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cin, ld, Cout, kh, kw = case
    M = B * H * W
    g = torch.Generator().manual_seed(77 + Cin + kh)
    x = torch.randn(B, H, W, ld, generator=g) * 1.5 + 0.4
    w = torch.randn(kh * kw, Cout, Cin, generator=g) / (Cin * kh * kw) ** 0.5
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    xs = x[..., :Cin].double().reshape(M, Cin)
    mean, var = xs.mean(0), xs.var(0, unbiased=False)
    a = torch.relu((xs - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()).reshape(B, H, W, Cin)
    want = _ref3x3(a, w.double()) if kh == 3 else a.reshape(M, Cin) @ w[0].double().t()
    xg, wg, gg, bg = x.to(dev), w.to(dev), gamma.to(dev), beta.to(dev)
    ssum = torch.zeros(Cin, dtype=torch.float64, device=dev)
    ssq = torch.zeros_like(ssum)
    check(L.rdm_bn_stats(ptr(xg), ld, M, Cin, ptr(ssum), ptr(ssq), stream()))
    d = ConvDesc(B, H, W, Cin, ld, Cout, Cout, kh, kw, 1, 1, kh // 2, kw // 2)
    pad = kh // 2
    outs = []
    for split, stats in ((1, kh == 1), (1, False), (3, False)):       # the statistics epilogue exists for the 1x1 (conv1 "part B") only
        y = torch.full((M, Cout), float("nan"), device=dev)
        s0 = torch.zeros(Cout, dtype=torch.float64, device=dev)
        s1 = torch.zeros_like(s0)
        check(L.rdm_conv2d_fwd_bnsums(C.byref(d), ptr(xg), ptr(wg), ptr(ssum), ptr(ssq), float(M), ptr(gg), ptr(bg), ptr(y), ptr(s0) if stats else None,
                                      ptr(s1) if stats else None, split, stream()))
        assert rel(y.cpu().double(), want) < TOL, (split, stats)
        if stats:
            assert rel(s0.cpu(), want.sum(0)) < 1e-5 and rel(s1.cpu(), (want ** 2).sum(0)) < 1e-5
        outs.append(y)
    # the same result, bit for bit, through the finalisation kernel (unsplit)
    coef = torch.empty(4, Cin, device=dev)
    rm, rv, nbt = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    check(L.rdm_bn_finalize(ptr(ssum), ptr(ssq), float(M), ptr(gg), ptr(bg), ptr(rm), ptr(rv), ptr(nbt), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), Cin, 1, stream()))
    y2 = torch.empty(M, Cout, device=dev)
    check(L.rdm_conv2d_fwd_ex(C.byref(d), ptr(xg), ptr(wg), None, ptr(coef[0]), ptr(coef[1]), ptr(y2), None, None, 1, stream()))
    assert torch.equal(outs[1], y2)
    if kh == 3:
        # the plan's form (round 5): added into a zeroed slice of a wider buffer, the statistics of the finished tile taken by the last K split to arrive
        ldo = 96
        for split in (0, 1, 5):
            yb = torch.zeros(M, ldo, device=dev)
            s0 = torch.zeros(Cout, dtype=torch.float64, device=dev)
            s1 = torch.zeros_like(s0)
            tickets = torch.zeros((M + 127) // 128, dtype=torch.int32, device=dev)
            da = ConvDesc(B, H, W, Cin, ld, Cout, ldo, kh, kw, 1, 1, 1, 1)
            for rep in range(2):                                          # twice into the same tickets: they are left zero
                yb.zero_(); s0.zero_(); s1.zero_()
                check(L.rdm_conv3x3_fwd_bnsums_acc(C.byref(da), ptr(xg), ptr(wg), ptr(ssum), ptr(ssq), float(M), ptr(gg), ptr(bg), C.c_void_p(yb.data_ptr() + 48 * 4), ptr(s0), ptr(s1), ptr(tickets), split, stream()))
                torch.cuda.synchronize()
                assert int(tickets.abs().max()) == 0
                got = yb[:, 48:48 + Cout].cpu().double()
                assert rel(got, want) < TOL and float(yb[:, :48].abs().max()) == 0.0, (split, rep)
                # the statistics are those of the STORED values (every tile exactly once), to float32 summation accuracy
                assert rel(s0.cpu(), got.sum(0)) < 2e-6 * M ** 0.5 and rel(s1.cpu(), (got ** 2).sum(0)) < 1e-5, (split, rep)
    cen = _lib.census()
    assert any(k.endswith("/rawbn") for k in cen), sorted(cen)
    assert kh != 3 or any(k.endswith("/rawbn/stats") for k in cen), sorted(cen)
    _RAN.add(("rawbn",) + case)


def test_conv_linearity_full_size():
    """Size-independent property at the bench geometry (B=16, 57x76, 96->2736): conv(a*x1+b*x2) == a*conv(x1)+b*conv(x2)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cin, Cout = 16, 57, 76, 96, 2736
    g = torch.Generator(device="cpu").manual_seed(3)
    x1 = torch.randn(B, H, W, Cin, generator=g).to(dev)
    x2 = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(1, Cout, Cin, generator=g) / Cin ** 0.5).to(dev)
    d = ConvDesc(B, H, W, Cin, Cin, Cout, Cout, 1, 1, 1, 1, 0, 0)

    def run(x):
        y = torch.empty(B, H, W, Cout, device=dev)
        check(L.rdm_conv2d_fwd(C.byref(d), ptr(x), ptr(w), None, None, None, ptr(y), None, None, stream()))
        return y

    lhs = run(0.75 * x1 - 1.5 * x2)
    rhs = 0.75 * run(x1) - 1.5 * run(x2)
    assert rel(lhs, rhs) < 1e-5
    # spot-check 64 output pixels against an f64 matmul
    idx = torch.randint(0, B * H * W, (64,), generator=g).to(dev)
    want = x1.view(-1, Cin)[idx].double() @ w[0].double().t()
    assert rel(run(x1).view(-1, Cout)[idx].double(), want) < 1e-5


def test_fused_adamw_matches_torch():
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    n = 1_000_003
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(n + 1, generator=g)[:n].contiguous()
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-4)
    pg = torch.zeros(n + 5, device=dev)[:n]
    pg.copy_(p0)
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        p.grad = gr.clone()
        opt.step()
        check(L.rdm_adamw_fused(ptr(pg), ptr(gr.to(dev)), ptr(m), ptr(v), n, 1e-4, 0.9, 0.999, 1e-8, 0.01, step, 1.0, stream()))
    assert rel(pg.cpu(), p.detach()) < 1e-6
