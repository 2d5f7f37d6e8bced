"""-m gpu: the fp32 MFMA implicit-GEMM family through the C ABI vs PyTorch-CPU conv2d + autograd
(fp32 reference of the same op).  Tolerance 2e-5 relative to the tensor's max (f32 accumulation order)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5


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
