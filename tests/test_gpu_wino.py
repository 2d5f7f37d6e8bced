"""-m gpu: Winograd F(2x2, 3x3) (csrc/wino.hip) through the C ABI vs a float64 torch-CPU evaluation of the same convolution (shifted-slice
matmuls: the definition, no library conv) - the SAME 2e-5-of-max tolerance the direct implicit-GEMM kernels are held to."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import conftest

pytestmark = pytest.mark.gpu
TOL = 2e-5
_RAN = conftest.RAN.setdefault(__name__, set())


@pytest.fixture(scope="module", autouse=True)
def _census(op_census):
    yield


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def ref3x3(a, w9):
    B, H, W, Cc = a.shape
    ap = F.pad(a, (0, 0, 1, 1, 1, 1))
    y = torch.zeros(B * H * W, w9.shape[1], dtype=torch.float64)
    for r in range(3):
        for q in range(3):
            y += ap[:, r:r + H, q:q + W, :].reshape(-1, Cc) @ w9[r * 3 + q].t()
    return y


CASES = [
    # B, H, W, Cb, ld, N, bn
    (2, 57, 76, 2736, 2736, 48, True),       # dense_e2's conv2 (RDM_Net.py:526): odd height (a half-valid last tile row)
    (16, 29, 38, 1392, 1392, 48, True),      # dense_e3's conv2 at the bench batch (RDM_Net.py:528)
    (3, 8, 10, 384, 400, 48, True),          # decoder-sized map, ld > C with NaN behind the contracted prefix
    (1, 7, 5, 48, 48, 48, False),            # odd width and height, fewer tiles than one workgroup, no prologue
    (5, 13, 21, 96, 96, 40, True),           # N < 48
    (1, 1, 1, 16, 16, 16, True),             # single pixel
]


@pytest.mark.parametrize("kernel", ["f32", "x6"])
@pytest.mark.parametrize("case", CASES, ids=[f"w{i}" for i in range(len(CASES))])
def test_wino_forward_vs_float64(case, kernel):
    """kernel: f32 = conv3x3_wino_fwd_kernel (f32 MFMA); x6 = conv3x3_wino_x6_kernel (round 5: both transformed operands split three ways, six
    bf16 MFMAs per float32 product - float32-equivalent).  Same cases, same 2e-5; the x6 kernel is additionally held to 5e-6 (measured 2.1e-6 at K = 24 624;
    the f32 Winograd kernel 6e-7: the splits of both transformed operands round at 2^-24 each)."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cb, ld, N, bn = case
    M = B * H * W
    g = torch.Generator().manual_seed(4000 + Cb + W)
    y = torch.randn(B, H, W, ld, generator=g)
    y[..., Cb:] = float("nan")
    w = torch.randn(9, N, Cb, generator=g) / (9 * Cb) ** 0.5
    sc = torch.rand(Cb, generator=g) + 0.5
    sh = torch.randn(Cb, generator=g) * 0.3
    a = (torch.relu(y[..., :Cb] * sc + sh) if bn else y[..., :Cb]).double()
    want = ref3x3(a, w.double())
    d = ConvDesc(B, H, W, Cb, ld, N, 64, 3, 3, 1, 1, 1, 1)               # output into a 64-wide buffer: a channel slice, as in the block buffers
    yg, wg, scg, shg = y.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    outs = []
    wsq = L.rdm_conv3x3_wino_x6_workspace_bytes if kernel == "x6" else L.rdm_conv3x3_wino_workspace_bytes
    fwd = L.rdm_conv3x3_wino_fwd_x6 if kernel == "x6" else L.rdm_conv3x3_wino_fwd
    tol = 5e-6 if kernel == "x6" else TOL
    for split, stats in ((1, False), (0, True), (3, True), (0, False)):
        nb = int(wsq(Cb, B, H, W, split))
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
        out = torch.full((M, 64), float("nan"), device=dev)
        ssum = torch.zeros(N, dtype=torch.float64, device=dev)
        ssq = torch.zeros_like(ssum)
        check(fwd(C.byref(d), ptr(yg), ptr(wg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(out),
                  ptr(ssum) if stats else None, ptr(ssq) if stats else None, ptr(ws), nb, split, stream()))
        got = out.cpu()
        assert torch.isnan(got[:, N:]).all()                               # nothing outside the N-channel slice is written
        assert rel(got[:, :N].double(), want) < tol, (split, stats, rel(got[:, :N].double(), want))
        if stats:
            assert rel(ssum.cpu(), want.sum(0)) < 1e-5 and rel(ssq.cpu(), (want ** 2).sum(0)) < 1e-5
        outs.append((split, got[:, :N].clone()))
    # no atomics anywhere: the same split is bit-reproducible
    nb = int(wsq(Cb, B, H, W, 3))
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device=dev)
    out = torch.full((M, 64), float("nan"), device=dev)
    check(fwd(C.byref(d), ptr(yg), ptr(wg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(out), None, None, ptr(ws), nb, 3, stream()))
    assert torch.equal(out.cpu()[:, :N], outs[2][1])
    _RAN.add((kernel, case))



def ref3x3_wgrad(gy, a):
    B, H, W, N = gy.shape
    Cc = a.shape[3]
    ap = F.pad(a, (0, 0, 1, 1, 1, 1))
    g2 = gy.reshape(-1, N).t().contiguous()
    return torch.stack([g2 @ ap[:, r:r + H, q:q + W, :].reshape(-1, Cc) for r in range(3) for q in range(3)])


WGRAD_CASES = [
    # B, H, W, Cb, ld, N, ldg, bn
    (4, 57, 76, 2736, 2736, 48, 384, True),      # dense_e2's conv2 (RDM_Net.py:526), gradient = a 48-channel slice of the 384-wide block gradient
    (16, 29, 38, 1392, 1392, 48, 768, True),     # dense_e3's conv2 at the bench batch (RDM_Net.py:528)
    (3, 8, 10, 384, 400, 48, 48, True),          # decoder-sized map, NaN behind the contracted prefix
    (1, 7, 5, 100, 100, 40, 40, False),          # odd sizes, ragged last 64-channel block, N < 48, no prologue
    (2, 9, 9, 64, 64, 48, 48, True),             # tile rows shorter than a 16-tile slab (several cursor wraps per slab)
    (1, 1, 1, 16, 16, 16, 16, True),             # single pixel
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[f"g{i}" for i in range(len(WGRAD_CASES))])
def test_wino_wgrad_vs_float64(case):
    """conv3x3_wino_wgrad_kernel (Winograd F(3x3, 2x2), the contraction runs over the tiles) vs a float64 evaluation; written, not
    accumulated; bit-reproducible."""
    from md_rdm_amd import _lib
    from md_rdm_amd._lib import ConvDesc, check, ptr, stream
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, H, W, Cb, ld, N, ldg, bn = case
    g = torch.Generator().manual_seed(6000 + Cb + W)
    x = torch.randn(B, H, W, ld, generator=g)
    if ld > Cb:
        x[..., Cb:] = float("nan")
    go = torch.randn(B, H, W, ldg, generator=g)
    if ldg > N:
        go[..., N:] = float("nan")
    sc = torch.rand(Cb, generator=g) + 0.5
    sh = torch.randn(Cb, generator=g) * 0.3
    a = (torch.relu(x[..., :Cb] * sc + sh) if bn else x[..., :Cb]).double()
    want = ref3x3_wgrad(go[..., :N].double(), a)
    d = ConvDesc(B, H, W, Cb, ld, N, ldg, 3, 3, 1, 1, 1, 1)
    xg, gog, scg, shg = x.to(dev), go.to(dev), sc.to(dev), sh.to(dev)
    nb = int(L.rdm_conv3x3_wino_wgrad_workspace_bytes(Cb, B, H, W))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        dw = torch.full((9, N, Cb), float("nan"), device=dev)
        check(L.rdm_conv3x3_wino_wgrad(C.byref(d), ptr(gog), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), ptr(ws), nb, stream()))
        outs.append(dw.cpu())
    assert rel(outs[0].double(), want) < TOL
    assert torch.equal(outs[0], outs[1])                                   # no atomics: bit-reproducible
    _RAN.add(("wgrad",) + case)
