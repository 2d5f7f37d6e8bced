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
        check(L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(gyg), ptr(xg), ptr(scg) if bn else None, ptr(shg) if bn else None, ptr(dw), split, stream()))
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
    assert L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(t), ptr(t), None, None, ptr(t), 0, stream()) < 0
