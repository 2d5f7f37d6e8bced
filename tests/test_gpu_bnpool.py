"""-m gpu: the BatchNorm / pooling operators of the conv stack through the C ABI (rdm_bn_*, rdm_maxpool3s2_*, rdm_padavgpool2_*)
vs torch-CPU autograd of the modules the reference composes (torchvision _DenseLayer / _Transition BatchNorm2d + ReLU,
nn.MaxPool2d(3, 2, 1), ZeroPad2d((0,1,0,1)) + AvgPool2d(2); network/RDM_Net.py:525,527,532).  These kernels otherwise only run
inside rdm_net_backward.  Tolerance 2e-5 of the tensor's max (f32 kernels with f64 reductions vs torch f32), ragged sizes, the
zero-padded transition row / column, train and eval mode."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from md_rdm_amd import filler

pytestmark = pytest.mark.gpu
U = filler.uniform
TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def L():
    from md_rdm_amd import _lib
    return _lib


def nhwc(t_nchw, ld, dev):
    """(B,C,H,W) cpu -> NHWC device buffer with pixel stride ld (extra columns poisoned with NaN: must never be read into results)."""
    B, Cc, H, W = t_nchw.shape
    buf = torch.full((B, H, W, ld), float("nan"), dtype=torch.float32)
    buf[..., :Cc] = t_nchw.permute(0, 2, 3, 1)
    return buf.to(dev)


def close(got, want, what):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    err = np.abs(got - want).max()
    assert err <= TOL * max(np.abs(want).max(), 1e-30), (what, err, np.abs(want).max())


@pytest.mark.parametrize("training", [1, 0])
@pytest.mark.parametrize("B,H,W,Cc,ld", [(2, 15, 19, 48, 64), (3, 8, 10, 100, 100), (1, 57, 76, 336, 384)])
def test_bn_relu_forward_backward(dev, B, H, W, Cc, ld, training):
    lib, st = L().lib(), L().stream()
    x = torch.from_numpy(U(f"bn.x{B}{H}{Cc}", (B, Cc, H, W), -2.0, 2.0))
    gz = torch.from_numpy(U(f"bn.g{B}{H}{Cc}", (B, Cc, H, W), -1.0, 1.0))
    gamma = torch.from_numpy(U("bn.gamma", (Cc,), 0.5, 1.5))
    beta = torch.from_numpy(U("bn.beta", (Cc,), -0.3, 0.3))
    rm0 = torch.from_numpy(U("bn.rm", (Cc,), -0.2, 0.2))
    rv0 = torch.from_numpy(U("bn.rv", (Cc,), 0.5, 1.5))
    # ---- torch CPU reference: BatchNorm2d -> ReLU, loss = sum(z * gz)
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
    bn.train(bool(training))
    xr = x.clone().requires_grad_(True)
    z = F.relu(bn(xr))
    (z * gz).sum().backward()
    # ---- HIP path
    M = B * H * W
    xd = nhwc(x, ld, dev)
    gd = {k: v.to(dev) for k, v in dict(gamma=gamma, beta=beta, rm=rm0.clone(), rv=rv0.clone()).items()}
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    s = torch.zeros(2, Cc, dtype=torch.float64, device=dev)
    sc = torch.empty(4, Cc, dtype=torch.float32, device=dev)                          # scale | shift | mean | rstd
    P = L().ptr
    row = lambda t, i: C.c_void_p(t.data_ptr() + i * t.stride(0) * t.element_size())
    L().check(lib.rdm_bn_stats(P(xd), ld, M, Cc, row(s, 0), row(s, 1), st))
    L().check(lib.rdm_bn_finalize(row(s, 0), row(s, 1), float(M), P(gd["gamma"]), P(gd["beta"]), P(gd["rm"]), P(gd["rv"]), P(nbt),
                                  row(sc, 0), row(sc, 1), row(sc, 2), row(sc, 3), Cc, training, st))
    zh = torch.relu(xd[..., :Cc] * sc[0] + sc[1])                                      # what a conv prologue applies
    close(zh.permute(0, 3, 1, 2).cpu(), z.detach(), "bn+relu forward")
    if training:
        close(gd["rm"].cpu(), bn.running_mean, "running_mean")
        close(gd["rv"].cpu(), bn.running_var, "running_var")
        assert int(nbt) == 1
    else:
        assert torch.equal(gd["rm"].cpu(), rm0) and torch.equal(gd["rv"].cpu(), rv0) and int(nbt) == 0
    dz = nhwc(gz, ld, dev)
    r = torch.zeros(2, Cc, dtype=torch.float64, device=dev)
    L().check(lib.rdm_bn_bwd_reduce(P(dz), ld, P(xd), ld, row(sc, 0), row(sc, 1), M, Cc, row(r, 0), row(r, 1), st))
    dx = torch.full((B, H, W, ld), float("nan"), dtype=torch.float32, device=dev)
    dgam = torch.empty(Cc, dtype=torch.float32, device=dev)
    dbet = torch.empty(Cc, dtype=torch.float32, device=dev)
    L().check(lib.rdm_bn_bwd(P(dx), ld, P(dz), ld, P(xd), ld, row(r, 0), row(r, 1), float(M), P(gd["gamma"]), row(sc, 2), row(sc, 3), P(dgam), P(dbet),
                             M, Cc, 0, training, st))
    close(dx[..., :Cc].permute(0, 3, 1, 2).cpu(), xr.grad, "dx")
    close(dgam.cpu(), bn.weight.grad, "dgamma")
    close(dbet.cpu(), bn.bias.grad, "dbeta")
    # accumulate mode adds into an existing gradient
    base = torch.from_numpy(U("bn.base", (B, H, W, ld), -1, 1)).to(dev)
    acc = base.clone()
    L().check(lib.rdm_bn_bwd(P(acc), ld, P(dz), ld, P(xd), ld, row(r, 0), row(r, 1), float(M), P(gd["gamma"]), row(sc, 2), row(sc, 3), None, None,
                             M, Cc, 1, training, st))
    close((acc - base)[..., :Cc].permute(0, 3, 1, 2).cpu(), xr.grad, "dx (accumulate)")


@pytest.mark.parametrize("B,H,W,Cc,ld", [(2, 114, 152, 96, 384), (1, 7, 9, 8, 8), (3, 16, 16, 20, 32)])
def test_maxpool3s2_forward_backward(dev, B, H, W, Cc, ld):
    lib, st, P = L().lib(), L().stream(), L().ptr
    x = torch.from_numpy(U(f"mp.x{H}", (B, Cc, H, W), -2.0, 2.0))
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.from_numpy(U(f"mp.g{H}", tuple(y.shape), -1.0, 1.0))
    (y * gy).sum().backward()
    Ho, Wo = y.shape[2], y.shape[3]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    yd = torch.full((B, Ho, Wo, ld), float("nan"), dtype=torch.float32, device=dev)
    arg = torch.empty(B, Ho, Wo, Cc, dtype=torch.uint8, device=dev)
    L().check(lib.rdm_maxpool3s2_fwd(P(xd), P(yd), ld, P(arg), B, H, W, Cc, st))
    assert torch.equal(yd[..., :Cc].permute(0, 3, 1, 2).cpu(), y.detach())             # a max is exact
    assert int(arg.max()) <= 8
    gd = nhwc(gy, ld, dev)
    dx = torch.full((B, H, W, Cc), float("nan"), dtype=torch.float32, device=dev)
    L().check(lib.rdm_maxpool3s2_bwd(P(gd), ld, P(arg), P(dx), B, H, W, Cc, st))
    close(dx.permute(0, 3, 1, 2).cpu(), xr.grad, "maxpool dx")                          # <= 4 windows summed per pixel


@pytest.mark.parametrize("training", [1, 0])
@pytest.mark.parametrize("B,H,W,Cc,ld", [(2, 15, 19, 48, 64), (2, 8, 10, 64, 64), (1, 57, 76, 384, 384), (3, 29, 38, 100, 128)])
def test_padavgpool2_transition_front_end(dev, B, H, W, Cc, ld, training):
    """ZeroPad2d((0,1,0,1)) -> BatchNorm2d -> ReLU -> AvgPool2d(2): odd extents (pad completes the last window) and even
    extents (the padded row / column falls outside every window but still counts in the batch statistics)."""
    lib, st, P = L().lib(), L().stream(), L().ptr
    x = torch.from_numpy(U(f"tp.x{H}{Cc}", (B, Cc, H, W), -2.0, 2.0))
    gamma = torch.from_numpy(U("tp.gamma", (Cc,), 0.5, 1.5))
    beta = torch.from_numpy(U("tp.beta", (Cc,), -0.3, 0.3))
    bn = torch.nn.BatchNorm2d(Cc)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
        bn.running_mean.copy_(torch.from_numpy(U("tp.rm", (Cc,), -0.2, 0.2))); bn.running_var.copy_(torch.from_numpy(U("tp.rv", (Cc,), 0.5, 1.5)))
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train(bool(training))
    xr = x.clone().requires_grad_(True)
    p = F.avg_pool2d(F.relu(bn(F.pad(xr, (0, 1, 0, 1)))), 2)
    gp = torch.from_numpy(U(f"tp.g{H}{Cc}", tuple(p.shape), -1.0, 1.0))
    (p * gp).sum().backward()
    Ho, Wo = p.shape[2], p.shape[3]
    assert (Ho, Wo) == ((H + 1) // 2, (W + 1) // 2)
    xd = nhwc(x, ld, dev)
    M, count = B * H * W, float(B * (H + 1) * (W + 1))
    s = torch.zeros(2, Cc, dtype=torch.float64, device=dev)
    sc = torch.empty(4, Cc, dtype=torch.float32, device=dev)
    row = lambda t, i: C.c_void_p(t.data_ptr() + i * t.stride(0) * t.element_size())
    rm, rv, nbt = rm0.to(dev), rv0.to(dev), torch.zeros(1, dtype=torch.int64, device=dev)
    gam, bet = gamma.to(dev), beta.to(dev)
    L().check(lib.rdm_bn_stats(P(xd), ld, M, Cc, row(s, 0), row(s, 1), st))             # the zero padding adds nothing to the sums, only to the count
    L().check(lib.rdm_bn_finalize(row(s, 0), row(s, 1), count, P(gam), P(bet), P(rm), P(rv), P(nbt), row(sc, 0), row(sc, 1), row(sc, 2), row(sc, 3),
                                  Cc, training, st))
    pooled = torch.empty(B, Ho, Wo, Cc, dtype=torch.float32, device=dev)
    L().check(lib.rdm_padavgpool2_fwd(P(xd), ld, row(sc, 0), row(sc, 1), P(pooled), B, H, W, Cc, st))
    close(pooled.permute(0, 3, 1, 2).cpu(), p.detach(), "pooled")
    if training:
        close(rm.cpu(), bn.running_mean, "running_mean (padded count)")
        close(rv.cpu(), bn.running_var, "running_var (padded count)")
    wsb = int(lib.rdm_padavgpool2_bwd_workspace_bytes(Cc))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dx = torch.full((B, H, W, ld), float("nan"), dtype=torch.float32, device=dev)
    dgam = torch.empty(Cc, dtype=torch.float32, device=dev)
    dbet = torch.empty(Cc, dtype=torch.float32, device=dev)
    gpd = gp.permute(0, 2, 3, 1).contiguous().to(dev)
    L().check(lib.rdm_padavgpool2_bwd(P(gpd), P(xd), ld, row(sc, 0), row(sc, 1), P(gam), row(sc, 2), row(sc, 3), P(dx), ld, P(dgam), P(dbet),
                                      B, H, W, Cc, training, P(ws), wsb, st))
    close(dx[..., :Cc].permute(0, 3, 1, 2).cpu(), xr.grad, "dx")
    close(dgam.cpu(), bn.weight.grad, "dgamma")
    close(dbet.cpu(), bn.bias.grad, "dbeta")
    rc = lib.rdm_padavgpool2_bwd(P(gpd), P(xd), ld, row(sc, 0), row(sc, 1), P(gam), row(sc, 2), row(sc, 3), P(dx), ld, P(dgam), P(dbet),
                                 B, H, W, Cc, training, P(ws), wsb - 256, st)
    assert rc == -2                                                                     # RDM_ERR_WORKSPACE_TOO_SMALL, no launch
