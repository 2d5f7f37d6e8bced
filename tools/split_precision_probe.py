"""Split-precision ("bf16x3" / "bf16x6") pricing on the network's 1x1 GEMM shapes (round-2 review, item 5) - an OPERATOR-LEVEL experiment:
a float32 operand is split into bf16 pieces a = a0 + a1 (+ a2), the product A.W^T is assembled from bf16 MFMA GEMMs with float32
accumulation (rdm_gemm_bf16, f32 output):  x3 = a0w0 + a0w1 + a1w0,  x6 = x3 + a1w1 + a0w2 + a2w0.
Reported per shape: max error relative to the output's max against a float64 matmul (the float32 MFMA kernel's own error beside it),
and the time of the 3 / 6 bf16 GEMMs (an UPPER bound on the speed of a fused kernel's MFMA+staging work; the piece extraction and
the adds are not counted) against the shipped float32 MFMA kernel.  Gate: error <= 2e-5 AND >= 1.3x.   python tools/split_precision_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, check, ptr, stream
L = _lib.lib(); dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def split(t, pieces):
    out, r = [], t.clone()
    for _ in range(pieces):
        p = r.bfloat16(); out.append(p); r = r - p.float()
    return out
def gemm_bf16(x, w, out):
    M, K = x.shape; N = w.shape[0]
    check(L.rdm_gemm_bf16(ptr(x), K, K, None, None, ptr(w), K, None, ptr(out), N, M, N, 1, None, 0, stream()))
for name, B, H, W_, K, N in [("dense_e2 conv1 (K=336)", 16, 57, 76, 336, 2736), ("dense_e2 conv1 (K=96)", 16, 57, 76, 96, 2736), ("dense_e3 conv1 (K=720)", 16, 29, 38, 720, 1392),
                             ("dense_e2 dgrad shape (K=2736)", 16, 57, 76, 2736, 336)]:
    M = B * H * W_
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.relu(torch.randn(M, K, generator=g) + 0.3).to(dev)          # post-ReLU activations
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    idx = torch.randint(0, M, (2048,), generator=g).to(dev)                # error on a row sample (the f64 reference of the full product is large)
    ref = a[idx].double() @ w.double().t()
    scale = ref.abs().max().item()
    # shipped f32 MFMA kernel
    d = ConvDesc(B, H, W_, K, K, N, N, 1, 1, 1, 1, 0, 0)
    y = torch.empty(M, N, device=dev); w1 = w.view(1, N, K).contiguous()
    t32 = timeit(lambda: check(L.rdm_conv2d_fwd(C.byref(d), ptr(a), ptr(w1), None, None, None, ptr(y), None, None, stream())))
    e32 = (y[idx].double() - ref).abs().max().item() / scale
    ap, wp = split(a, 3), split(w, 3)
    tmp = torch.empty(M, N, device=dev)
    tb = timeit(lambda: gemm_bf16(ap[0], wp[0], tmp))
    res = {}
    for label, terms in (("bf16x1", [(0, 0)]), ("bf16x3", [(0, 0), (0, 1), (1, 0)]), ("bf16x6", [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])):
        acc = torch.zeros(2048, N, dtype=torch.float32, device=dev)
        for i, j in reversed(terms):                                       # small terms first
            gemm_bf16(ap[i], wp[j], tmp)
            acc += tmp[idx]
        res[label] = (acc.double() - ref).abs().max().item() / scale
    fl = 2.0 * M * N * K
    print(f"{name}: f32 MFMA {t32:.3f} ms ({fl / t32 / 1e9:.0f} TF, err {e32:.1e}) | one bf16 GEMM {tb:.3f} ms | x3: {3 * tb:.3f} ms = {t32 / (3 * tb):.2f}x, err {res['bf16x3']:.1e} | "
          f"x6: {6 * tb:.3f} ms = {t32 / (6 * tb):.2f}x, err {res['bf16x6']:.1e} | plain bf16 err {res['bf16x1']:.1e}", flush=True)
