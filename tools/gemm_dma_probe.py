"""LDS-DMA f32 GEMM experiment vs the shipped 1x1 forward kernel at dense_e2 / e3 shapes (C = A W^T, both k-contiguous)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, ptr, stream, check
L = _lib.lib()
dev = torch.device("cuda")
def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / reps)
    return best
for (B, H, W, N, K) in ((16, 57, 76, 2736, 320), (16, 57, 76, 2736, 96), (16, 29, 38, 1392, 704), (16, 15, 19, 720, 1536)):
    M = B * H * W
    A = torch.randn(M, K, device=dev); Wt = torch.randn(N, K, device=dev) * 0.05
    Cn = torch.empty(M, N, device=dev); Cr = torch.empty(M, N, device=dev)
    d1 = ConvDesc(B, H, W, K, K, N, N, 1, 1, 1, 1, 0, 0)
    ref = lambda: check(L.rdm_conv2d_fwd(C.byref(d1), ptr(A), ptr(Wt), None, None, None, ptr(Cr), None, None, stream()))
    fl = 2.0 * M * N * K
    t0 = bench(ref)
    line = f"M={M} N={N} K={K}: shipped conv_fwd (no BN, plain store) {t0*1e3:.3f} ms {fl/t0/1e12:.1f} TF ({fl/t0/1e12/157.3*100:.0f}%)"
    for v in (3, 803, 20, 820, 420, 30, 830):
        new = lambda: check(L.rdm_microbench_gemm_dma_f32(ptr(A), K, ptr(Wt), K, ptr(Cn), N, M, N, K, v, stream()))
        t1 = bench(new)
        err = (Cn - Cr).abs().max().item() / Cr.abs().max().item()
        line += f" | v{v}: {t1*1e3:.3f} ms {fl/t1/1e12:.1f} TF ({fl/t1/1e12/157.3*100:.0f}%) relerr {err:.1e}"
    print(line, flush=True)
