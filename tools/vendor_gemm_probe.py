"""Calibration only: what does the vendor f32 GEMM (torch.matmul -> hipBLASLt / rocBLAS, exact f32, TF32 off) reach on the GEMM shapes of this
network?  Not used by the product (the tier's hot path is hand-written HIP); DESIGN.md quotes it beside the numbers of csrc/igemm.hip."""
import time
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda")
def bench(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / reps)
    return best
for name, M, N, K in (("e2 1x1 fwd", 69312, 2736, 320), ("e2 1x1 fwd K=96", 69312, 2736, 96), ("e2 1x1 dgrad", 69312, 336, 2736), ("long K", 69312, 2736, 1024),
                      ("e3 1x1 fwd", 17632, 1392, 704), ("e4 1x1 fwd", 4560, 720, 1536), ("e2 wgrad (K = pixels)", 2736, 336, 69312)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    t = bench(lambda: torch.matmul(A, W.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"{name:24s} M={M} N={N} K={K}: {t*1e3:.3f} ms {fl/t/1e12:.1f} TFLOP/s ({fl/t/1e12/157.3*100:.0f} % of the f32 MFMA peak)", flush=True)
