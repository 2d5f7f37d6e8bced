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
for (B, H, W, N, K) in ((16, 29, 38, 1392, 704), (16, 29, 38, 1392, 384), (16, 57, 76, 2736, 320), (16, 15, 19, 720, 1536)):
    M = B * H * W
    A = torch.randn(M, K, device=dev); Wt = torch.randn(N, K, device=dev) * 0.05
    Cn = torch.empty(M, N, device=dev); Cr = torch.empty(M, N, device=dev)
    d1 = ConvDesc(B, H, W, K, K, N, N, 1, 1, 1, 1, 0, 0)
    ref = lambda: check(L.rdm_conv2d_fwd(C.byref(d1), ptr(A), ptr(Wt), None, None, None, ptr(Cr), None, None, stream()))
    fl = 2.0 * M * N * K
    t0 = bench(ref)
    line = f"M={M} N={N} K={K}: shipped conv_fwd (no BN, plain store) {t0*1e3:.3f} ms {fl/t0/1e12:.1f} TF ({fl/t0/1e12/157.3*100:.0f}%)"
    # pre-tiled, pre-swizzled operand images for variant 40 (M, N padded to whole tiles)
    def tiled(T, rows_per_tile):
        R, Kk = T.shape
        Rp = (R + rows_per_tile - 1) // rows_per_tile * rows_per_tile
        P = torch.zeros(Rp, Kk, device=dev); P[:R] = T
        P = P.view(Rp // rows_per_tile, rows_per_tile, Kk // 32, 8, 4).permute(0, 2, 1, 3, 4).contiguous()      # [tile][slab][row][chunk][4]
        r = torch.arange(rows_per_tile, device=dev)
        src = (torch.arange(8, device=dev)[None, :] ^ ((r[:, None] >> 1) & 7))                                    # slot c of row r <- logical chunk c ^ ((r >> 1) & 7)
        return torch.gather(P, 3, src[None, None, :, :, None].expand(P.shape[0], P.shape[1], rows_per_tile, 8, 4).contiguous()).contiguous()
    At, Wtt = (tiled(A, 128), tiled(Wt, 96)) if K % 32 == 0 else (None, None)
    for v in (20, 32, 33, 34):
        if v == 40 and At is None: continue
        if v == 40:
            new = lambda: check(_lib.bench_lib().rdm_microbench_gemm_dma_f32(ptr(At), K, ptr(Wtt), K, ptr(Cn), N, M, N, K, v, stream()))
            t1 = bench(new)
            err = (Cn - Cr).abs().max().item() / Cr.abs().max().item()
            line += f" | v{v} (pre-tiled operands): {t1*1e3:.3f} ms {fl/t1/1e12:.1f} TF ({fl/t1/1e12/157.3*100:.0f}%) relerr {err:.1e}"
            continue
        new = lambda: check(_lib.bench_lib().rdm_microbench_gemm_dma_f32(ptr(A), K, ptr(Wt), K, ptr(Cn), N, M, N, K, v, stream()))
        t1 = bench(new)
        err = (Cn - Cr).abs().max().item() / Cr.abs().max().item()
        line += f" | v{v}: {t1*1e3:.3f} ms {fl/t1/1e12:.1f} TF ({fl/t1/1e12/157.3*100:.0f}%) relerr {err:.1e}"
    print(line, flush=True)
