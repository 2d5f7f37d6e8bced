"""How many conv3x3_bf16 / gemm_bf16 workgroups run per CU at once?  Time vs grid size (steps at multiples of 256 x k reveal k)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
L = _lib.lib(); P = _lib.ptr; dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
H = W = 64; Cc = 512
for B in (8, 16, 20, 24, 32, 40, 48, 64, 80, 96):
    M = B * H * W
    Y = torch.randn(M, Cc, device=dev).bfloat16(); Wp = (torch.randn(9, 48, Cc, device=dev) * 0.02).bfloat16()
    sc = torch.rand(Cc, device=dev) + 0.5; sh = torch.rand(Cc, device=dev) - 0.5
    out = torch.empty(M, 48, dtype=torch.bfloat16, device=dev)
    st = _lib.stream()
    us = timeit(lambda: _lib.check(L.rdm_conv3x3_bf16(P(Y), Cc, Cc, P(sc), P(sh), P(Wp), P(out), 48, B, H, W, None, 0, st)))
    print(f"conv3 tiles(256px)={M // 256:4d}  {us:8.1f} us   per-tile-round {us / max(1, -(-M // 256 // 256)):7.1f}", flush=True)
