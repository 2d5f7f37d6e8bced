"""The 3x3 convolution of the few-pixel blocks (dense_e4: 16 x 15 x 19 x 720 -> 48, decoder d_1: 16 x 8 x 10 x 384 -> 48) on the direct
kernels (the plan's choice, at several K splits) against the Winograd F(2x2, 3x3) forward (f32 and three-way-split bf16): ms per launch.
    python tools/fewpix_3x3.py"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, check, ptr, stream
L = _lib.lib(); dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, Cb) in [(16, 15, 19, 720), (16, 8, 10, 384), (16, 29, 38, 1392)]:
    M = B * H * W
    y = torch.randn(M, Cb, device=dev); w = torch.randn(9, 48, Cb, device=dev) / (9 * Cb) ** 0.5
    sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
    out = torch.zeros(M, 48, device=dev)
    d = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
    res = []
    for split in (0, 1, 4, 8, 16):
        res.append((f"direct split {split}", round(timeit(lambda: check(L.rdm_conv2d_fwd_ex(C.byref(d), ptr(y), ptr(w), None, ptr(sc), ptr(sh), ptr(out), None, None, split, stream()))), 1)))
    for name, wsq, fwd in (("wino f32", L.rdm_conv3x3_wino_workspace_bytes, L.rdm_conv3x3_wino_fwd), ("wino x6", L.rdm_conv3x3_wino_x6_workspace_bytes, L.rdm_conv3x3_wino_fwd_x6)):
        for split in (0, 1, 2, 4):
            nb = int(wsq(Cb, B, H, W, split)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            try:
                res.append((f"{name} split {split} (+ weight transform)", round(timeit(lambda: check(fwd(C.byref(d), ptr(y), ptr(w), ptr(sc), ptr(sh), ptr(out), None, None, ptr(ws), nb, split, stream()))), 1)))
            except Exception as e:
                res.append((f"{name} split {split}", str(e)[:80]))
    print((B, H, W, Cb), "us per launch:")
    for r in res: print("   ", r, flush=True)
