import ctypes as C, sys, os, torch
sys.path.insert(0, ".")
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, check, ptr, stream
L = _lib.lib(); dev = torch.device("cuda:0")
for (B, H, W, Cb) in [(16, 57, 76, 2736), (16, 29, 38, 1392)]:
    M = B * H * W
    y = torch.randn(M, Cb, device=dev); w = torch.randn(9, 48, Cb, device=dev) / (9 * Cb) ** 0.5
    sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
    out = torch.empty(M, 48, device=dev)
    d = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
    def timeit(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    res = []
    for name, wsq, fwd in (("f32", L.rdm_conv3x3_wino_workspace_bytes, L.rdm_conv3x3_wino_fwd), ("x6", L.rdm_conv3x3_wino_x6_workspace_bytes, L.rdm_conv3x3_wino_fwd_x6)):
        nb = int(wsq(Cb, B, H, W, 0)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        res.append((name, round(timeit(lambda: check(fwd(C.byref(d), ptr(y), ptr(w), ptr(sc), ptr(sh), ptr(out), None, None, ptr(ws), nb, 0, stream()))), 3)))
    print("abl", os.environ.get("RDM_WX6_ABL", "0"), (B, H, W, Cb), res, flush=True)
