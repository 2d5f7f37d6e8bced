"""In-process A/B of the 3x3 forward: Winograd F(2x2,3x3) (rdm_conv3x3_wino_fwd) vs the direct halo kernel (rdm_conv2d_fwd), dense_e2 / e3
shapes at the bench batch, BN-ReLU prologue on.  Prints ms and algorithmic TFLOP/s (2*M*48*9*C / time)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import os
from md_rdm_amd import _lib
if os.environ.get("RDM_LIB"):                        # A/B of differently built libraries (development only)
    _lib.LIB_PATH = os.environ["RDM_LIB"]
from md_rdm_amd._lib import ConvDesc, check, ptr, stream
L = _lib.lib()
dev = torch.device("cuda:0")
splits = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
shapes = [(16, 57, 76, 2736), (16, 29, 38, 1392), (16, 15, 19, 720)]
if len(sys.argv) > 2:
    shapes = shapes[:int(sys.argv[2])]
if os.environ.get("SHAPES"):                          # e.g. SHAPES=8,22,76,720;8,44,152,1392
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
for (B, H, W, Cb) in shapes:
    M = B * H * W
    y = torch.randn(M, Cb, device=dev); w = torch.randn(9, 48, Cb, device=dev) / (9 * Cb) ** 0.5
    sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
    out = torch.empty(M, 48, device=dev)
    d = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
    fl = 2.0 * M * 48 * 9 * Cb
    def timeit(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t = timeit(lambda: check(L.rdm_conv2d_fwd(C.byref(d), ptr(y), ptr(w), None, ptr(sc), ptr(sh), ptr(out), None, None, stream())))
    print(f"{B}x{H}x{W} C={Cb}: direct halo {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s", flush=True)
    for sp in splits:
        nb = int(L.rdm_conv3x3_wino_workspace_bytes(Cb, B, H, W, sp))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        t = timeit(lambda: check(L.rdm_conv3x3_wino_fwd(C.byref(d), ptr(y), ptr(w), ptr(sc), ptr(sh), ptr(out), None, None, ptr(ws), nb, sp, stream())))
        print(f"    winograd split {sp}: {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s (algorithmic)", flush=True)
    # weight gradient of the same conv: direct row kernel (rdm_conv2d_wgrad, accumulating with f32 atomics) vs Winograd F(3x3,2x2)
    go = torch.randn(M, 48, device=dev); dw = torch.zeros(9, 48, Cb, device=dev)
    t = timeit(lambda: check(L.rdm_conv2d_wgrad(C.byref(d), ptr(go), ptr(y), ptr(sc), ptr(sh), ptr(dw), stream())))
    print(f"    wgrad direct {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s", flush=True)
    nb = int(L.rdm_conv3x3_wino_wgrad_workspace_bytes(Cb, B, H, W))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    t = timeit(lambda: check(L.rdm_conv3x3_wino_wgrad(C.byref(d), ptr(go), ptr(y), ptr(sc), ptr(sh), ptr(dw), ptr(ws), nb, stream())))
    print(f"    wgrad winograd {t:.3f} ms = {fl / t / 1e9:.1f} TFLOP/s (algorithmic)", flush=True)
