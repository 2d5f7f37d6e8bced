"""In-process A/B of the split-precision (bf16x3) gradient kernels (csrc/xsplit.hip) against the exact-f32 MFMA kernels they replace, through the
C ABI, on the network's shapes, with the error of both against a float64 product.

    python tools/xsplit_bench.py [wg1] [reps]
"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, ptr, stream, check
L = _lib.lib()
dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "wg1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def timeit(fns):
    res = {k: [] for k in fns}
    for rnd in range(4):
        for k, fn in fns.items():
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / reps)
    return {k: min(v) for k, v in res.items()}


def wg1(B, H, W, Cb, cin, ld):
    M = B * H * W
    g = torch.Generator(device="cpu").manual_seed(cin)
    X = torch.randn(M, ld, device=dev)
    dZ = torch.randn(M, Cb, device=dev)
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.3
    d = ConvDesc(B, H, W, cin, ld, Cb, Cb, 1, 1, 1, 1, 0, 0)
    dW = torch.zeros(Cb, cin, device=dev)
    def f32(): check(L.rdm_conv2d_wgrad(C.byref(d), ptr(dZ), ptr(X), ptr(sc), ptr(sh), ptr(dW), stream()))
    def x3(): check(L.rdm_conv2d_wgrad_x3(C.byref(d), ptr(dZ), ptr(X), ptr(sc), ptr(sh), ptr(dW), 0, 0, stream()))
    # error vs float64 on a slice of the output rows (the full product is 2 x 69312 x 2736 x 336 flops in f64 on the GPU: fine)
    a = torch.relu(X[:, :cin].double() * sc.double() + sh.double())
    want = dZ.double().t() @ a
    errs = {}
    for k, fn in (("f32", f32), ("x3", x3)):
        dW.zero_(); fn(); torch.cuda.synchronize()
        errs[k] = ((dW.double() - want).abs().max() / want.abs().max()).item()
    del a, want
    t = timeit({"f32": f32, "x3": x3})
    fl = 2.0 * M * Cb * cin
    by = 4.0 * M * (Cb + cin)
    print(f"wg1 M={M} N={Cb} C={cin}: f32 {t['f32']*1e3:.3f} ms {fl/t['f32']/1e12:.0f} TF err {errs['f32']:.1e} | x3 {t['x3']*1e3:.3f} ms {fl/t['x3']/1e12:.0f} TF-equiv "
          f"({by/t['x3']/1e12:.2f} TB/s algorithmic) err {errs['x3']:.1e} | speedup {t['f32']/t['x3']:.2f}x", flush=True)


def dg3(B, H, W, Cb):
    M = B * H * W
    Y = torch.randn(M, Cb, device=dev)
    sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
    w3 = torch.randn(9, 48, Cb, device=dev) * 0.05
    g48 = torch.randn(M, 48, device=dev)
    dZ = torch.empty(M, Cb, device=dev)
    s0 = torch.zeros(Cb, dtype=torch.float64, device=dev); s1 = torch.zeros_like(s0)
    d3 = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
    wsb = L.rdm_conv3x3_dgrad_x3_workspace_bytes(Cb)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def f32(): check(L.rdm_conv2d_dgrad(C.byref(d3), ptr(g48), ptr(w3), ptr(dZ), Cb, ptr(Y), Cb, ptr(sc), ptr(sh), ptr(s0), ptr(s1), stream()))
    def x3(): check(L.rdm_conv3x3_dgrad_x3(C.byref(d3), ptr(g48), ptr(w3), ptr(dZ), Cb, ptr(Y), Cb, ptr(sc), ptr(sh), ptr(s0), ptr(s1), ptr(ws), wsb, 0, stream()))
    outs = {}
    for k, fn in (("f32", f32), ("x3", x3)):
        fn(); torch.cuda.synchronize(); outs[k] = dZ[:4096].clone()
    diff = ((outs["x3"] - outs["f32"]).abs().max() / outs["f32"].abs().max()).item()
    t = timeit({"f32": f32, "x3": x3})
    fl = 2.0 * M * Cb * 432
    by = 4.0 * M * (2 * Cb + 48)
    print(f"dg3 M={M} Cb={Cb}: f32 {t['f32']*1e3:.3f} ms {fl/t['f32']/1e12:.0f} TF | x3 {t['x3']*1e3:.3f} ms {fl/t['x3']/1e12:.0f} TF-equiv ({by/t['x3']/1e12:.2f} TB/s algorithmic) "
          f"| x3 vs f32 kernel {diff:.1e} | speedup {t['f32']/t['x3']:.2f}x", flush=True)


def dg1(B, H, W, Cb, cin, ld):
    M = B * H * W
    X = torch.randn(M, ld, device=dev)
    dZ = torch.randn(M, Cb, device=dev)
    w1 = torch.randn(Cb, cin, device=dev) * 0.05
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.3
    d = ConvDesc(B, H, W, cin, cin, Cb, Cb, 1, 1, 1, 1, 0, 0)
    dX = torch.empty(M, cin, device=dev)
    s0 = torch.zeros(cin, dtype=torch.float64, device=dev); s1 = torch.zeros_like(s0)
    wsb = L.rdm_conv1x1_dgrad_x3_workspace_bytes(Cb, cin)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def f32(): check(L.rdm_conv2d_dgrad(C.byref(d), ptr(dZ), ptr(w1), ptr(dX), cin, ptr(X), ld, ptr(sc), ptr(sh), ptr(s0), ptr(s1), stream()))
    def x3(): check(L.rdm_conv1x1_dgrad_x3(C.byref(d), ptr(dZ), ptr(w1), ptr(dX), cin, ptr(X), ld, ptr(sc), ptr(sh), ptr(s0), ptr(s1), ptr(ws), wsb, 0, stream()))
    outs = {}
    for k, fn in (("f32", f32), ("x3", x3)):
        fn(); torch.cuda.synchronize(); outs[k] = dX[:4096].clone()
    diff = ((outs["x3"] - outs["f32"]).abs().max() / outs["f32"].abs().max()).item()
    t = timeit({"f32": f32, "x3": x3})
    fl = 2.0 * M * Cb * cin
    by = 4.0 * M * (Cb + 2 * cin)
    print(f"dg1 M={M} K={Cb} N={cin}: f32 {t['f32']*1e3:.3f} ms {fl/t['f32']/1e12:.0f} TF | x3 {t['x3']*1e3:.3f} ms {fl/t['x3']/1e12:.0f} TF-equiv ({by/t['x3']/1e12:.2f} TB/s algorithmic) "
          f"| x3 vs f32 kernel {diff:.1e} | speedup {t['f32']/t['x3']:.2f}x", flush=True)


def wg3(B, H, W, Cb):
    M = B * H * W
    Y = torch.randn(M, Cb, device=dev)
    sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
    g48 = torch.randn(M, 48, device=dev)
    d3 = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
    dW = torch.zeros(9, 48, Cb, device=dev)
    wsb = L.rdm_conv3x3_wino_wgrad_workspace_bytes(Cb, B, H, W)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def f32(): check(L.rdm_conv3x3_wino_wgrad(C.byref(d3), ptr(g48), ptr(Y), ptr(sc), ptr(sh), ptr(dW), ptr(ws), wsb, 0, stream()))
    def x3(): check(L.rdm_conv2d_wgrad_x3(C.byref(d3), ptr(g48), ptr(Y), ptr(sc), ptr(sh), ptr(dW), 0, 0, stream()))
    outs = {}
    for k, fn in (("f32", f32), ("x3", x3)):
        dW.zero_(); fn(); torch.cuda.synchronize(); outs[k] = dW.clone()
    diff = ((outs["x3"] - outs["f32"]).abs().max() / outs["f32"].abs().max()).item()
    t = timeit({"f32": f32, "x3": x3})
    fl = 2.0 * M * Cb * 432
    by = 4.0 * M * (Cb + 48)
    print(f"wg3 M={M} Cb={Cb}: winograd f32 {t['f32']*1e3:.3f} ms {fl/t['f32']/1e12:.0f} TF | x3 {t['x3']*1e3:.3f} ms {fl/t['x3']/1e12:.0f} TF-equiv ({by/t['x3']/1e12:.2f} TB/s algorithmic) "
          f"| x3 vs winograd f32 {diff:.1e} | speedup {t['f32']/t['x3']:.2f}x", flush=True)


def fw1(B, H, W, Cb, cin, ld):
    M = B * H * W
    X = torch.randn(M, ld, device=dev)
    w1 = torch.randn(Cb, cin, device=dev) * 0.05
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev) * 0.3
    d = ConvDesc(B, H, W, cin, ld, Cb, Cb, 1, 1, 1, 1, 0, 0)
    Y = torch.empty(M, Cb, device=dev)
    s0 = torch.zeros(Cb, dtype=torch.float64, device=dev); s1 = torch.zeros_like(s0)
    wsb = L.rdm_conv1x1_fwd_x6_workspace_bytes(cin, Cb)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def f32(): check(L.rdm_conv2d_fwd(C.byref(d), ptr(X), ptr(w1), None, ptr(sc), ptr(sh), ptr(Y), ptr(s0), ptr(s1), stream()))
    def x6(): check(L.rdm_conv1x1_fwd_x6(C.byref(d), ptr(X), ptr(w1), ptr(sc), ptr(sh), ptr(Y), ptr(s0), ptr(s1), ptr(ws), wsb, 0, stream()))
    a = torch.relu(X[:8192, :cin].double() * sc.double() + sh.double())
    want = a @ w1.double().t()
    errs = {}
    for k, fn in (("f32", f32), ("x6", x6)):
        fn(); torch.cuda.synchronize()
        errs[k] = ((Y[:8192].double() - want).abs().max() / want.abs().max()).item()
    t = timeit({"f32": f32, "x6": x6})
    fl = 2.0 * M * Cb * cin
    by = 4.0 * M * (Cb + cin)
    print(f"fw1 M={M} N={Cb} K={cin}: f32 {t['f32']*1e3:.3f} ms {fl/t['f32']/1e12:.0f} TF err {errs['f32']:.1e} | x6 {t['x6']*1e3:.3f} ms {fl/t['x6']/1e12:.0f} TF-equiv "
          f"({by/t['x6']/1e12:.2f} TB/s algorithmic) err {errs['x6']:.1e} | speedup {t['f32']/t['x6']:.2f}x", flush=True)


if which in ("fw1", "all"):
    for cin in (96, 192, 336):
        fw1(16, 57, 76, 2736, cin, 384)
    for cin in (192, 480, 720):
        fw1(16, 29, 38, 1392, cin, 768)
if which in ("wg3", "all"):
    wg3(16, 57, 76, 2736)
    wg3(16, 29, 38, 1392)
if which in ("dg1", "all"):
    for cin in (96, 192, 336):
        dg1(16, 57, 76, 2736, cin, 384)
    for cin in (192, 480, 720):
        dg1(16, 29, 38, 1392, cin, 768)
if which in ("dg3", "all"):
    dg3(16, 57, 76, 2736)
    dg3(16, 29, 38, 1392)
if which in ("wg1", "all"):
    for cin in (96, 144, 192, 240, 288, 336):
        wg1(16, 57, 76, 2736, cin, 384)
    for cin in (192, 336, 480, 720):
        wg1(16, 29, 38, 1392, cin, 768)
