"""In-process A/B microbenchmark of the conv kernels through the C ABI (the numbers DESIGN.md 4.1 quotes).

    SHAPE=B,H,W,Cb CIN=c VARIANTS=0,9 python tools/conv_microbench.py {fwd1|dg1|wg1|fwd3|dg3|wg3|all} [reps]

SHAPE defaults to the dense_e2 geometry (16,57,76,2736), CIN to 336; VARIANTS are rdm_debug_variant values timed alternately
in one process (box-to-box variance is 5-10 %, so only in-process comparisons are meaningful)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from md_rdm_amd import _lib
from md_rdm_amd._lib import ConvDesc, ptr, stream, check
L = _lib.lib()
dev = torch.device("cuda")
B, H, W, Cb = [int(v) for v in os.environ.get("SHAPE", "16,57,76,2736").split(",")]
M = B * H * W
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Y = torch.randn(M, Cb, device=dev)
sc = torch.rand(Cb, device=dev) + 0.5; sh = torch.randn(Cb, device=dev) * 0.3
w3 = torch.randn(9, 48, Cb, device=dev) * 0.01
out = torch.empty(M, 48, device=dev)
g48 = torch.randn(M, 48, device=dev)
dZ = torch.empty(M, Cb, device=dev)
dW3 = torch.zeros(9, 48, Cb, device=dev)
s0 = torch.zeros(Cb, dtype=torch.float64, device=dev); s1 = torch.zeros_like(s0)
d3 = ConvDesc(B, H, W, Cb, Cb, 48, 48, 3, 3, 1, 1, 1, 1)
CIN = int(os.environ.get("CIN", "336"))
X = torch.randn(M, CIN + 48, device=dev)
w1 = torch.randn(1, Cb, CIN, device=dev) * 0.05
sc1 = torch.rand(CIN, device=dev) + 0.5; sh1 = torch.randn(CIN, device=dev) * 0.3
d1 = ConvDesc(B, H, W, CIN, CIN + 48, Cb, Cb, 1, 1, 1, 1, 0, 0)
dW1 = torch.zeros(1, Cb, CIN, device=dev)
dX = torch.empty(M, CIN, device=dev)
t0s = torch.zeros(CIN, dtype=torch.float64, device=dev); t1s = torch.zeros_like(t0s)
def fwd3(): check(L.rdm_conv2d_fwd(C.byref(d3), ptr(Y), ptr(w3), None, ptr(sc), ptr(sh), ptr(out), None, None, stream()))
def wg3(): check(L.rdm_conv2d_wgrad(C.byref(d3), ptr(g48), ptr(Y), ptr(sc), ptr(sh), ptr(dW3), stream()))
def dg3(): check(L.rdm_conv2d_dgrad(C.byref(d3), ptr(g48), ptr(w3), ptr(dZ), Cb, ptr(Y), Cb, ptr(sc), ptr(sh), ptr(s0), ptr(s1), stream()))
def fwd1(): check(L.rdm_conv2d_fwd(C.byref(d1), ptr(X), ptr(w1), None, ptr(sc1), ptr(sh1), ptr(dZ), ptr(s0), ptr(s1), stream()))
def wg1(): check(L.rdm_conv2d_wgrad(C.byref(d1), ptr(dZ), ptr(X), ptr(sc1), ptr(sh1), ptr(dW1), stream()))
def dg1(): check(L.rdm_conv2d_dgrad(C.byref(d1), ptr(dZ), ptr(w1), ptr(dX), CIN, ptr(X), CIN + 48, ptr(sc1), ptr(sh1), ptr(t0s), ptr(t1s), stream()))
def fwd3n(): check(L.rdm_conv2d_fwd(C.byref(d3), ptr(Y), ptr(w3), None, None, None, ptr(out), None, None, stream()))
def wg3n(): check(L.rdm_conv2d_wgrad(C.byref(d3), ptr(g48), ptr(Y), None, None, ptr(dW3), stream()))
def fwd1n(): check(L.rdm_conv2d_fwd(C.byref(d1), ptr(X), ptr(w1), None, None, None, ptr(dZ), ptr(s0), ptr(s1), stream()))
def wg1n(): check(L.rdm_conv2d_wgrad(C.byref(d1), ptr(dZ), ptr(X), None, None, ptr(dW1), stream()))
ops = {"fwd3_nobn": (fwd3n, 2*M*48*Cb*9), "wg3_nobn": (wg3n, 2*M*48*Cb*9), "fwd1_nobn": (fwd1n, 2*M*Cb*CIN), "wg1_nobn": (wg1n, 2*M*Cb*CIN), "fwd3": (fwd3, 2*M*48*Cb*9), "wg3": (wg3, 2*M*48*Cb*9), "dg3": (dg3, 2*M*48*Cb*9), "fwd1": (fwd1, 2*M*Cb*CIN), "wg1": (wg1, 2*M*Cb*CIN), "dg1": (dg1, 2*M*Cb*CIN)}
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
for name, (fn, fl) in ops.items():
    if which != "all" and which != name: continue
    res = {v: [] for v in variants}
    for rnd in range(4):
        for v in variants:
            L.rdm_debug_variant(v)
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / reps)
    L.rdm_debug_variant(0)
    print(name + ": " + "  ".join(f"v{v}: {min(r)*1e3:.3f} ms {fl/min(r)/1e12:.1f} TF ({fl/min(r)/1e12/157.3*100:.0f}%)" for v, r in res.items()), flush=True)
