"""Isolated speed of the BatchNorm-backward apply pass (rdm_bn_bwd) at the layer shapes of the B=16 228x304 step."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
for name, M, Cn, acc in (("e2 norm2", 69312, 2736, 0), ("e3 norm2", 17632, 1392, 0), ("e4 norm2", 4560, 720, 0), ("e2 norm1", 69312, 336, 1), ("e3 norm1", 17632, 720, 1),
                         ("e4 norm1", 4560, 1600, 1)):
    dz = torch.randn(M, Cn, device=dev); x = torch.randn(M, Cn, device=dev); dx = torch.zeros(M, Cn, device=dev)
    s0 = torch.randn(Cn, device=dev, dtype=torch.float64); s1 = torch.randn(Cn, device=dev, dtype=torch.float64)
    g = torch.rand(Cn, device=dev) + 0.5; mu = torch.randn(Cn, device=dev); rs = torch.rand(Cn, device=dev) + 0.5
    dg = torch.empty(Cn, device=dev); db = torch.empty(Cn, device=dev)
    def run():
        _lib.check(L.rdm_bn_bwd(_lib.ptr(dx), Cn, _lib.ptr(dz), Cn, _lib.ptr(x), Cn, _lib.ptr(s0), _lib.ptr(s1), float(M), _lib.ptr(g), _lib.ptr(mu), _lib.ptr(rs),
                                _lib.ptr(dg), _lib.ptr(db), M, Cn, acc, 1, _lib.stream()))
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    by = M * Cn * 4 * (4 if acc else 3)
    print(f"{name}: M={M} C={Cn} acc={acc}: {dt*1e6:8.1f} us  {by/dt/1e12:5.2f} TB/s", flush=True)
