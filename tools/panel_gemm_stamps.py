"""In-kernel phase stamps of gemm_panel_bf16_kernel (development build: RDM_DEV_VARIANTS=1 python -m md_rdm_amd.build): cycles per item spent
waiting for the DMA pieces, at the barrier, issuing the next item's pieces, multiplying, and in the epilogue, per wave of one workgroup."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
L = _lib.lib(); P = _lib.ptr; dev = torch.device("cuda:0")
M, K, N, ldx = 8*57*76, 240, 2736, 384
X = torch.randn(M, ldx, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
sc = torch.rand(K, device=dev) + 0.5; sh = torch.rand(K, device=dev) - 0.5
osc = torch.rand(N, device=dev) + 0.5; osh = torch.rand(N, device=dev) - 0.5
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
dbg = torch.zeros(4096, dtype=torch.float32, device=dev)
st = _lib.stream()
L.rdm_debug_variant(208)
for _ in range(3):
    dbg.zero_()
    _lib.check(L.rdm_gemm_bf16_act(P(X), ldx, K, P(sc), P(sh), P(W), K, P(osc), P(osh), P(out), N, M, N, P(dbg), dbg.numel() * 4, st))
    torch.cuda.synchronize()
d = dbg.cpu().view(-1, 8)[:8]
print("wave: wait  barrier  request  phaseA  phaseB  items   (cycles per item)")
for w in range(8):
    n = max(d[w, 5].item(), 1)
    print(w, [round(d[w, i].item() / n) for i in range(5)], int(n))
