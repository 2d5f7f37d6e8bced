"""Price of a per-XCD barrier and a same-XCD hand-off through the L2 (csrc/bench/xcd_sync.hip): the numbers behind DESIGN.md 7's plan for a
per-image persistent kernel over the few-pixel blocks of the bf16 forward (60 layers x 2 syncs per B=8 forward).
    python tools/xcd_sync_probe.py            -> profiles/rNN_xcd_sync.txt is this script's stdout"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from md_rdm_amd import _lib
B = _lib.bench_lib()
dev = torch.device("cuda")
state = torch.zeros(384, dtype=torch.int32, device=dev)
for blocks in (256, 128):
    for payload in (0, 1024, 4096, 16384):          # floats per workgroup and round: 0, 4 KB, 16 KB, 64 KB
        slots = torch.zeros(8 * 256 * max(payload, 4), device=dev)
        res = torch.zeros(blocks * 8, dtype=torch.int32, device=dev)
        for rounds in (0, 200):
            _lib.check(B.rdm_microbench_xcd_sync(_lib.ptr(state), _lib.ptr(slots), payload, rounds, blocks, _lib.ptr(res), _lib.stream()))
            torch.cuda.synchronize()
            r = res.view(blocks, 8).cpu().numpy().astype("int64") & 0xFFFFFFFF
            if rounds == 0:
                base = r[:, 4].max()
                continue
            per_xcc = {int(k): int((r[:, 0] == k).sum()) for k in sorted(set(r[:, 0]))}
            agree = all(int(r[i, 2]) == per_xcc[int(r[i, 0])] for i in range(blocks))
            same_mod8 = all(len(set(int(b) % 8 for b in range(blocks) if r[b, 0] == k)) == 1 for k in per_xcc)
            us = (r[:, 4].max() - base) / 100.0 / rounds          # 100 MHz wall clock
            print(f"blocks {blocks:3d} payload {payload * 4 // 1024:3d} KB/WG: workgroups per XCC {per_xcc} (n_k consistent: {agree}; blockIdx % 8 constant within an XCC: {same_mod8}); "
                  f"{us:6.2f} us per round = write + barrier + neighbour read (sc1) + barrier; stale values read: {int(r[:, 3].sum())}; timeouts: {int(r[:, 6].sum())}")
