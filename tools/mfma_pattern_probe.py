"""Register-only MFMA rate: one operand pair for all MFMAs (k_mfma_loop) vs the conv kernels' 4 x 3 tile pattern with distinct operands."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
L = _lib.lib()
sc = torch.zeros(1 << 20, device="cuda")
for blocks in (256 * 1, 256 * 4):
    for name, iters in (("same operands", 4000), ("tile pattern, registers only", -16000), ("+ ds_read_b128 fragments", -16004), ("+ barrier per slab", -16008), ("+ LDS-DMA staging", -16012), ("+ LDS-DMA staging, A tile only", -16016), ("+ LDS-DMA issued, never waited for", -16020), ("+ LDS-DMA staging, buffer form (SRD + 32-bit offset)", -16024)):
        _lib.check(_lib.bench_lib().rdm_microbench_mfma_f32(_lib.ptr(sc), blocks, iters, _lib.stream())); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): _lib.check(_lib.bench_lib().rdm_microbench_mfma_f32(_lib.ptr(sc), blocks, iters, _lib.stream()))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
        n = (abs(iters) if iters > 0 else abs(iters) // 4 * 4) * 12                     # MFMAs per wave (both loops: 12 per iteration unit; the tile loop runs iters/4 x 48)
        fl = blocks * 4 * n * 2048.0
        print(f"{blocks // 256} workgroups/CU, {name}: {fl / dt / 1e12:.1f} TFLOP/s", flush=True)
