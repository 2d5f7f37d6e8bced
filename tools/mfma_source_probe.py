"""Where must a slab come from for the MFMA loop to keep its rate?  k_mfma_loop_tile mode 6 (buffer-form LDS-DMA staging of a
128 x 96 x 16 slab next to its 48 MFMAs per wave) with the DMA source window sized for L1, L2, the Infinity Cache and HBM."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
L = _lib.lib()
n = (1 << 29) + 16384                      # 2 GiB of floats + slack
sc = torch.zeros(n, device="cuda")
for blocks in (256 * 2, 256 * 4):
    for span in (1 << 14, 1 << 17, 1 << 19, 1 << 21, 1 << 23, 1 << 25, 1 << 27, 1 << 29):
        run = lambda: _lib.check(_lib.bench_lib().rdm_microbench_mfma_staged_f32(_lib.ptr(sc), n, blocks, 4000, 6, span, _lib.stream()))
        run(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        fl = blocks * 4 * 4000 * 48 * 2048.0
        print(f"{blocks // 256} workgroups/CU, source window {span * 4 / 2**20:8.2f} MiB: {fl / dt / 1e12:6.1f} TFLOP/s", flush=True)
