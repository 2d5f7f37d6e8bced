"""Op-level timing of the bf16 forward kernels at the layer shapes of BASELINE config 2 (B=8, 228x304): one line per
(kernel, shape, variant).  Development tool: python tools/bf16_microbench.py [variant ...]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from md_rdm_amd import _lib
L = _lib.lib(); P = _lib.ptr; dev = torch.device("cuda:0")
variants = [int(v) for v in sys.argv[1:]] or [0]
B = 8
shapes_g = [("e2.conv1", B*57*76, 240, 2736, 384), ("e3.k336", B*29*38, 336, 1408, 768), ("e3.k192", B*29*38, 192, 1408, 768), ("e3.conv1", B*29*38, 456, 1392, 768), ("e4.conv1", B*15*19, 1224, 720, 2112), ("d1.conv1", B*8*10, 1632, 384, 2208),
            ("stem", B*114*152, 160, 96, 160), ("trans_e2", B*29*38, 384, 192, 384)]
shapes_c = [("e2.conv2", B, 57, 76, 2736), ("e3.conv2", B, 29, 38, 1392), ("e4.conv2", B, 15, 19, 720), ("d1.conv2", B, 8, 10, 384)]
def timeit(fn, n=int(os.environ.get('REPS', '30'))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
only = os.environ.get("ONLY", "")
for v in variants:
    L.rdm_debug_variant(v)
    for name, M, K, N, ldx in (shapes_g if only in ("", "gemm") else []):
        X = torch.randn(M, ldx, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.rand(K, device=dev) - 0.5
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        st = _lib.stream()
        wsg = torch.empty(8 * M * N * 4 if M <= 1024 else 256, dtype=torch.uint8, device=dev)
        for pro in (1, 0):
            us = timeit(lambda: _lib.check(L.rdm_gemm_bf16(P(X), ldx, K, P(sc) if pro else None, P(sh) if pro else None, P(W), K, None, P(out), N, M, N, 0,
                                                           P(wsg) if M <= 1024 else None, wsg.numel() if M <= 1024 else 0, st)))
            fl = 2.0 * M * N * K; by = 2.0 * (M * K + N * K + M * N)
            print(f"v{v} gemm {name:9s} M={M:6d} K={K:5d} N={N:5d} prologue={pro}: {us:8.1f} us  {fl/us/1e6:7.1f} TF  {by/us/1e3:7.1f} GB/s", flush=True)
        osc = torch.rand(N, device=dev) + 0.5; osh = torch.rand(N, device=dev) - 0.5
        us = timeit(lambda: _lib.check(L.rdm_gemm_bf16_act(P(X), ldx, K, P(sc), P(sh), P(W), K, P(osc), P(osh), P(out), N, M, N, None, 0, st)))
        print(f"v{v} gemm_act {name:9s} M={M:6d} K={K:5d} N={N:5d}: {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)
    for name, b, H, W_, Cc in (shapes_c if only in ("", "conv3") else []):
        M = b * H * W_
        Y = torch.randn(M, Cc, device=dev).bfloat16(); Wp = (torch.randn(9, 48, Cc, device=dev) * 0.02).bfloat16()
        sc = torch.rand(Cc, device=dev) + 0.5; sh = torch.rand(Cc, device=dev) - 0.5
        out = torch.empty(M, 48, dtype=torch.bfloat16, device=dev)
        wsb = int(L.rdm_conv3x3_bf16_workspace_bytes(Cc, b, H, W_)); ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        st = _lib.stream()
        for use_ws in (1, 0):
            us = timeit(lambda: _lib.check(L.rdm_conv3x3_bf16(P(Y), Cc, Cc, P(sc), P(sh), P(Wp), P(out), 48, b, H, W_, P(ws) if (use_ws and wsb) else None, wsb if use_ws else 0, st)))
            fl = 2.0 * M * 48 * Cc * 9
            print(f"v{v} conv3 {name:9s} M={M:6d} C={Cc:5d} split_ws={use_ws}: {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)
    for name, b, H, W_, Cc in (shapes_c if only in ("", "act") else []):                       # the activated-input form (LDS-DMA, in-launch combine)
        M = b * H * W_
        Cp = (Cc + 31) // 32 * 32
        Y = torch.rand(M, Cp, device=dev).bfloat16(); w = torch.randn(48, Cc, 3, 3, device=dev) * 0.02
        wimg = torch.empty(int(L.rdm_conv3x3_act_bf16_weight_bytes(Cc)), dtype=torch.uint8, device=dev)
        st = _lib.stream()
        _lib.check(L.rdm_conv3x3_act_bf16_pack(P(w), Cc, P(wimg), st))
        out = torch.empty(M, 48, dtype=torch.bfloat16, device=dev)
        wsb = int(L.rdm_conv3x3_act_bf16_workspace_bytes(Cp, b, H, W_)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        for use_ws in (1, 0):
            us = timeit(lambda: _lib.check(L.rdm_conv3x3_act_bf16(P(Y), Cp, Cp, P(wimg), P(out), 48, b, H, W_, P(ws) if use_ws else None, wsb if use_ws else 0, st)))
            fl = 2.0 * M * 48 * Cc * 9
            print(f"v{v} conv3act {name:9s} M={M:6d} C={Cc:5d} split_ws={use_ws}: {us:8.1f} us  {fl/us/1e6:7.1f} TF", flush=True)
