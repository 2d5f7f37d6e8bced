#!/usr/bin/env python3
"""Secondary measurements for the HBM-bound kernels of the path: achieved GB/s of the
compulsory bytes against the MI355X HBM3E peak (8 TB/s spec; ~6.3 TB/s attainable).  One JSON line
per kernel.  bench.py remains the driver-facing headline benchmark."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from md_rdm_amd import _lib, filler  # noqa: E402
from md_rdm_amd.network import RDM_Net, computations as cp  # noqa: E402

PEAK = 8000.0


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def report(name, seconds, nbytes, note):
    gbs = nbytes / seconds / 1e9
    print(json.dumps({"kernel": name, "ms": round(seconds * 1e3, 4), "algorithmic_MB": round(nbytes / 1e6, 2), "achieved_GBps": round(gbs, 1),
                      "hbm_peak_GBps": PEAK, "frac": round(gbs / PEAK, 4), "note": note}), flush=True)


def main():
    dev = torch.device("cuda:0")
    B = 16
    L0 = _lib.lib()
    # attainable peaks on THIS device (denominators next to the datasheet values)
    n = 1 << 28                                                     # 1 GiB each way
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    t = timeit(lambda: _lib.check(_lib.bench_lib().rdm_microbench_copy(_lib.ptr(a), _lib.ptr(b), n, _lib.stream())), reps=10)
    report("microbench: float4 stream copy 1 GiB -> 1 GiB", t, 2 * n * 4, "attainable HBM3E rate (read + write)")
    del a, b
    sc = torch.empty(4096, device=dev)
    blocks, iters = 256 * 4, 20000
    t = timeit(lambda: _lib.check(_lib.bench_lib().rdm_microbench_mfma_f32(_lib.ptr(sc), blocks, iters, _lib.stream())), reps=3)
    tf = blocks * 4 * iters * 12 * 2048 / t / 1e12
    print(json.dumps({"kernel": "microbench: v_mfma_f32_16x16x4_f32 loop, 4 waves/SIMD, 12 accumulators", "ms": round(t * 1e3, 3),
                      "achieved_TFLOPs": round(tf, 1), "mfma_peak_TFLOPs": 157.3, "frac": round(tf / 157.3, 4),
                      "note": "attainable fp32 matrix-core rate at the clock the chip holds"}), flush=True)
    quant = RDM_Net.Quantization()
    # d_10 scale: 128x128 relative map, 64 pages of 16x16
    dn = torch.from_numpy(filler.log_uniform("bo.dn", (B, 1, 128, 128), 0.5, 2.0)).to(dev)
    dn1 = cp.resize(dn, 64)
    q, inv = quant.device_tables(7, dev)
    R = cp.ratio_grid_lloyd_paged(dn, dn1, q, inv)
    t = timeit(lambda: cp.ratio_grid_lloyd_paged(dn, dn1, q, inv))
    report("ratio_grid_lloyd_paged d_10 B=16", t, R.numel() * 8 + dn.numel() * 4 + dn1.numel() * 8, "compulsory: write 1024x256x64 f64 grid (+ read maps)")
    Rf = R.float()
    t = timeit(lambda: cp.als_pages(Rf, limit=100), reps=10)
    comp = Rf.numel() * 4 + 1024 * 256 * 4
    report("als_rank1 d_10 B=16 (1024 matrices, 100 iters) - compulsory bytes", t, comp,
           "R read once (register-resident), p written; two passes: squared errors + the first 8 iterates recorded (8 MB instead of the 106 MB full history), a late arg-min is replayed")
    t = timeit(lambda: cp.als_pages(R, limit=100), reps=10)
    report("als_rank1 d_10 B=16, float64 grid input", t, R.numel() * 8 + 1024 * 256 * 4, "the standalone operator on the f64 grid (f64 -> f32 at load)")
    t = timeit(lambda: cp.als_pages_fused(dn, dn1, q, inv, limit=100), reps=10)
    # issue floor of the fused kernel: per iteration and thread ~320 dependent VALU (two 64-term dot products + the residual) and 6 barriers;
    # 1024 matrices x 4 waves x 100 iterations x ~330 wave-instructions over 1024 SIMDs at ~4 cycles each -> ~0.22 ms at 2.4 GHz
    floor_ms = 1024 * 4 * 100 * 330 * 4 / 1024 / 2.4e9 * 1e3
    print(json.dumps({"kernel": "als_rank1_paged d_10 B=16 (grid + Lloyd formed inside the ALS load; what Ordinal_Layer.forward runs)", "ms": round(t * 1e3, 4),
                      "hbm_traffic_MB": round((dn.numel() * 4 + dn1.numel() * 8 + 1024 * 256 * 4 * 9) / 1e6, 2), "bound": "VALU issue + barriers (register-resident matrix)",
                      "valu_floor_ms": round(floor_ms, 3), "frac_of_floor": round(floor_ms / (t * 1e3), 3),
                      "note": "no float64 grid in HBM (the two-step path wrote and read 134 MB); traffic = the two maps + the recorded iterates; an HBM fraction would be meaningless here"}), flush=True)
    # decomposition / recombination at the harness sizes
    y = torch.from_numpy(filler.log_uniform("bo.y", (B, 1, 128, 128), 0.5, 9.5)).double().to(dev)
    t = timeit(lambda: cp.decompose_depth_map([], y, 7))
    report("decompose_depth_map n=7 B=16", t, y.numel() * 8 * (1 + 4 / 3), "read map, write packed pyramid")
    # fused AdamW over the flat parameter buffer
    n = 90_529_720
    p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
    L = _lib.lib()
    t = timeit(lambda: _lib.check(L.rdm_adamw_fused(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, _lib.stream())))
    report("adamw_fused 90.5M params", t, n * 28, "28 B/param: read p,g,m,v, write p,m,v")
    lg = torch.from_numpy(filler.uniform("bo.lg", (B, 180, 8, 10), -2, 3)).to(dev)
    t = timeit(lambda: cp.dorn_ordinal_regression(lg))
    report("dorn_fwd B=16 8x10", t, lg.numel() * 4 + lg.numel() // 2 * 8 + B * 80 * 8, "latency-bound (1.5 MB)")
    bench_input_pipeline(dev)
    bench_relative_decoder(dev)


def bench_relative_decoder(dev, B=16):
    """the WSM convolutions of the d_10 decoder at full width (RDM_Net.py:163-236, WSM_4: 208 channels at
    128x128) as fp32-MFMA conv launches with their fraction of the 157.3 TFLOP/s peak, and the whole d_10 head (64 paged ratio
    grids -> Lloyd -> rank-1 ALS -> reconstruct) as one figure."""
    import ctypes as C
    from md_rdm_amd._lib import ConvDesc
    L = _lib.lib()
    PEAK = 157.3

    def conv_line(name, Bn, H, W, cin, cout, kh, kw, ph, pw, note):
        cin_p, cout_p = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
        x = torch.randn(Bn, H, W, cin_p, device=dev)
        wp = torch.randn(kh * kw, cout_p, cin_p, device=dev) * 0.02
        Ho, Wo = H + 2 * ph - kh + 1, W + 2 * pw - kw + 1
        y = torch.empty(Bn, Ho, Wo, cout_p, device=dev)
        d = ConvDesc(Bn, H, W, cin_p, cin_p, cout_p, cout_p, kh, kw, 1, 1, ph, pw)
        t = timeit(lambda: _lib.check(L.rdm_conv2d_fwd(C.byref(d), _lib.ptr(x), _lib.ptr(wp), None, None, None, _lib.ptr(y), None, None, _lib.stream())), reps=10)
        fl = 2.0 * Bn * Ho * Wo * cout * cin * kh * kw            # algorithmic (unpadded) FLOPs
        print(json.dumps({"kernel": name, "ms": round(t * 1e3, 4), "algorithmic_GFLOP": round(fl / 1e9, 2), "achieved_TFLOPs": round(fl / t / 1e12, 1),
                          "mfma_peak_TFLOPs": PEAK, "frac": round(fl / t / 1e12 / PEAK, 4), "note": note}), flush=True)

    conv_line("WSM_4 conv2_2 5x5 52->52 @128x128 B=16 (d_10)", B, 128, 128, 52, 52, 5, 5, 2, 2, "the largest WSM conv by MACs per image at d_10 (RDM_Net.py:176); 52 channels pad to 64")
    conv_line("WSM_1 conv2_2 5x5 416->416 @16x16 B=16 (d_7..d_10)", B, 16, 16, 416, 416, 5, 5, 2, 2, "the widest 5x5 (WSM_1, 1664/4 channels)")
    conv_line("WSM_4 conv2_1 3x3 52->52 @128x128 B=16", B, 128, 128, 52, 52, 3, 3, 1, 1, "RDM_Net.py:175")
    conv_line("WSM_4 input_adjustment 1x1 416->208 @64x64 B=16", B, 64, 64, 416, 208, 1, 1, 0, 0, "RDM_Net.py:186")
    conv_line("WSM_4 deconv as 1x1 208->4x208 @64x64 B=16", B, 64, 64, 208, 4 * 208, 1, 1, 0, 0, "ConvTranspose2d(k=2,s=2) = 1x1 conv to 4 phases + pixel shuffle (:169)")
    conv_line("WSM_4 strip (3,128) conv 26->26 as (3,1) over rows-as-channels B=16", B, 128, 1, 128 * 32, 26, 3, 1, 1, 0, "one conv per row: K = 128 pixels x 32 (padded 26) channels x 3 (:181-184)")
    # the whole d_10 head on a (B,1,128,128) feature map
    quant = RDM_Net.Quantization()
    head = RDM_Net.Ordinal_Layer(10, False, quant)
    feat = torch.from_numpy(filler.log_uniform("bo.feat", (B, 1, 128, 128), 0.8, 1.25)).to(dev)
    with torch.no_grad():
        t = timeit(lambda: head(feat), reps=10)
    head_mb = (B * 128 * 128 * 4 + B * 64 * 64 * 8 + 1024 * 256 * 4 * 9 + B * 128 * 128 * 4 * 2) / 1e6
    print(json.dumps({"kernel": "d_10 head B=16: resize 128->64 + fused [64 paged ratio grids + Lloyd(128 table) + ALS(100)] + reconstruct", "ms": round(t * 1e3, 4),
                      "maps_per_s": round(B / t, 1), "hbm_traffic_MB": round(head_mb, 2), "bound": "VALU issue + barriers of the register-resident ALS (see als_rank1_paged above)",
                      "note": "round 3: the 134 MB float64 grid is no longer written and read back (round 2: 270 MB, 0.70 ms); the reference spends minutes here in Python loops (SURVEY 3.4)"}), flush=True)


def bench_input_pipeline(dev, B=16, H=480, W=640):
    """training_preprocess for a batch of raw NYU frames - GPU kernel chain (the Pillow CPU baseline of the same frames is
    timed in bench.py's cpu_baseline leg, the only bench code allowed to touch oracle/)."""
    import numpy as np
    from md_rdm_amd.dataloaders import nyu
    rng = np.random.default_rng(0)
    raws = [(rng.integers(0, 256, (H, W, 3)).astype(np.uint8), (rng.random((H, W)) * 9.5 + 0.5).astype(np.float32)) for _ in range(B)]
    draws = [nyu.draw_training_params(rng, (H, W)) for _ in range(B)]
    rgb = torch.from_numpy(np.stack([r for r, _ in raws])).to(dev)
    dep = torch.from_numpy(np.stack([d for _, d in raws])).to(dev)
    pre = nyu.NyuGpuPreprocessor()
    params = [p for _, p in draws]
    t = timeit(lambda: pre(rgb, dep, params), reps=20)
    nbytes = B * (H * W * 7 + 228 * 304 * 16)
    report("nyu training_preprocess B=16 480x640 -> 228x304 (inputs resident in HBM)", t, nbytes,
           f"{B / t:.0f} images/s; compulsory: read rgb u8 + depth f32, write x (3 planes) + y f32; 13 launches, intermediates stay in L2 / Infinity Cache")


if __name__ == "__main__":
    main()
