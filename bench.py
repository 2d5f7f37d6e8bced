#!/usr/bin/env python3
"""Headline benchmark: depth-maps/s of one full training step (forward + losses + backward +
AdamW) of DepthEstimationNet at NYU geometry 228x304, batch 16 per GPU (BASELINE.json metric /
configs[2]), synthetic data, deterministic hash-filled weights.  Arithmetic of the shipped (parity) configuration:
float32 tensors everywhere; conv1 of dense_e2 / e3 forward on three-way-split bf16 MFMAs (float32-equivalent), the gradient
GEMMs of the dense blocks on two-way-split bf16 MFMAs ("bf16x3", ~5e-6 of a gradient's maximum), everything else on the
exact-f32 MFMA.  `--forward-split 0 --backward-precision f32` = exact-f32 MFMA everywhere.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL).

One JSON line on rank 0 (< 4 KB: the driver's record keeps a bounded tail of stdout; tests/test_bench_line.py holds the
size), with
  roofline     - the DOMINANT kernel by summed duration: FLOPs it executes on its matrix pipe / its average launch
                 duration measured with HIP events on its launch stream, against that pipe's dense peak
  cpu_baseline - the oracle (CPU restatement of the reference, oracle/) timed on the host cores on a
                 bounded sample (batch-16 full train steps incl. AdamW) - a reported baseline, never the thing shipped.
Everything else that used to ride on the line (per-kernel table, the other BASELINE configs in full, the prose on the
arithmetic) goes to DETAIL_FILE next to this script; the line names it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# Multi-process GPU work on this ROCm stack (RCCL, tensors shared across ranks) needs dmabuf IPC: the host driver does not support the legacy
# IPC mode and hipIpcGetMemHandle fails with "invalid argument" without this.  It must be in the environment BEFORE the first HIP call of the
# process (the driver's launcher exports it as well; setdefault keeps an explicit choice of the caller).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (weak scaling)")
    ap.add_argument("--height", type=int, default=228)
    ap.add_argument("--width", type=int, default=304)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the product path); gloo only to rehearse N>1 on a one-GPU box")
    ap.add_argument("--exchange", default=os.environ.get("RDM_DP_EXCHANGE", "all_reduce"), choices=["all_reduce", "reduce_scatter"],
                    help="N > 1: all_reduce per backward stage (default), or reduce-scatter of the gradient bucket -> AdamW on the owned shard -> all-gather of the parameters")
    ap.add_argument("--workload", default="train", choices=["train", "fwd_bf16"],
                    help="train = the headline metric (BASELINE configs[2]); fwd_bf16 = BASELINE configs[1], batch=8 forward-only bf16")
    ap.add_argument("--backward-precision", default="bf16x3", choices=["bf16x3", "f32"],
                    help="bf16x3 (default, what the package ships): the gradient GEMMs of dense_e2 / dense_e3 on the split-precision kernels; f32: exact-f32 MFMA everywhere")
    ap.add_argument("--forward-split", type=int, default=1, help="1 (default, what the package ships): conv1 of dense_e2 / e3 on the three-way-split bf16x6 forward kernel (float32-equivalent); 0: f32 MFMA")
    ap.add_argument("--split-rows", type=int, default=1, help="1 (default, what the package ships): dY and relu1(norm1(x)) reach the split conv1 gradient kernels as split rows written once by their producers; 0: each kernel converts and splits per tile (A/B)")
    ap.add_argument("--fuse-stats3", type=int, default=1, help="1 (default, what the package ships): the K-split 3x3 conv of the few-pixel blocks takes its output's channel statistics in the same launch; 0: separate column reduction (A/B)")
    ap.add_argument("--wino-x6", type=int, default=0, help="0 (default, what the package ships): the f32 MFMA Winograd kernel; 1: conv2 of dense_e2 / e3 forward as Winograd on three-way-split bf16 MFMAs (float32-equivalent; measured not faster)")
    ap.add_argument("--gemm-bf16", type=int, default=0, choices=[0, 1, 2, 3],
                    help="MIXED-PRECISION arithmetic (the reference's default --precision 16): the GEMMs the two options above route to the split kernels round their operands to bf16, "
                    "one MFMA per product. 1: forward and gradient GEMMs, 2: forward only, 3: gradient GEMMs only. NOT the parity configuration - a separately labelled line, never the headline")
    ap.add_argument("--no-extra-configs", action="store_true", help="headline line only (skip BASELINE configs[1] and configs[4] at N=1)")
    args = ap.parse_args()
    if args.workload == "fwd_bf16":
        line = bench_fwd_bf16(args)
        if line is not None:
            emit(line)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    headline = (args.batch, args.height, args.width) == (16, 228, 304)
    out = bench_train(args, args.batch, args.height, args.width, with_cpu_baseline=not args.no_cpu_baseline)
    if out is not None and world == 1 and headline and not args.no_extra_configs:
        # BASELINE configs[1] (batch=8 forward-only bf16) and configs[4] (KITTI 352x1216 batch=8 train step), measured in this same
        # process right after the headline so the driver's BENCH record carries them under its own clock; the headline fields
        # above are untouched by them
        import gc

        def release():                                       # the previous configuration's workspace (7.6 GiB) goes back before the next is built
            gc.collect()
            torch.cuda.empty_cache()
        extra = []
        release()
        a2 = argparse.Namespace(**vars(args))
        a2.batch, a2.no_cpu_baseline, a2.steps, a2.warmup = 8, True, max(args.steps, 10), max(args.warmup, 2)
        def guarded(fn, *a, **kw):                           # a failing extra configuration must not cost the headline its line
            try:
                e = fn(*a, **kw)
                extra.append({k: e[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline")})
            except Exception as exc:                         # noqa: BLE001 - recorded in the line, loudly on stderr
                print(f"[bench] extra configuration failed: {exc!r}", file=sys.stderr, flush=True)
                extra.append({"metric": kw.get("metric", getattr(fn, "__name__", "?")), "value": None, "unit": "images/s", "ms_per_step": None, "dtype": "FAILED: " + repr(exc)[:80]})
        guarded(bench_fwd_bf16, a2)
        release()
        a3 = argparse.Namespace(**vars(args))
        a3.steps, a3.warmup = min(args.steps, 5), min(args.warmup, 2)
        guarded(bench_train, a3, 8, 352, 1216, with_cpu_baseline=False, metric="depth-maps/sec KITTI 352x1216 batch=8 fwd+bwd")
        if not args.gemm_bf16:
            # the headline geometry once more in the mixed-precision arithmetic mode (reference default --precision 16, train.py:11,57-58)
            release()
            for mode, what in ((1, "bf16 operands: fwd + gradient GEMMs"), (3, "bf16 operands: gradient GEMMs only")):
                a4 = argparse.Namespace(**vars(args))
                a4.gemm_bf16 = mode
                guarded(bench_train, a4, 16, 228, 304, with_cpu_baseline=False, metric=f"depth-maps/sec NYU 228x304 batch=16 fwd+bwd, mixed precision ({what})")
                release()
        out["config"]["extra_configs"] = extra
    if out is not None:
        emit(out)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 4096                      # bytes of the printed line; the driver's record keeps ~8.9 KB of stdout (round 4 lost a 25.7 KB line)


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(full):
    """The printed record: the contract's fields + `roofline` + `cpu_baseline`, every string bounded, nothing nested deeper than one
    summary row per extra configuration.  `full` (per-kernel tables, the extra configurations with their own rooflines, the prose) is
    what DETAIL_FILE holds."""
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = _short(full.get("dtype"), 120)
    line["data"] = full.get("data")
    c = {"workload": _short(cfg.get("workload"), 110), "global_batch": cfg.get("global_batch"), "parallelism": cfg.get("parallelism")}
    if cfg.get("precision_short"):
        c["precision"] = _short(cfg["precision_short"], 200)
    if cfg.get("loss") is not None:
        c["loss"] = round(float(cfg["loss"]), 4)
    comm = cfg.get("comm")
    if comm:
        c["comm"] = {k: (_short(v, 90) if isinstance(v, str) else v) for k, v in comm.items()
                     if k in ("exchange_short", "allreduce_bytes_per_step", "stages", "exposed_wait_ms_per_step", "per_stage_join_overhead_ms", "optimizer")}
    if cfg.get("extra_configs"):
        c["extra_configs"] = [{"metric": _short(e.get("metric"), 100), "value": e.get("value"), "unit": e.get("unit"), "ms_per_step": e.get("ms_per_step"),
                               "dtype": _short(e.get("dtype"), 40), "roofline_frac": (e.get("roofline") or {}).get("frac"),
                               "roofline_bound": (e.get("roofline") or {}).get("bound")} for e in cfg["extra_configs"]]
    line["config"] = c
    r = full.get("roofline")
    if r:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "pipe", "algorithmic_tflops", "algorithmic_GBps", "traffic", "traffic_source", "avg_launch_us",
                "launches_per_step", "mfma_busy_frac", "algorithmic_tflop_per_step", "executed_f32_tflop_per_step", "executed_bf16_tflop_per_step",
                "library_launches_per_step_all_kernels", "timing")
        line["roofline"] = {k: (_short(r[k], 150) if isinstance(r[k], str) else r[k]) for k in keep if k in r and r[k] is not None or k == "traffic" and k in r}
        # next kernels by summed duration: name, ms per step, fraction of their own pipe (the full table is in DETAIL_FILE)
        if r.get("per_kernel"):
            line["roofline"]["next"] = [{"kernel": e["kernel"].split(" ")[0], "ms": e["ms_sum_per_step"], "frac": e.get("frac", e.get("mfma_frac")), "pipe": e.get("pipe", "bf16")}
                                        for e in r["per_kernel"][1:5]]
    else:
        line["roofline"] = None
    b = full.get("cpu_baseline")
    if b:
        line["cpu_baseline"] = {k: (_short(b[k], 230) if isinstance(b[k], str) else b[k]) for k in ("value", "unit", "cores", "kind", "cpu", "sample") if k in b}
    else:
        line["cpu_baseline"] = None
    line["detail"] = DETAIL_FILE
    return line


def emit(full):
    """Write the full record to DETAIL_FILE (next to this script, and under gpurun_out/ when that exists so a gpurun call brings it
    back) and print the compact line.  The line is checked against LINE_LIMIT before it is printed: an oversized record must fail
    here, not silently in the driver's parser."""
    line = compact_line(full)
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:                              # drop the optional parts, in this order
        for victim in (("roofline", "next"), ("roofline", "timing"), ("roofline", "traffic_source"), ("config", "extra_configs")):
            node = line.get(victim[0]) or {}
            node.pop(victim[1], None)
            text = json.dumps(line)
            if len(text) < LINE_LIMIT:
                break
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes"
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as fh:
                    json.dump(full, fh, indent=1)
            except OSError:
                pass
    print(text, flush=True)
    return line


def host_threads():
    """Threads the CPU baseline may use = what the box really gives this process: the affinity mask, capped by the cgroup CPU quota
    when one is set (a quota of 16 cores under a 64-core mask makes 64 threads SLOWER than 16), overridable with RDM_CPU_THREADS."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    n = min(aff, quota) if quota else aff
    if os.environ.get("RDM_CPU_THREADS"):
        n = max(1, min(n, int(os.environ["RDM_CPU_THREADS"])))
    return n, f"affinity {aff}, cgroup quota {quota if quota else 'none'}, cpu_count {os.cpu_count()}"


def newest_profile(pattern):
    """profiles/rNN_<pattern>: the file of the latest round (PMC passes cannot run inside the bench process; tools/hbm_traffic.py
    turns the two rocprofv3 --pmc passes of the same command into it)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return c[-1] if c else None


_dist_ready = False


def bench_train(args, B, H, W, with_cpu_baseline, metric="depth-maps/sec NYU 228x304 batch=16 fwd+bwd"):
    global _dist_ready
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1 and not _dist_ready:
        _dist_ready = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = torch.cuda.device_count()
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if world > 1 and args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from md_rdm_amd import _lib, filler, harness, parallel
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    L = _lib.lib()
    if os.environ.get("RDM_VARIANT"):                     # development A/B switch, 0 = shipped
        L.rdm_debug_variant(int(os.environ["RDM_VARIANT"]))

    model = DepthEstimationNet()
    model.backward_precision = args.backward_precision
    model.forward_split = bool(args.forward_split)
    model.gemm_bf16 = int(args.gemm_bf16)
    model.split_rows = bool(args.split_rows)
    model.wino_x6 = bool(args.wino_x6)
    model.fuse_stats3 = bool(args.fuse_stats3)
    filler.fill_state_dict(model.state_dict())
    model = model.to(dev)
    model.train()
    x, y = filler.synthetic_batch(B, H, W, seed=1234 + rank)
    xg, yg = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    model.flatten_parameters()
    sync = parallel.attach(model, exchange=args.exchange)
    opt = harness.FusedAdamW(model, lr=1e-4)

    def step():
        opt.zero_grad()
        loss, _ = harness.training_step(model, xg, yg)
        loss.backward()
        if world > 1:
            opt.step(sync=sync)                               # per bucket: stage k's AdamW as its reduction lands
        else:
            opt.step(grad_scale=sync.finish())
        return loss

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"model ready, workspace {model._plan(B, H, W)[1] / 2**30:.2f} GiB")
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        note(f"warmup {i} done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    launches0 = L.rdm_launch_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib_launches_per_step = (L.rdm_launch_count() - launches0) / args.steps
    if not args.no_roofline:
        # roofline leg: the SAME K steps again, immediately after the timed region, with a HIP event pair
        # recorded around every conv launch (the events cost ~3.5 us each between dependent kernels,
        # ~3 % of a step, so they are kept out of the throughput clock)
        L.rdm_profile_enable(1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roof = None
    if not args.no_roofline:
        ms_sum, ms, fl, n = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        _lib.check(L.rdm_profile_read(C.byref(ms_sum), C.byref(ms), C.byref(fl), C.byref(n)))
        L.rdm_profile_enable(0)
        h = model._plan(B, H, W)[0]
        algo = (L.rdm_net_forward_flops(h) + L.rdm_net_backward_flops(h)) * args.steps   # reference-algorithmic conv FLOPs
        PEAK_F32, PEAK_BF16 = 157.3, 2500.0                     # dense MFMA peaks (TFLOP/s), MI355X_MICROARCH.md: f32-in/f32-acc, bf16
        peak = PEAK_F32
        achieved = algo / (ms.value * 1e-3) / 1e12
        # what each kernel family EXECUTES on the matrix pipe per algorithmic FLOP: the Winograd kernels (kinds 9, 10) multiply 1 / 2.25 as
        # much on the f32 pipe; the split-precision kernels (13-16) run three bf16 MFMAs per product on the bf16 pipe
        def pipe_of(kind):
            if kind == 18:                                      # Winograd on the bf16 pipe: 6 products per float32 product, 1 / 2.25 of the direct FLOPs
                return ("bf16", 6.0 / 2.25, PEAK_BF16)
            return ("bf16", (1.0 if args.gemm_bf16 in (1, 2) else 6.0) if kind == 17 else (1.0 if args.gemm_bf16 in (1, 3) else 3.0), PEAK_BF16) if kind >= 13 else ("f32", 1.0 / 2.25 if kind in (9, 10) else 1.0, PEAK_F32)
        per_kernel, busy_ms = [], 0.0
        executed_f32 = executed_bf16 = 0.0
        for kind in list(range(11)) + [13, 14, 15, 16, 17, 18]:
            nm, kms, kfl, kn = C.c_char_p(), C.c_double(), C.c_double(), C.c_int32()
            _lib.check(L.rdm_profile_kind(kind, C.byref(nm), C.byref(kms), C.byref(kfl), C.byref(kn)))
            if kn.value:
                pipe, mult, ppeak = pipe_of(kind)
                sec = kms.value * 1e-3
                ex = kfl.value * mult
                if pipe == "f32":
                    executed_f32 += ex
                else:
                    executed_bf16 += ex
                busy_ms += ex / (ppeak * 1e12) * 1e3             # time the pipe needs for these FLOPs at its peak
                per_kernel.append({"kernel": nm.value.decode(), "launches_per_step": kn.value // max(args.steps, 1),
                                   "avg_launch_us": round(kms.value / kn.value * 1e3, 1), "ms_sum_per_step": round(kms.value / args.steps, 3),
                                   "tflops": round(kfl.value / sec / 1e12, 1), "pipe": pipe, "executed_tflops": round(ex / sec / 1e12, 1),
                                   "pipe_peak": ppeak, "frac": round(ex / sec / 1e12 / ppeak, 3)})
        per_kernel.sort(key=lambda r: -r["ms_sum_per_step"])
        # `roofline` follows the contract literally: the DOMINANT kernel (largest summed duration) with algorithmic FLOPs per launch /
        # its average launch duration measured live (HIP events on its launch stream; NB it runs on the side stream BESIDE the dgrad
        # chain, so this is its rate while sharing the chip).  The whole conv family over the union of its intervals and the
        # driver-timed whole-step figure are given next to it.
        dom = per_kernel[0]
        traffic, traffic_src, fam_traffic = None, None, None
        tpath = newest_profile("hbm_traffic.json")
        if tpath and (B, H, W) == (16, 228, 304):     # PMC passes cannot run inside this process: measured figures of the same workload
            with open(tpath) as fh:
                tj = json.load(fh)
            fam_traffic = tj["conv_kernels"]["bytes_per_launch"]
            key = dom["kernel"].split(" ")[0]
            fam = [e for e in tj.get("per_kernel", []) if key in e["kernel"] and (("taps" in dom["kernel"]) == ("true>" in e["kernel"]) or "wgrad_kernel" not in key)]
            if fam:       # HBM-side bytes per launch, averaged over every instantiation of the dominant kernel family in the step
                traffic = round(sum(2 * e["fetch_raw_bytes_per_step"] + e["write_bytes_per_step"] for e in fam) / sum(e["launches_per_step"] for e in fam))
            traffic_src = ("profiles/" + os.path.basename(tpath) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH doubled per the gfx950 "
                           "correction, checked on k_adamw), bytes per launch of this kernel averaged over one step")
        step_frac = algo / args.steps / (elapsed / args.steps) / 1e12 / peak
        roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["executed_tflops"], "peak": dom["pipe_peak"], "unit": "TFLOP/s", "frac": dom["frac"],
                "algorithmic_tflops": dom["tflops"], "pipe": dom["pipe"],
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": dom["avg_launch_us"], "launches_per_step": dom["launches_per_step"],
                "note": "dominant kernel by summed duration; achieved = FLOPs it executes on its matrix pipe / its average launch duration (for a direct f32 kernel = the algorithmic FLOPs)",
                # the whole conv family and the whole step as ALGORITHMIC rates only: 94 % of the executed FLOPs run on the bf16 pipe at 3x / 6x the
                # algorithmic count and Winograd executes 1 / 2.25 of it on the f32 pipe, so no single peak divides them (round 4 printed a
                # "fraction" of the f32 peak here that exceeded 1); the utilisation figure of the step is mfma_busy_frac below
                "conv_family": {"kernel": "all MFMA conv kernels (f32, Winograd f32, bf16 split)", "algorithmic_tflops": round(achieved, 2),
                                "basis": "algorithmic conv FLOPs of the step / union of the conv kernels' intervals", "traffic_per_launch": fam_traffic,
                                "launches_per_step": n.value // max(args.steps, 1), "kernel_ms_per_step": round(ms.value / args.steps, 3),
                                "kernel_ms_sum_per_step": round(ms_sum.value / args.steps, 3)},
                "whole_step": {"algorithmic_tflops": round(step_frac * peak, 2), "basis": "algorithmic conv FLOPs / ms_per_step (driver-timed)"},
                "per_kernel": per_kernel, "library_launches_per_step_all_kernels": round(lib_launches_per_step, 1),
                "timing": "HIP events on the launch streams over K further steps run right after the timed region",
                "algorithmic_tflop_per_step": round(algo / args.steps / 1e12, 4),
                "executed_f32_tflop_per_step": round(executed_f32 / args.steps / 1e12, 4), "executed_bf16_tflop_per_step": round(executed_bf16 / args.steps / 1e12, 4),
                # fraction of the step during which the matrix pipes would be busy if every MFMA ran at its pipe's peak rate
                "mfma_busy_frac": round(busy_ms / args.steps / (elapsed / args.steps * 1e3), 4)}

    # communication figures of the line (config.comm): what the exchange moves, how much of it the backward pass did not hide, what the
    # per-stage ordering of the weight-gradient stream costs - measured over a few further steps, outside the timed region
    comm = {"allreduce_bytes_per_step": sync.bytes_per_step() if world > 1 else 0, "stages": len(sync.slices),
            "exchange": "one asynchronous RCCL all-reduce (sum) per backward stage on the flat gradient buffer, last layers first; AdamW applies 1/N",
            "exchange_short": sync.exchange if world > 1 else "none (N=1)", "optimizer": getattr(opt, "mode", "after the last stage")}
    if world > 1:
        sync.timing = True
        for _ in range(min(args.steps, 3)):
            step()
        comm["exposed_wait_ms_per_step"] = round(sync.exposed_ms() or 0.0, 3)
        comm["exposed_wait_basis"] = "device events around GradSync.finish(): compute stream idle until the last reduction has landed"
        comm["side_stream_join"] = "after every backward stage (the exchange consumes gradients stage by stage)"
        sync.timing = False
    elif not args.no_roofline:
        # N = 1: the cost of joining the weight-gradient stream after each of the 13 stages (what N > 1 pays) instead of once per segment
        hook = model.grad_ready_hook
        model.grad_ready_hook = (lambda stage: None) if hook is None or world == 1 else hook
        torch.cuda.synchronize()
        tj = time.perf_counter()
        for _ in range(min(args.steps, 5)):
            step()
        torch.cuda.synchronize()
        comm["per_stage_join_overhead_ms"] = round((time.perf_counter() - tj) / min(args.steps, 5) * 1e3 - elapsed / args.steps * 1e3, 3)
        model.grad_ready_hook = hook
    note(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    cpu = None
    if rank == 0 and world == 1 and with_cpu_baseline:
        cpu = cpu_baseline(H, W)
        note("cpu baseline done")

    out = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        out = {"metric": metric, "value": round(B * world * args.steps / elapsed, 3), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": ("bf16-mixed (bf16 GEMM operands, f32 accumulation, in " + {1: "dense_e2/e3 conv1 forward and the gradient GEMMs of dense_e2/e3/e4", 2: "dense_e2/e3 conv1 forward",
                                                                                                                                    3: "the gradient GEMMs of dense_e2/e3/e4"}[args.gemm_bf16] + "; all else f32)" if args.gemm_bf16 else "f32" if args.backward_precision == "f32" and not args.forward_split else
                                                                   "f32 (" + ", ".join(([] if not args.forward_split else ["bf16x6-split 1x1 forward of dense_e2/e3"]) +
                                                                                      ([] if args.backward_precision == "f32" else ["bf16x3-split gradient GEMMs of dense_e2/e3/e4"])) + ")"), "data": "synthetic",
               "config": {"workload": f"{'KITTI' if (H, W) == (352, 1216) else 'NYU-v2'} {H}x{W} batch={B}/GPU full train step (fwd+losses+bwd+AdamW), DepthEstimationNet 90.5M params",
                          "global_batch": B * world, "parallelism": f"dp{world}", "loss": float(loss.item()),
                          "precision": ("MIXED (not the parity configuration): " + {1: "the 1x1 forward GEMMs of dense_e2 / e3 and the weight / input gradient GEMMs of dense_e2 / e3 / e4",
                                                                                                   2: "the 1x1 forward GEMMs of dense_e2 / e3", 3: "the weight / input gradient GEMMs of dense_e2 / e3 / e4"}[args.gemm_bf16] +
                                        " round their float32 operands to bf16 (ONE bf16 MFMA per product, float32 accumulation); activations, weights, BatchNorm statistics, every other convolution, "
                                        "losses and AdamW stay float32. Measured against the default path at B=4 228x304 (tests/test_gpu_mixed.py): " +
                                        ("logits identical, every gradient tensor's cosine >= 0.9999, loss after 4 AdamW steps within 0.1 %" if args.gemm_bf16 == 3 else
                                         "logits RMS 3.2 % (36 bf16 GEMMs under training-mode BatchNorm), gradient cosine 0.84 .. 1.0 per tensor (the FORWARD rounding moves the loss gradient: "
                                         "gradient-only rounding keeps >= 0.9999), loss after 4 AdamW steps within 1 %")
                                        if args.gemm_bf16 else
                                        ("forward, losses, BatchNorm, AdamW: float32 - convolutions on the exact-f32 MFMA"
                                         + (", except conv1 (1x1) of dense_e2 / dense_e3: operands split three ways into bf16, six bf16 MFMAs per product, float32 accumulation (float32-equivalent: "
                                            "3-9e-7 of the result's maximum vs float64, the f32 kernel's own level); " if args.forward_split else "; ")
                                         + ("weight / input gradient GEMMs of dense_e2 / e3 / e4: float32 operands split into bf16 hi + lo, three bf16 MFMAs per product, float32 accumulation "
                                            "(~5e-6 of a gradient's maximum vs ~1e-6 for the f32 kernels)" if args.backward_precision != "f32" else "every gradient GEMM on the f32 pipe")
                                         + "; --forward-split 0 --backward-precision f32 = float32 MFMA everywhere")),
                          "precision_short": ("MIXED, not the parity configuration: bf16-rounded GEMM operands (one MFMA per product, f32 accumulation), mode " + str(args.gemm_bf16)
                                              if args.gemm_bf16 else
                                              "f32 tensors; " + ("dense_e2/e3 conv1 fwd on bf16x6 split MFMA (f32-equivalent); " if args.forward_split else "") +
                                              ("dense-block gradient GEMMs on bf16x3 split MFMA (~5e-6 of max); " if args.backward_precision != "f32" else "") +
                                              "all else exact-f32 MFMA"),
                          "comm": comm},
               "roofline": roof, "cpu_baseline": cpu}
    return out


def bench_fwd_bf16(args):
    """BASELINE configs[1]: "NYU-v2 batch=8 forward-only bf16, 1xMI355X".  A step = one eval-mode DepthEstimationNet.forward (conv
    stack on the bf16 MFMA path + DORN head + decomposition tail) on a synthetic batch resident in HBM."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:                                           # inference shards by sample: independent replicas, no data-path collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    from md_rdm_amd import _lib, filler
    from md_rdm_amd.network.RDM_Net import DepthEstimationNet
    L = _lib.lib()
    if os.environ.get("RDM_VARIANT"):                     # development A/B switch, 0 = shipped
        L.rdm_debug_variant(int(os.environ["RDM_VARIANT"]))
    B = 8 if args.batch == 16 else args.batch
    H, W = args.height, args.width
    model = DepthEstimationNet()
    filler.fill_state_dict(model.state_dict())
    model = model.to(dev).eval().set_precision("bf16")
    x, _ = filler.synthetic_batch(B, H, W, seed=1234 + rank)
    xg = torch.from_numpy(x).to(dev)

    def step():
        with torch.no_grad():
            return model(xg)

    for _ in range(max(args.warmup, 1)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # conv-stack-only time (the tail is ~10 small launches): same K steps of the native forward alone
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        with torch.no_grad():
            model._native_forward_bf16(xg)
    torch.cuda.synchronize()
    stack_ms = (time.perf_counter() - t1) / args.steps * 1e3
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roof = None
    if not args.no_roofline:
        L.rdm_profile_enable(1)
        for _ in range(args.steps):
            with torch.no_grad():
                model._native_forward_bf16(xg)
        torch.cuda.synchronize()
        ms_sum, ms, fl, n = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        _lib.check(L.rdm_profile_read(C.byref(ms_sum), C.byref(ms), C.byref(fl), C.byref(n)))
        L.rdm_profile_enable(0)
        h = model._plan(B, H, W)[0]
        PEAK_MFMA, PEAK_HBM = 2500.0, 8000.0                     # dense bf16 MFMA TFLOP/s, HBM3E GB/s (MI355X_MICROARCH.md)
        per_kernel = []
        for kind in (7, 8, 11, 12):
            nm, kms, kfl, kn = C.c_char_p(), C.c_double(), C.c_double(), C.c_int32()
            _lib.check(L.rdm_profile_kind(kind, C.byref(nm), C.byref(kms), C.byref(kfl), C.byref(kn)))
            kby = L.rdm_profile_kind_bytes(kind)
            if kn.value:
                sec = kms.value * 1e-3
                per_kernel.append({"kernel": nm.value.decode(), "launches_per_step": kn.value // args.steps, "avg_launch_us": round(kms.value / kn.value * 1e3, 2),
                                   "ms_sum_per_step": round(kms.value / args.steps, 3), "tflops": round(kfl.value / sec / 1e12, 1),
                                   "mfma_frac": round(kfl.value / sec / 1e12 / PEAK_MFMA, 4), "algorithmic_GBps": round(kby / sec / 1e9, 1),
                                   "hbm_frac": round(kby / sec / 1e9 / PEAK_HBM, 4), "flop_per_byte": round(kfl.value / max(kby, 1), 1)})
        per_kernel.sort(key=lambda r: -r["ms_sum_per_step"])
        dom = per_kernel[0]
        bound = "mfma" if dom["mfma_frac"] >= dom["hbm_frac"] else "hbm"       # the roofline the dominant kernel sits closer to
        algo_fl = L.rdm_net_forward_flops(h)
        traffic, traffic_src = None, None
        tpath = newest_profile("hbm_traffic_bf16.json")
        if tpath and (B, H, W) == (8, 228, 304):      # PMC passes cannot run inside this process: measured figures of the same workload
            with open(tpath) as fh:
                tj = json.load(fh)
            e = tj["kernels"].get(dom["kernel"].split(" ")[0])
            if e:
                traffic, traffic_src = e["traffic_bytes_per_launch"], "profiles/" + os.path.basename(tpath) + ": " + tj["source"]
        roof = {"bound": bound, "kernel": dom["kernel"],
                "achieved": dom["tflops"] if bound == "mfma" else dom["algorithmic_GBps"], "peak": PEAK_MFMA if bound == "mfma" else PEAK_HBM,
                "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": dom["mfma_frac"] if bound == "mfma" else dom["hbm_frac"], "traffic": traffic, "traffic_source": traffic_src,
                "per_kernel": per_kernel, "conv_kernel_ms_per_step": round(ms_sum.value / args.steps, 3), "conv_launches_per_step": n.value // args.steps,
                "whole_stack": {"algorithmic_tflop_per_step": round(algo_fl / 1e12, 4), "algorithmic_GB_per_step": round(L.rdm_net_bf16_forward_bytes(h) / 1e9, 3),
                                "stack_ms_per_step": round(stack_ms, 3), "mfma_frac": round(algo_fl / (stack_ms * 1e-3) / 1e12 / PEAK_MFMA, 4),
                                "hbm_frac": round(L.rdm_net_bf16_forward_bytes(h) / (stack_ms * 1e-3) / 1e9 / PEAK_HBM, 4)},
                "timing": "HIP events on the launch stream around every bf16 conv launch over K further forwards run right after the timed region"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_forward(H, W)
    out_line = None
    if rank == 0:
        name = "NYU 228x304" if (H, W) == (228, 304) else ("KITTI 352x1216" if (H, W) == (352, 1216) else f"{H}x{W}")
        out_line = {"metric": f"depth-maps/sec {name} batch={B} forward bf16", "value": round(B * world * args.steps / elapsed, 3), "unit": "images/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": f"{name} batch={B}/GPU forward-only bf16 (eval-mode DepthEstimationNet.forward: conv stack + DORN head + decomposition tail)",
                               "global_batch": B * world, "parallelism": f"replicas{world}"},
                    "roofline": roof, "cpu_baseline": cpu}
    if world > 1:
        dist.destroy_process_group()
    return out_line


def cpu_baseline_forward(H, W, batch=2, iters=3):
    """The oracle's eval-mode forward (PyTorch-CPU restatement of the reference, float32) on the host cores."""
    from md_rdm_amd import filler
    from oracle import rdm_net_cpu as onet
    n, how = host_threads()
    torch.set_num_threads(n)
    sd = onet.new_state_dict(filler.state_value)
    x, _ = filler.synthetic_batch(batch, H, W, seed=1234)
    xt = torch.from_numpy(x)
    with torch.no_grad():
        onet.forward(sd, xt, training=False)
        t0 = time.perf_counter()
        for _ in range(iters):
            onet.forward(sd, xt, training=False)
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} eval-mode forwards at batch {batch}, {H}x{W}, float32 (the CPU has no bf16 conv path worth timing), after 1 warm-up"}


def cpu_baseline(H, W, batch=16, warmup=1, iters=2):
    """the oracle's FULL train step (PyTorch-CPU restatement of the reference: forward + losses + backward
    + torch.optim.AdamW over the 491 parameter tensors) at the headline batch of 16, float32, BatchNorm in train mode, on the GPU
    box's host cores.  The plan's "3 warm-up + 5 timed" would be ~2.5 minutes of CPU at ~15 s per step; bounded here to 1 warm-up
    + 2 timed steps (~45 s) so the default bench run stays within minutes (BASELINE.md 3 records this as what ships) - the sample
    says so.  RDM_CPU_BASELINE_FULL=1 runs 3 + 5 (profiles/ keeps one such sample per round)."""
    import numpy as np
    if os.environ.get("RDM_CPU_BASELINE_FULL", "0") not in ("", "0"):
        warmup, iters = 3, 5
    from md_rdm_amd import filler
    from oracle import rdm_net_cpu as onet
    n, how = host_threads()                                             # every core the box really gives this process
    torch.set_num_threads(n)
    batch = int(os.environ.get("RDM_CPU_BATCH", str(batch)))
    sd = onet.new_state_dict(filler.state_value)
    params = [(k, v) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k and v.numel()]
    leaves = [torch.nn.Parameter(v) for _, v in params]
    for (k, _), p in zip(params, leaves):
        sd[k] = p.data                                    # the optimiser steps the very tensors the forward reads
    opt = torch.optim.AdamW(leaves, lr=1e-4)
    x, y = filler.synthetic_batch(batch, H, W, seed=1234)
    xt = torch.from_numpy(x)

    def step():
        out = onet.training_step(sd, xt, y)
        for (k, _), p in zip(params, leaves):
            p.grad = out["grads"].get(k)
        opt.step()

    times = []
    for i in range(warmup + iters):
        t0 = time.perf_counter()
        step()
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "")
    except OSError:
        pass
    out = {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model,
           "sample": f"median of {iters} full train steps (fwd+losses+bwd+AdamW) at batch {batch}, {H}x{W}, fp32, BN train mode, after {warmup} warm-up; {dt:.1f} s/step; "
                     f"threads={n} ({how}); RDM_CPU_BASELINE_FULL=1 runs 3+5"}
    out["input_pipeline"] = input_pipeline_baseline(H, W)
    return out


def input_pipeline_baseline(H, W, B=16, seconds=3.0):
    """SURVEY.md 8(f)1 beside the step: training_preprocess of raw 480x640 NYU frames - the reference's Pillow path
    (oracle/preprocess_cpu.py: the calls torchvision's transforms forward to) on one host core vs the GPU kernel chain."""
    import numpy as np
    from md_rdm_amd.dataloaders import nyu
    from oracle import preprocess_cpu as P
    rng = np.random.default_rng(0)
    raws = [(rng.integers(0, 256, (480, 640, 3)).astype(np.uint8), (rng.random((480, 640)) * 9.5 + 0.5).astype(np.float32)) for _ in range(B)]
    draws = [nyu.draw_training_params(rng, (480, 640), 250, (H, W)) for _ in range(B)]
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        d = draws[n % B][0]
        P.pil_training_preprocess(raws[n % B][0], raws[n % B][1], d["s"], d["angle"], d["flip"], d["jitter"], 250, (H, W))
        n += 1
    cpu_rate = n / (time.perf_counter() - t0)
    dev = torch.device("cuda", torch.cuda.current_device())
    rgb = torch.from_numpy(np.stack([r for r, _ in raws])).to(dev)
    dep = torch.from_numpy(np.stack([d for _, d in raws])).to(dev)
    pre, params = nyu.NyuGpuPreprocessor(250, (H, W)), [p for _, p in draws]
    pre(rgb, dep, params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pre(rgb, dep, params)
    torch.cuda.synchronize()
    gpu_rate = 20 * B / (time.perf_counter() - t0)
    return {"cpu_images_per_s_per_core": round(cpu_rate, 1), "gpu_images_per_s": round(gpu_rate, 0), "kind": "reference dependency (Pillow)",
            "sample": f"{n} frames in {seconds:.0f} s on 1 core; GPU: 20 batches of {B}, inputs resident in HBM"}


if __name__ == "__main__":
    main()
