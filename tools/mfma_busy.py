"""Per kernel of one training step: matrix-pipe busy fraction per active cycle, wait / active shares of the wave cycles, from a rocprofv3 counter collection
(counters serialise the kernels: these are the kernels ALONE on the chip, not sharing it with the other stream).
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d OUT -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-configs
    python tools/mfma_busy.py OUT/p_counter_collection.csv
mfma busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32): the share of the kernel's busy cycles in which a SIMD's matrix pipe is executing (the formula of tools/pmc_summary.py)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
disp = {}
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
order = sorted(disp.values(), key=lambda d: d["start"])
idx = [i for i, d in enumerate(order) if "k_im2col_stem" in d["name"]]
step = order[idx[-2]:idx[-1]] if len(idx) >= 2 else order
agg = {}
for d in step:
    n = d["name"].replace("void ", "").replace("rdm::", "").replace("(anonymous namespace)::", "").split("(")[0][:56]
    a = agg.setdefault(n, {"n": 0, "dur": 0.0})
    a["n"] += 1; a["dur"] += d["dur"]
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
        a[c] = a.get(c, 0.0) + d.get(c, 0.0)
print(f"{len(step)} launches of one step, {sum(d['dur'] for d in step) / 1e3:.1f} ms of kernel time (serialised by the counter collection)")
print(f"{'kernel':56s} {'launches':>8s} {'ms':>8s} {'us each':>8s} {'mfma busy':>9s} {'wait':>6s} {'active':>6s}")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    if a["dur"] < 50: continue
    busy = a.get("SQ_BUSY_CYCLES", 0.0) / 32 + 1
    wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print(f"{n:56s} {a['n']:8d} {a['dur'] / 1e3:8.3f} {a['dur'] / a['n']:8.1f} {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / 1024 / busy:9.2f} {a.get('SQ_WAIT_ANY', 0.0) / wc:6.2f} {a.get('SQ_ACTIVE_INST_ANY', 0.0) / wc:6.2f}")
