"""HBM-side bytes of the bf16 forward kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, each with
--kernel-trace only) over tools/bf16_fwd_loop.py, next to the algorithmic bytes per launch of a bench line.

    N=2 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcb_f -o f -- python3 tools/bf16_fwd_loop.py
    N=2 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcb_w -o w -- python3 tools/bf16_fwd_loop.py
    python tools/hbm_traffic_bf16.py gpurun_out/pmcb_f/f_counter_collection.csv gpurun_out/pmcb_w/w_counter_collection.csv [bench_fwd_bf16.json] > profiles/rNN_hbm_traffic_bf16.json

Units / corrections (MI355X_MICROARCH.md, "HBM"): both counters are KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
(16 B / lane) streaming reads - plain loads and LDS-DMA alike - at 64 bytes, so the read side is doubled; WRITE_SIZE is taken as is."""
import collections, csv, json, re, sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != name:
                continue
            k = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "").replace("rdm::", "")).split("(")[0]
            a = agg[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"]) * 1024.0
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    algo = {}
    if len(sys.argv) > 3:
        line = json.load(open(sys.argv[3]))
        line = line.get("extra_configs", [{}])[0] if "per_kernel" not in (line.get("roofline") or {}) or "bf16" not in line.get("dtype", "") else line
        for pk in (line.get("roofline") or {}).get("per_kernel", []):
            name = pk["kernel"].split(" ")[0]
            algo[name] = pk["algorithmic_GBps"] * 1e9 * pk["avg_launch_us"] * 1e-6
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) over tools/bf16_fwd_loop.py "
                     "(B=8 228x304 bf16 forwards); FETCH_SIZE is KiB and DOUBLED per the gfx950 correction of MI355X_MICROARCH.md (128-B requests of wide "
                     "16-B-per-lane reads, plain and LDS-DMA, tallied at 64 B); WRITE_SIZE (KiB) as is",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0, 0])[1])):
        if not any(t in k for t in ("bf16", "k_reduce_partials")):
            continue
        n = fetch[k][0]
        f, w = 2.0 * fetch[k][1] / n, write.get(k, [0, 0.0, 0])[1] / max(write.get(k, [1])[0], 1)
        e = {"launches": n, "fetch_bytes_per_launch_corrected": round(f), "write_bytes_per_launch": round(w), "traffic_bytes_per_launch": round(f + w),
             "avg_us_under_pmc": round(fetch[k][2] / n, 1)}
        if k in algo:
            e["algorithmic_bytes_per_launch"] = round(algo[k])
            e["traffic_over_algorithmic"] = round((f + w) / algo[k], 2)
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


main()
